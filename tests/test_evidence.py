"""The record must say what was measured: every figure of DESIGN.md's evidence tables (section 6: rows that end in a
`profiles/...` file) is looked up in that file -- some number there must agree with it to the digits DESIGN.md prints -- and every
profiles/ file DESIGN.md, INTEGRATION.md or include/i3rc_hip.h cites must exist.  (Round 3's DESIGN.md quoted measurements whose
files had been overwritten by an older run: VERDICT.md round 3, "Evidence at HEAD is not the evidence DESIGN.md quotes".)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM = re.compile(r"(?<![\w.])[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def _sig(token):
    mant = re.split(r"[eE]", token)[0].lstrip("+-")
    digits = mant.replace(".", "").lstrip("0")
    if "." not in mant:
        digits = digits.rstrip("0") or "0"
    return max(1, len(digits))


def _rounded(x, s):
    return "%.*e" % (s - 1, x)


def _evidence_rows():
    rows = []
    for line in open(os.path.join(ROOT, "DESIGN.md")):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 3 and re.fullmatch(r"`profiles/[\w.\-]+`", cells[-1]):
            rows.append((cells[0], cells[1:-1], cells[-1].strip("`")))
    return rows


def test_design_evidence_tables_quote_what_the_profiles_hold():
    rows = _evidence_rows()
    assert len(rows) >= 25, len(rows)          # the tables are there at all
    missing = []
    for what, values, path in rows:
        full = os.path.join(ROOT, path)
        assert os.path.exists(full), (what, path)
        text = open(full).read()
        if path.endswith("_pmc.json"):
            # one file, every workload's counters: a row must say WHOSE figures it quotes -- `workload` in its first cell -- and only that
            # workload's entry is searched (round 4's DESIGN.md printed the step cloud's figures in the Landsat row and this test, which
            # then accepted any number anywhere in the file, could not see it)
            import json

            j = json.loads(text)
            named = [k for k in j if f"`{k}`" in what]
            assert len(named) == 1, ("a row that cites a *_pmc.json names its workload as `key` in its first cell", what, sorted(j))
            text = json.dumps(j[named[0]])
        numbers = [float(t) for t in NUM.findall(text)]
        for cell in values:
            for tok in NUM.findall(cell):
                s = _sig(tok)
                want = _rounded(float(tok), s)
                if not any(_rounded(x, s) == want for x in numbers if x != 0.0):
                    missing.append((what, tok, path))
    assert not missing, missing


def test_every_cited_profile_exists():
    cited = set()
    for doc in ("DESIGN.md", "INTEGRATION.md", os.path.join("include", "i3rc_hip.h"), "README.md", os.path.join("tools", "README.md")):
        cited |= set(re.findall(r"profiles/(r\d\d[\w.\-]*\.(?:txt|json|csv))", open(os.path.join(ROOT, doc)).read()))
    absent = sorted(c for c in cited if not os.path.exists(os.path.join(ROOT, "profiles", c)))
    assert not absent, absent
