"""Inputs and the stdin / stdout protocol of oracle/ref_dump.f95 (the caller of numericUtilities / surfaceProperties that
is linked once against the REFERENCE's modules -- oracle/_ref/ref_dump -- and once against the shell's).

`cases()` makes the inputs (deterministic: a seeded generator and written-out special values), `script()` turns a list
of cases into the program's stdin, `parse()` its stdout into arrays.  tests/golden/make_ref_numerics.py stores inputs AND
the reference's outputs in tests/golden/ref_numerics.npz; the tests read that file and never this module's `cases()`.

Reals travel as int32 bit patterns, so nothing is rounded on the way."""
import numpy as np

F = np.float32


def bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.int32)


def _near(values):
    """every value, its two float32 neighbours"""
    v = np.asarray(values, dtype=F)
    with np.errstate(over="ignore"):   # (the neighbour above huge is infinity: wanted)
        return np.concatenate([v, np.nextafter(v, F(np.inf)), np.nextafter(v, F(-np.inf))]).astype(F)


def cases():
    rng = np.random.default_rng(20261005)
    out = []

    def find_index(name, table, values, guess=None):
        out.append(dict(kind="findIndex", name=name, table=np.asarray(table, dtype=F), values=np.asarray(values, dtype=F),
                        guess=None if guess is None else np.asarray(guess, dtype=np.int32)))

    # -- findIndex (Code/numericUtilities.f95:195-248) ------------------------------------------------------------------
    # regular edges (the step cloud's x edges), irregular edges, cumulative extinctions (/0, cum/) as :637 passes them
    # (equal neighbours: a component without extinction in the cell), the uniform surface's (/0, huge/), a CDF
    tables = {
        "stepX": np.arange(33, dtype=F) * F(15.625),
        "irregularZ": np.cumsum(np.concatenate([[0.0], rng.uniform(0.01, 2.0, 20)])).astype(F),
        "cum1": np.array([0.0, 1.0], dtype=F),
        "cum3": np.array([0.0, 0.3, 0.3, 1.0], dtype=F),
        "cum3first": np.array([0.0, 0.0, 0.25, 1.0], dtype=F),
        "cum3last": np.array([0.0, 0.25, 1.0, 1.0], dtype=F),
        "cum12": np.concatenate([[0.0], np.cumsum(rng.uniform(0.0, 1.0, 12))]).astype(F),
        "huge": np.array([0.0, np.finfo(F).max], dtype=F),
        "one": np.array([2.5], dtype=F),
        "cdf": (np.linspace(0.0, 1.0, 64) ** 3).astype(F),
    }
    tables["cum12"] = (tables["cum12"] / tables["cum12"][-1]).astype(F)
    for name, t in tables.items():
        lo, hi = float(t[0]), float(t[-1]) if np.isfinite(t[-1]) and t[-1] < 1e30 else 1e6
        inside = rng.uniform(lo, hi, 40).astype(F)
        mids = ((t[:-1].astype(np.float64) + t[1:]) / 2).astype(F)
        special = _near(t)
        below = np.array([lo - 1.0, lo - 1e-3, -1e30], dtype=F)
        above = np.array([hi + 1.0, 1e30, 3.0e38], dtype=F)
        # without a first guess: anything (below the table -> 0)
        find_index(name, t, np.concatenate([special, mids, inside, below, above]))
        # with one: values at or above table(1) only -- below it the reference's hunting loop never ends (:218-231:
        # lowerBound stays 1, the increment doubles for ever); its callers never do that
        v = np.concatenate([special, mids, inside, above])
        v = v[v >= t[0]]
        n = len(t)
        if n <= 4:   # every guess for every value
            vv, gg = np.repeat(v, n), np.tile(np.arange(1, n + 1), len(v))
        else:
            vv = np.tile(v, 3)
            gg = rng.integers(1, n + 1, len(vv))
        find_index(name + "+guess", t, vv, gg)
    # ... and the chain of inversePhaseFunctions.f95:133-136: each index is the next search's first guess
    cdf = tables["cdf"]
    probs = (np.arange(1, 2001, dtype=F) - F(1.0)) / F(2000.0)
    chain, g = [], 1
    for p in probs:   # (the guesses of the chain follow from the function's own definition: largest i with cdf(i) <= p)
        g = max(int(np.searchsorted(cdf, p, side="right")), 1)
        chain.append(g)
    find_index("cdf+chain", cdf, probs[1:], np.asarray(chain[:-1]))

    # -- quadratures and the Legendre recurrence (:15-193) ----------------------------------------------------------------
    for n in (2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 32, 33, 64, 65, 100, 128, 299, 300):
        out.append(dict(kind="lobatto", name=str(n), n=n))
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 32, 33, 64, 65, 128, 299):
        out.append(dict(kind="gauss", name=str(n), n=n))
    mus = np.concatenate([[-1.0, 1.0, 0.0, 0.5, -0.5, 0.85, 0.9999999, -0.9999999], np.cos(np.linspace(0, np.pi, 11)),
                          rng.uniform(-1, 1, 13)]).astype(F)
    for maxL in (1, 2, 3, 64, 299):
        out.append(dict(kind="legendre", name=str(maxL), maxL=maxL, mus=mus))

    # -- computeSurfaceReflectance (Code/surfaceProperties.f95:121-162) ---------------------------------------------------
    def surface(name, xs, ys, R, x, y):
        out.append(dict(kind="surface", name=name, xs=np.asarray(xs, dtype=F), ys=np.asarray(ys, dtype=F), R=np.asarray(R, dtype=F),
                        x=np.asarray(x, dtype=F), y=np.asarray(y, dtype=F)))

    def points(xs, ys, n):
        """inside, on the inner edges, one to three periods outside on either side -- never ON the outer edges or a whole
        number of periods away from them: makePeriodic maps those onto xMax (:232-241), findIndex answers size(xPosition)
        and the reference reads BRDFParameters one past its extent"""
        wx, wy = float(xs[-1] - xs[0]), float(ys[-1] - ys[0])
        x = rng.uniform(float(xs[0]), float(xs[-1]), n)
        y = rng.uniform(float(ys[0]), float(ys[-1]), n)
        k = rng.integers(-3, 4, n)
        l = rng.integers(-3, 4, n)
        x, y = (x + k * wx).astype(F), (y + l * wy).astype(F)
        if len(xs) > 2:   # inner edges and their neighbours
            ex = _near(xs[1:-1]); ey = _near(ys[1:-1]) if len(ys) > 2 else ys[:0]
            x = np.concatenate([x, ex, rng.uniform(float(xs[0]), float(xs[-1]), len(ey)).astype(F)])
            y = np.concatenate([y, rng.uniform(float(ys[0]), float(ys[-1]), len(ex)).astype(F), ey])
        ok = np.ones(len(x), bool)
        for a, e in ((x, xs), (y, ys)):   # keep clear of the outer edges' images
            w = np.float64(e[-1]) - np.float64(e[0])
            frac = np.abs(((a.astype(np.float64) - np.float64(e[0])) / w + 0.5) % 1.0 - 0.5)
            ok &= frac > 1e-3
        return x[ok], y[ok]

    xs = np.array([0.0, 1.5, 2.0, 7.0, 10.0], dtype=F); ys = np.array([-5.0, 0.0, 2.5, 20.0], dtype=F)
    surface("grid4x3", xs, ys, rng.uniform(0, 1, (3, 4)), *points(xs, ys, 300))            # R[y][x]: x fastest
    xs = (np.arange(9, dtype=F) * F(62.5)); ys = (np.arange(3, dtype=F) * F(250.0))
    surface("regular8x2", xs, ys, rng.uniform(0, 1, (2, 8)), *points(xs, ys, 300))
    surface("one1x1", [0.0, 500.0], [0.0, 500.0], [[0.35]], *points(np.array([0.0, 500.0], F), np.array([0.0, 500.0], F), 50))
    # what new_SurfaceDescription refuses (:72-83): positions not increasing, a reflectance above 1
    surface("refused:positions", [0.0, 2.0, 2.0], [0.0, 1.0], [[0.1, 0.2]], [0.5], [0.5])
    surface("refused:reflectance", [0.0, 1.0], [0.0, 1.0], [[1.5]], [0.5], [0.5])
    # the uniform surface new_SurfaceDescription((/ R /)): positions (/0, huge/) -- photons anywhere, also left of 0
    x = np.array([1.0, 250.0, 1e6, 1e30, -1.0, -250.0, -1e30, 3.0e38], dtype=F)
    out.append(dict(kind="uniform", name="0.2", R=np.array([0.2], dtype=F), x=x, y=x[::-1].copy()))
    # -- ErrorMessages (the status object of the boundary) and CharacterUtils ----------------------------------------------------------
    ops = ["S new_Domain: ok", "W getInfo: nothing to report", "f", "F computeRadiativeTransfer: no photons to process.", "s", "w",
           "C all clear", "F again", "I", "S after initializeState", "c", "W a text with   inner   blanks and trailing   "]
    ops += ["F " + "x" * 300]                                   # longer than a message can be (256)
    ops += [f"F overflow {i}" for i in range(1, 106)]           # more than the history holds (100): the last slot is overwritten
    ops += ["S the one after the overflow", "C cleared again"]
    out.append(dict(kind="errors", name="history", ops=np.array(ops)))
    cops = [f"I {v}" for v in (0, 7, -7, 42, 2147483647, -2147483647, 1000000)] + ["C 17", "C   -3  ", "C 0", "R 2.5", "R   -1.e-3", "R 100", "R 0.85"]
    out.append(dict(kind="chars", name="conversions", ops=np.array(cops)))
    return out


def _ints(a):
    a = np.asarray(a).astype(np.int64).ravel()
    return "\n".join(" ".join(str(int(v)) for v in a[i:i + 8]) for i in range(0, len(a), 8)) + "\n"


def script(cs):
    s = []
    for c in cs:
        k = c["kind"]
        if k == "findIndex":
            g = c.get("guess")
            s.append(f"findIndex {len(c['table'])} {len(c['values'])} {0 if g is None else 1}\n" + _ints(bits(c["table"])) + _ints(bits(c["values"])))
            if g is not None:
                s.append(_ints(g))
        elif k in ("lobatto", "gauss"):
            s.append(f"{k} {c['n']}\n")
        elif k == "legendre":
            s.append(f"legendre {c['maxL']} {len(c['mus'])}\n" + _ints(bits(c["mus"])))
        elif k == "surface":
            s.append(f"surface {len(c['xs']) - 1} {len(c['ys']) - 1} {len(c['x'])}\n" + _ints(bits(c["xs"])) + _ints(bits(c["ys"])) +
                     _ints(bits(c["R"])) + _ints(bits(c["x"])) + _ints(bits(c["y"])))
        elif k == "uniform":
            s.append(f"uniform {len(c['x'])}\n" + _ints(bits(c["R"])) + _ints(bits(c["x"])) + _ints(bits(c["y"])))
        elif k in ("errors", "chars"):
            s.append(f"{k} {len(c['ops'])}\n" + "".join(str(o) + "\n" for o in c["ops"]))
        else:
            raise ValueError(k)
    return "".join(s)


def parse(text, cs):
    """stdout of ref_dump -> one dict of arrays per case (same order)"""
    lines = text.split("\n")
    at, res = 0, []
    for c in cs:
        head = lines[at].split()
        assert head[0] == c["kind"], (head, c["kind"], c["name"])
        if c["kind"] in ("errors", "chars"):   # text: every line up to "end"
            end = lines.index("end", at)
            res.append(dict(lines=np.array(lines[at + 1:end])))
            at = end + 1
            continue
        n = int(head[1])
        v = np.array([int(x) for x in lines[at + 1:at + 1 + n]], dtype=np.int64)
        at += 1 + n
        k = c["kind"]
        if k == "findIndex":
            res.append(dict(index=v.astype(np.int32)))
        elif k in ("lobatto", "gauss"):
            f = v.astype(np.int32).view(F)
            res.append(dict(mus=f[:c["n"]].copy(), weights=f[c["n"]:].copy()))
        elif k == "legendre":
            res.append(dict(P=v.astype(np.int32).view(F).reshape(len(c["mus"]), c["maxL"] + 1).copy()))   # [mu][l]
        else:
            res.append(dict(refused=np.int32(v[0]), reflectance=v[1:].astype(np.int32).view(F).copy()))
    assert at == len([l for l in lines if l.strip()]), (at, len(lines))
    return res


INPUT_KEYS = {"findIndex": ("table", "values", "guess"), "lobatto": ("n",), "gauss": ("n",), "legendre": ("maxL", "mus"),
              "surface": ("xs", "ys", "R", "x", "y"), "uniform": ("R", "x", "y"), "errors": ("ops",), "chars": ("ops",)}


def save(path, cs, results, header):
    d = {"header": np.array(header), "order": np.array([f"{c['kind']}|{c['name']}" for c in cs])}
    for c, r in zip(cs, results):
        p = f"{c['kind']}|{c['name']}|"
        for k in INPUT_KEYS[c["kind"]]:
            if c.get(k) is not None:
                d[p + "in|" + k] = np.asarray(c[k])
        for k, v in r.items():
            d[p + "out|" + k] = np.asarray(v)
    np.savez_compressed(path, **d)


def load(path):
    """-> (cases, reference outputs) as save() got them"""
    z = np.load(path)
    cs, rs = [], []
    for key in z["order"]:
        kind, name = str(key).split("|", 1)
        c, r = dict(kind=kind, name=name), {}
        p = f"{kind}|{name}|"
        for k in INPUT_KEYS[kind]:
            c[k] = z[p + "in|" + k] if p + "in|" + k in z.files else None
            if k in ("n", "maxL") and c[k] is not None:
                c[k] = int(c[k])
        for f in z.files:
            if f.startswith(p + "out|"):
                r[f[len(p) + 4:]] = z[f]
        cs.append(c); rs.append(r)
    return cs, rs, str(z["header"])
