"""Instruction census of one kernel of the device assembly (hipcc -S): per basic block the number of vector, scalar,
LDS and memory instructions, so that the voxel-step and event phases can be priced without a GPU.
  python3 tools/asm_blocks.py /tmp/i3rc.s 'photon_kernelINS_12PhiloxStreamELb0ELb0ELi0' [--dump BLOCK]"""
import re
import sys
from collections import Counter


def kernel_lines(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
    ls = kernel_lines(path, pat)
    blocks, cur = [], None
    for l in ls:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = dict(name=m.group(1) if m else "entry", v=0, s=0, ds=0, mem=0, depth=0, ops=Counter(), text=[])
            d = re.findall(r"Depth=(\d+)", l)
            cur["depth"] = max([int(x) for x in d] + [0])
            blocks.append(cur)
            if m:
                continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            d = re.findall(r"Depth=(\d+)", l)
            if d and cur["v"] + cur["s"] == 0:
                cur["depth"] = max(cur["depth"], max(int(x) for x in d))
            continue
        op = t.split()[0]
        cur["text"].append(t)
        cur["ops"][op] += 1
        if op.startswith("v_"):
            cur["v"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            cur["mem"] += 1
    tot = Counter()
    for b in blocks:
        if dump and b["name"] == dump:
            print("\n".join(b["text"]))
        if not dump and b["v"] + b["s"] + b["ds"] + b["mem"] > 0:
            print(f"{b['name']:12s} depth {b['depth']} v {b['v']:4d} s {b['s']:4d} ds {b['ds']:3d} mem {b['mem']:3d}")
        tot.update(b["ops"])
    if not dump:
        print("total v", sum(b["v"] for b in blocks), "s", sum(b["s"] for b in blocks))
        print("top ops:", tot.most_common(40))


main()
