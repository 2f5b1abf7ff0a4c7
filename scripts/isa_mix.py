"""Instruction mix of the hot loop of one kernel in a gfx950 assembly file (dev tool for DESIGN §6's issue-cost model).
    python scripts/isa_mix.py <file.s> <kernel name substring> [--json out.json]
The hot loop = the innermost loop (a label with a backward branch to it) that holds the most instructions."""
import collections
import json
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
txt = open(path, errors="replace").read()
m = None
for mm in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(pat), txt, re.S | re.M):
    if mm.group(1).startswith("_Z"):
        m = mm
        break
if m is None:
    raise SystemExit("kernel not found")
name, lines = m.group(1), m.group(2).split("\n")
ins, labels = [], {}
for ln in lines:
    t = ln.split(";")[0].strip()
    if not t or t.startswith("."):
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
        continue
    if t.endswith(":"):
        labels[t[:-1]] = len(ins)
        continue
    ins.append(t)
loops = []
for i, t in enumerate(ins):
    p = t.split()
    if p[0].startswith(("s_cbranch", "s_branch")) and p[1] in labels and labels[p[1]] <= i:
        loops.append((labels[p[1]], i))
# innermost = loops that contain no other loop; take the largest of those
inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
lo, hi = max(inner, key=lambda l: l[1] - l[0])
body = ins[lo:hi + 1]


def cls(t):
    op = t.split()[0]
    dpp = any(k in t for k in ("row_shr", "row_shl", "row_ror", "row_mirror", "row_half_mirror", "quad_perm", "row_bcast", "row_newbcast"))
    op = re.sub(r"_e(32|64)$", "", op)
    if dpp and not op.endswith("_dpp"):
        op += "_dpp"
    return op


hist = collections.Counter(cls(t) for t in body)
print(f"{name[:90]}\nhot loop: {len(body)} instructions (of {len(ins)})")
valu = sum(v for k, v in hist.items() if k.startswith("v_") and not k.startswith("v_mfma"))
print(f"VALU {valu}  MFMA {sum(v for k, v in hist.items() if k.startswith('v_mfma'))}  "
      f"LDS {sum(v for k, v in hist.items() if k.startswith('ds_'))}  VMEM {sum(v for k, v in hist.items() if k.startswith(('buffer_', 'global_')))}  "
      f"SALU/other {sum(v for k, v in hist.items() if k.startswith('s_'))}")
for k, v in hist.most_common():
    print(f"{v:5d}  {k}")
if "--json" in sys.argv:
    json.dump(dict(hist), open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1, sort_keys=True)
