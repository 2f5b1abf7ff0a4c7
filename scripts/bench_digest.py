"""Print the key numbers of a bench.py JSON line (dev tool): python scripts/bench_digest.py <log>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'{d["value"]:.4g} {d["unit"]}  {d["ms_per_step"]:.3f} ms/step  kernels {r["kernel_ms"]}  hbm frac {r["frac"]:.3f}  flops frac {r.get("flops_frac", 0):.3f}  '
      f'exact lattices/utt {r.get("exact_lattices_per_utterance")}')
for k, v in (d.get("extra") or {}).items():
    print(" ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "workload"})
c, ca = d.get("cpu_baseline"), d.get("cpu_baseline_all_cores")
if c:
    print("  cpu 1 core", round(c["value"]), "paths identical", c["viterbi_paths_identical_on_sample"], "mfcc diff", c["mfcc_max_abs_diff_on_sample"])
if ca:
    print("  cpu all cores", ca.get("value"), "workers", ca.get("cores"))
