import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_viterbi_gpu import _run_pruned, _ragged, trained_like_models
D, ns, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sp, A, mu, cv = trained_like_models(W, ns, D, seed=3)
utts = _ragged(64, D, seed=21)
full, dec, batch, pack = _run_pruned(utts, sp, A, mu, cv, tie="high", approx="auto")
asc, aeps, exs, cslot, ccnt = dec.views()
print("T:", [len(u) for u in utts[:8]])
print("asc", asc[:4].cpu().numpy())
print("exact", full.scores[:4].cpu().numpy())
print("eps", aeps[:4].cpu().numpy())
print("eps median", float(aeps.median()), "kept", float((cslot >= 0).double().mean()))
