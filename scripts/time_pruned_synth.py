"""Pruned decoder on synthetic features / trained-like models of a chosen shape, both bounding-pass
implementations (dev tool): python scripts/time_pruned_synth.py D n_states W N"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd import _lib
from sapr_amd.trellis import DiagModelPack, FeatureBatch, PrunedDecoder
from tests._synth import trained_like_models
D, ns, W, N = (int(a) for a in sys.argv[1:5])
sp, A, mu, cv = trained_like_models(W, ns, D, seed=3)
rng = np.random.default_rng(0)
T = 100
# frames drawn around the states of a random word, in order: scores separate the way trained models do
utts = []
for n in range(N):
    w = rng.integers(W)
    st = np.sort(rng.integers(1, ns + 1, T))
    utts.append((mu[w, st] + rng.normal(0, 1, (T, D)) * np.sqrt(cv[w, st])).astype(np.float32))
batch = FeatureBatch.from_arrays(utts, layout="TD")
pack = DiagModelPack.from_params(sp, A, mu, cv)
st_ = _lib.current_stream()
outs = {}
for approx in ("auto", "valu"):
    dec = PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, pack, batch.feats.device, approx=approx)
    for _ in range(2):
        dec.launch(batch.feats, batch.offsets, batch.order, _lib.TIE_HIGH, _lib.SUM_TVIEW, st_)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5):
        dec.launch(batch.feats, batch.offsets, batch.order, _lib.TIE_HIGH, _lib.SUM_TVIEW, st_)
    b.record(); torch.cuda.synchronize()
    asc, aeps, exs, cslot, ccnt = dec.views()
    outs[approx] = (dec.best_word.clone(), dec.best_score.clone(), dec.path.clone())
    print(f"D={D} S={ns + 2} W={W} N={N} approx={approx}: {a.elapsed_time(b) / 5:.3f} ms  candidates/utt "
          f"{int(ccnt.sum()) / N:.3f}  eps median {float(aeps.median()):.3g}  flags {pack.flags}")
print("identical:", all(torch.equal(x, y) for x, y in zip(outs["auto"], outs["valu"])))
