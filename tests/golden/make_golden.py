#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's ``custom_hmm.py``.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

Inputs are regenerated from seeds by ``tests/_synth.py``; this script commits the
reference's OUTPUTS (float64 ``.npz``) for

  G0  dataset-independent known answers quoted from the reference's own
      ``pytest_results/*.txt`` (flat start, T = 47, a_ii = 0.8392062244694939)
  G1  flat-start init (global mean / covariance / A)         custom_hmm.py:35-116
  G2  compute_emission_matrix                                custom_hmm.py:146-174
  G3  forward, backward, compute_gamma, compute_xi, seq LL   custom_hmm.py:176-322
  G4  baum_welch LL history + parameters after 1..3 iters    custom_hmm.py:402-460
  G5  decode (13-frame quirk; 16-state/13-dim unreachable exit; 16-state/39-dim)
                                                             custom_hmm.py:462-514
  G6  Decoder.decode_sequence semantics (strict '>' arg-max over word models in
      load order, decoder.py:35-49) evaluated with the reference HMM objects

``hmmlearn_hmm.py`` / ``mfcc_extract.py`` cannot be run (hmmlearn / librosa absent);
their oracles stay "parity unpinned" (oracle/__init__.py).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/assignment2")

from custom_hmm import HMM  # noqa: E402  (the reference itself)

from tests._synth import VOCAB, synth_feature_set  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def pack_model(h):
    return dict(A=h.A.copy(), mean=h.B["mean"].copy(), cov=h.B["covariance"].copy(),
                gmean=h.global_mean.copy(), gcov=h.global_covariance.copy())


def main():
    out = {}
    by_word, flat = synth_feature_set(VOCAB, 6, D=13, seed=0)

    # ---------------------------------------------------------------- G1 flat start
    h = HMM(8, 13, flat)
    m0 = pack_model(h)
    for k, v in m0.items():
        out[f"g1_{k}"] = v

    # ------------------------------------------- G2/G3 on three utterances, 3 models
    probe = [by_word["heed"][0], by_word["heed"][2], by_word["hood"][1]]  # [2] has silence ties
    stages = {}
    hh = HMM(8, 13, flat, model_name="heed")
    stages[0] = pack_model(hh)
    for it in (1, 2):
        hh2 = HMM(8, 13, flat, model_name="heed")
        with np.errstate(all="ignore"):
            quiet(hh2.baum_welch, by_word["heed"], it + 1)  # it M-steps happen within it+1 iterations
        stages[it] = pack_model(hh2)
    for it, m in stages.items():
        hx = HMM(8, 13, flat)
        hx.A, hx.B = m["A"].copy(), {"mean": m["mean"].copy(), "covariance": m["cov"].copy()}
        for k, v in m.items():
            out[f"g2_s{it}_{k}"] = v
        for u, f in enumerate(probe):
            with np.errstate(all="ignore"):
                E = hx.compute_emission_matrix(f)
                al, sc = hx.forward(E)
                be = hx.backward(E, sc)
                ga = hx.compute_gamma(al, be)
                xi = hx.compute_xi(al, be, E)
                ll = np.logaddexp.reduce(al[-1])
            out[f"g2_s{it}_u{u}_E"] = E
            out[f"g3_s{it}_u{u}_alpha"] = al
            out[f"g3_s{it}_u{u}_scale"] = np.float64(sc)
            out[f"g3_s{it}_u{u}_beta"] = be
            out[f"g3_s{it}_u{u}_gamma"] = ga
            out[f"g3_s{it}_u{u}_xi"] = xi
            out[f"g3_s{it}_u{u}_ll"] = np.float64(ll)

    # --------------------------------------------------------------- G4 Baum-Welch
    for n_it in (1, 2, 3, 4):
        hb = HMM(8, 13, flat, model_name="heed")
        with np.errstate(all="ignore"):
            hist = quiet(hb.baum_welch, by_word["heed"], n_it)
        out[f"g4_it{n_it}_hist"] = np.asarray(hist, dtype=np.float64)
        for k, v in pack_model(hb).items():
            out[f"g4_it{n_it}_{k}"] = v

    # -------------------------------------------------------------------- G5 decode
    # eleven word models trained 2 iterations each (train.py:106-113 flow)
    models = []
    for w in VOCAB:
        hw = HMM(8, 13, flat, model_name=w)
        with np.errstate(all="ignore"):
            quiet(hw.baum_welch, by_word[w], 2)
        models.append(hw)
        for k, v in pack_model(hw).items():
            out[f"g5_model_{w}_{k}"] = v
    scores = np.empty((len(flat), len(VOCAB)))
    paths = np.empty((len(flat), len(VOCAB), 13), dtype=np.int64)
    for u, f in enumerate(flat):
        for w, hw in enumerate(models):
            with np.errstate(all="ignore"):
                lp, p = hw.decode(f)
            scores[u, w] = lp
            paths[u, w] = np.asarray(p, dtype=np.int64)
    out["g5_scores"] = scores
    out["g5_paths"] = paths
    # flat-start decode (all states identical → pervasive ties)
    fs_scores = np.empty(len(flat))
    fs_paths = np.empty((len(flat), 13), dtype=np.int64)
    for u, f in enumerate(flat):
        lp, p = h.decode(f)
        fs_scores[u] = lp
        fs_paths[u] = np.asarray(p, dtype=np.int64)
    out["g5_flat_scores"] = fs_scores
    out["g5_flat_paths"] = fs_paths

    # G6: decoder.py:35-49 semantics over the reference models (first strict maximum wins)
    best = np.empty(len(flat), dtype=np.int64)
    for u in range(len(flat)):
        b, bw = float("-inf"), -1
        for w in range(len(VOCAB)):
            if scores[u, w] > b:
                b, bw = scores[u, w], w
        best[u] = bw
    out["g6_best_word"] = best

    # 16 emitting states, D = 13: exit needs t >= 16 but the trellis has 13 rows → -inf
    h16 = HMM(16, 13, flat)
    lp, p = h16.decode(flat[0])
    out["g5_s16_d13_score"] = np.float64(lp)
    out["g5_s16_d13_path"] = np.asarray(p, dtype=np.int64)
    # 16 emitting states, D = 39
    by39, flat39 = synth_feature_set(VOCAB[:3], 4, D=39, seed=5)
    h39 = HMM(16, 39, flat39, model_name="heed")
    for k, v in pack_model(h39).items():
        out[f"g5_s16_d39_init_{k}"] = v
    with np.errstate(all="ignore"):
        hist39 = quiet(h39.baum_welch, by39["heed"], 2)
    out["g5_s16_d39_hist"] = np.asarray(hist39)
    for k, v in pack_model(h39).items():
        out[f"g5_s16_d39_{k}"] = v
    sc39 = np.empty(len(flat39))
    pa39 = np.empty((len(flat39), 39), dtype=np.int64)
    for u, f in enumerate(flat39):
        with np.errstate(all="ignore"):
            lp, p = h39.decode(f)
        sc39[u] = lp
        pa39[u] = np.asarray(p, dtype=np.int64)
    out["g5_s16_d39_scores"] = sc39
    out["g5_s16_d39_paths"] = pa39
    E39 = h39.compute_emission_matrix(flat39[1])
    al39, s39 = h39.forward(E39)
    out["g5_s16_d39_E"] = E39
    out["g5_s16_d39_alpha"] = al39
    out["g5_s16_d39_scale"] = np.float64(s39)

    # ------------------------------ G0: known answers from the reference's own logs
    # pytest_results/training_results.txt:12,41,46 and forward_backward_results.txt:229-232
    out["g0_aii"] = np.float64(0.8392062244694939)
    out["g0_T"] = np.int64(47)
    out["g0_gamma2"] = np.array([0.0, 0.84090909, 0.15909091])
    out["g0_gamma23"] = np.array([0.00445046, 0.04283564, 0.15874384, 0.29397007,
                                  0.29397007, 0.15874384, 0.04283564, 0.00445046])
    out["g0_xi_1_1_1"] = np.float64(37.0 / 44.0)

    path = os.path.join(HERE, "custom_hmm_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    main()
