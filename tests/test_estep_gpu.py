"""GPU parity of the hmmlearn-compat training / scoring path (estep.hip, sapr_amd.hmmlearn_hmm) against
the CPU restatement of hmmlearn 0.3.3 (oracle/hmmlearn_oracle.py — parity with hmmlearn itself is
UNPINNED, see its header).  float64 both sides; device exp/log differ in the last ulps and sums over
utterances run in a different (fixed) order: rtol 1e-9 on statistics, 1e-7 on parameters after EM."""
import pickle

import numpy as np
import pytest

from oracle import c_oracle, hmmlearn_oracle as ho
from tests._synth import VOCAB, synth_feature_set, trained_like_models

pytestmark = pytest.mark.gpu


def _batch(utts_td):
    import torch
    from sapr_amd.trellis import FeatureBatch
    packed = np.ascontiguousarray(np.concatenate(utts_td, axis=0), dtype=np.float32)
    return FeatureBatch.from_packed(torch.from_numpy(packed).cuda(), np.asarray([u.shape[0] for u in utts_td]))


def test_forward_loglik_matches_oracle():
    from sapr_amd.trellis import DiagModelPack, forward_loglik
    sp, A, mu, cv = trained_like_models(4, 8, 13, seed=3)
    by_word, flat = synth_feature_set(VOCAB[:4], 40, D=13, seed=21, tmin=1, tmax=110)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    utt_model = np.repeat(np.arange(4), 40)
    ll = forward_loglik(_batch(utts), DiagModelPack.from_params(sp, A, mu, cv), utt_model).cpu().numpy()
    feats = np.concatenate(utts, axis=0)
    offs = np.r_[0, np.cumsum([u.shape[0] for u in utts])]
    sc, _, _ = c_oracle.decode_batch(feats, offs, sp, A, mu, cv, which=1, sum_order=0)
    ref = sc[np.arange(len(utts)), utt_model]
    np.testing.assert_allclose(ll, ref, rtol=1e-11)
    # numpy restatement on two of them
    for u in (0, 57):
        w = utt_model[u]
        lp, _ = ho.forward_log(sp[w], A[w], ho.log_density_diag(utts[u], mu[w], cv[w]))
        assert abs(lp - ll[u]) <= 1e-11 * abs(lp)


@pytest.mark.parametrize("topology", ["bidiag", "dense"])
def test_estep_statistics_match_oracle(topology):
    from sapr_amd.trellis import DiagModelPack, EStep, split_stats
    W, S, D = 3, 10, 13
    sp, A, mu, cv = trained_like_models(W, 8, D, seed=5)
    if topology == "dense":
        rng = np.random.default_rng(1)
        A = rng.dirichlet(np.ones(S), (W, S))
        sp = rng.dirichlet(np.ones(S), W)
    by_word, flat = synth_feature_set(VOCAB[:W], 9, D=D, seed=8, tmin=1, tmax=70)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    utt_model = np.repeat(np.arange(W), 9)
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    es = EStep(_batch(utts), utt_model, W, S)
    stats = es.run(pack).cpu().numpy()
    ll = es.loglik.cpu().numpy()
    for w in range(W):
        ref = ho.new_stats(S, D)
        lps = [ho.accumulate(ref, utts[u], sp[w], A[w], mu[w], cv[w]) for u in range(len(utts)) if utt_model[u] == w]
        got = split_stats(stats[w], S, D)
        assert got["nobs"] == ref["nobs"] == 9
        np.testing.assert_allclose(got["logprob"], sum(lps), rtol=1e-11)
        np.testing.assert_allclose(ll[utt_model == w], lps, rtol=1e-11)
        for k_ref, k_got in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                             ("obs2", "obs**2")):
            np.testing.assert_allclose(got[k_got], ref[k_ref], rtol=1e-9, atol=1e-9, err_msg=k_ref)


@pytest.mark.parametrize("D,ns", [(13, 8), (39, 8), (13, 16), (39, 16)])
def test_estep_every_instantiated_shape_multi_tile(D, ns):
    """All four (D, S) instantiations; word 0 spans two 256-utterance tiles (one of them ragged), word 1
    has a handful of utterances: exercises the in-tile lane reduction, the tile-level and the
    word-level reductions of fb_obs / fb_tile_reduce / fb_reduce."""
    from sapr_amd.trellis import DiagModelPack, EStep, split_stats
    W, S = 2, ns + 2
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=15)
    counts = (270, 5)
    utts, utt_model = [], []
    for w, n in enumerate(counts):
        _, flat = synth_feature_set([VOCAB[w]], n, D=D, seed=30 + w, tmin=1, tmax=24)
        utts += [np.ascontiguousarray(f.T) for f in flat]
        utt_model += [w] * n
    utt_model = np.asarray(utt_model)
    # interleave so the TileLayout has to gather each word's utterances
    perm = np.random.default_rng(0).permutation(len(utts))
    utts, utt_model = [utts[i] for i in perm], utt_model[perm]
    es = EStep(_batch(utts), utt_model, W, S)
    stats = es.run(DiagModelPack.from_params(sp, A, mu, cv)).cpu().numpy()
    for w in range(W):
        ref = ho.new_stats(S, D)
        lps = [ho.accumulate(ref, utts[u], sp[w], A[w], mu[w], cv[w]) for u in range(len(utts)) if utt_model[u] == w]
        got = split_stats(stats[w], S, D)
        assert got["nobs"] == counts[w]
        np.testing.assert_allclose(got["logprob"], sum(lps), rtol=1e-11)
        for k_ref, k_got in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                             ("obs2", "obs**2")):
            np.testing.assert_allclose(got[k_got], ref[k_ref], rtol=1e-9, atol=1e-9, err_msg=k_ref)


@pytest.mark.parametrize("D,ns", [(13, 5), (39, 12)])
def test_estep_and_fit_with_padded_state_counts(D, ns):
    """num_states other than 8 / 16: the kernels run on models padded with unreachable states
    (trellis.kernel_states); statistics, log-likelihoods and the EM trajectory are those of the
    unpadded numpy restatement."""
    from sapr_amd.hmmlearn_hmm import GaussianHMM
    from sapr_amd.trellis import DiagModelPack, EStep
    S = ns + 2
    by_word, flat = synth_feature_set(VOCAB[:2], 7, D=D, seed=17, tmin=1, tmax=50)
    sp, A, mu, cv = trained_like_models(2, ns, D, seed=19)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    utt_model = np.repeat(np.arange(2), 7)
    es = EStep(_batch(utts), utt_model, 2, S)
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    assert pack.S_model == S and es.S == pack.S and pack.S in (10, 18)
    stats = es.run(pack).cpu().numpy()
    for w in range(2):
        ref = ho.new_stats(S, D)
        lps = [ho.accumulate(ref, utts[u], sp[w], A[w], mu[w], cv[w]) for u in range(14) if utt_model[u] == w]
        got = es.split(stats[w])
        assert got["trans"].shape == (S, S) and got["obs"].shape == (S, D)
        np.testing.assert_allclose(got["logprob"], sum(lps), rtol=1e-11)
        for k_ref, k_got in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                             ("obs2", "obs**2")):
            np.testing.assert_allclose(got[k_got], ref[k_ref], rtol=1e-9, atol=1e-9, err_msg=k_ref)
    # GaussianHMM.fit / score / decode on such a model
    fsp, fA, fmu, fcv = ho.flat_start(flat, ns)
    X = np.concatenate([f.T for f in by_word[VOCAB[0]]], axis=0)
    lengths = [f.shape[1] for f in by_word[VOCAB[0]]]
    m = GaussianHMM(n_components=S, covariance_type="diag", n_iter=3, params="stmc", implementation="log",
                    min_covar=0.01, init_params="")
    m.means_, m.covars_, m.transmat_, m.startprob_ = fmu, fcv, fA, fsp
    m.fit(X, lengths)
    rsp, rA, rmu, rcv, hist = ho.fit(X, lengths, fsp, fA, fmu.astype(np.float64), fcv.astype(np.float64), n_iter=3)
    np.testing.assert_allclose(list(m.monitor_.history), hist, rtol=1e-9)
    np.testing.assert_allclose(m.means_, rmu, rtol=1e-7, atol=1e-9)
    assert m.means_.shape == (S, D) and m.transmat_.shape == (S, S)
    lp, st = m.decode(flat[0].T)
    rlp, rst = ho.decode(flat[0].T, m.startprob_, m.transmat_, m.means_, m._covars_, tie="high")
    assert lp == rlp
    np.testing.assert_array_equal(st, rst)


def test_estep_is_deterministic():
    from sapr_amd.trellis import DiagModelPack, EStep
    sp, A, mu, cv = trained_like_models(2, 8, 13, seed=5)
    _, flat = synth_feature_set(VOCAB[:2], 300, D=13, seed=3)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    es = EStep(_batch(utts), np.repeat(np.arange(2), 300), 2, 10)
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    a = es.run(pack).cpu().numpy().copy()
    b = es.run(pack).cpu().numpy().copy()
    np.testing.assert_array_equal(a, b)  # fixed-order reductions, no atomics


def test_exact_only_pack_serves_training_and_scoring_and_is_refused_by_the_pruned_decoder():
    """SAPR_PACK_EXACT_ONLY (what a Baum-Welch loop packs per iteration): the E-step statistics, the forward scores and
    the all-vocabulary Viterbi are the full pack's bit for bit; the pruned decoder, whose bounding-pass operands were
    left out, does not accept it."""
    from sapr_amd import _lib
    from sapr_amd.trellis import (DiagModelPack, EStep, PrunedDecoder, forward_loglik, viterbi_decode,
                                  viterbi_decode_best)
    sp, A, mu, cv = trained_like_models(3, 8, 13, seed=9)
    _, flat = synth_feature_set(VOCAB[:3], 120, D=13, seed=4)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    utt_model = np.repeat(np.arange(3), 120)
    batch = _batch(utts)
    full = DiagModelPack.from_params(sp, A, mu, cv)
    lean = DiagModelPack.from_params(sp, A, mu, cv, exact_only=True)
    assert full.prunable and not lean.prunable
    assert lean.flags & _lib.PACK_EXACT_ONLY and not lean.flags & (_lib.PACK_BOUND_OK | _lib.PACK_GEMM_OK)
    assert lean.fast_div == full.fast_div
    es = EStep(batch, utt_model, 3, 10)
    np.testing.assert_array_equal(es.run(full).cpu().numpy(), es.run(lean).cpu().numpy())
    np.testing.assert_array_equal(forward_loglik(batch, full, utt_model).cpu().numpy(),
                                  forward_loglik(batch, lean, utt_model).cpu().numpy())
    a, b = viterbi_decode(batch, full), viterbi_decode(batch, lean)
    np.testing.assert_array_equal(a.scores.cpu().numpy(), b.scores.cpu().numpy())
    np.testing.assert_array_equal(a.path.cpu().numpy(), b.path.cpu().numpy())
    with pytest.raises(_lib.SaprHipError):
        PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, lean, batch.feats.device)
    # the best-word entry point falls back to the all-vocabulary evaluation: the same answer
    wf, sf, pf = viterbi_decode_best(batch, full)
    wl, sl, pl = viterbi_decode_best(batch, lean)
    np.testing.assert_array_equal(wf.cpu().numpy(), wl.cpu().numpy())
    np.testing.assert_array_equal(sf.cpu().numpy(), sl.cpu().numpy())
    np.testing.assert_array_equal(pf.cpu().numpy(), pl.cpu().numpy())


def test_gaussian_hmm_fit_score_decode_and_pickle():
    from sapr_amd.hmmlearn_hmm import GaussianHMM
    by_word, flat = synth_feature_set(VOCAB[:3], 8, D=13, seed=11)
    sp, A, mu, cv = ho.flat_start(flat, 8)
    X = np.concatenate([f.T for f in by_word["heed"]], axis=0)
    lengths = [f.shape[1] for f in by_word["heed"]]
    m = GaussianHMM(n_components=10, covariance_type="diag", n_iter=5, params="stmc", implementation="log",
                    min_covar=0.01, init_params="")
    m.means_, m.covars_, m.transmat_, m.startprob_ = mu, cv, A, sp
    assert m.fit(X, lengths) is m
    rsp, rA, rmu, rcv, hist = ho.fit(X, lengths, sp, A, mu.astype(np.float64), cv.astype(np.float64), n_iter=5)
    np.testing.assert_allclose(list(m.monitor_.history), hist, rtol=1e-9)
    assert len(m.monitor_.history) == len(hist)
    np.testing.assert_allclose(m.transmat_, rA, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(m.means_, rmu, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m._covars_, rcv, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(m.startprob_, rsp, atol=1e-12)
    assert m.covars_.shape == (10, 13, 13)  # hmmlearn exposes full matrices
    # score = sum of forward log-probs (hmmlearn_hmm.py:104)
    np.testing.assert_allclose(m.score(X, lengths), ho.score(X, lengths, rsp, rA, m.means_, m._covars_), rtol=1e-10)
    # decode on the decoder.py:59 view and on a contiguous copy: numpy's two summation orders
    f = by_word["heed"][0]
    for Xd in (f.T, np.ascontiguousarray(f.T)):
        lp, st = m.decode(Xd)
        rlp, rst = ho.decode(Xd, m.startprob_, m.transmat_, m.means_, m._covars_, tie="high")
        assert lp == rlp
        np.testing.assert_array_equal(st, rst)
    # pickles without device state and still works (train.py:74-78 → decoder.py:26-27)
    m2 = pickle.loads(pickle.dumps(m))
    assert m2.decode(f.T)[0] == m.decode(f.T)[0]
    assert list(m2.monitor_.history) == list(m.monitor_.history)


def _np_mean_var_sequential_f32(X):
    """np.mean / np.var over axis 0 of a float32 (N, D) array with the row-after-row float32 accumulation numpy
    performs on the build container's CPU (there ``np.sum(X, axis=0) == np.cumsum(X, axis=0)[-1]`` bit for bit).
    numpy's order on this axis is NOT the same on every CPU — on the GPU box's EPYC ``np.sum`` differs from the
    sequential chain in the last bits — so the kernel is compared with the chain stated explicitly."""
    n = X.shape[0]
    s = np.cumsum(X, axis=0, dtype=np.float32)[-1]
    mean = np.true_divide(s, n, out=s, casting="unsafe")
    d = X - mean
    q = np.cumsum(d * d, axis=0, dtype=np.float32)[-1]
    return mean, np.true_divide(q, n, out=q, casting="unsafe")


def test_hmmlearn_model_wrapper_and_decoder_end_to_end(tmp_path, monkeypatch, capsys):
    """train.py:114-120 → decoder.py flow on a synthetic feature_set directory."""
    from sapr_amd.decoder import Decoder
    from sapr_amd.hmmlearn_hmm import HMMLearnModel
    words = VOCAB[:4]
    by_word, flat = synth_feature_set(words, 6, D=13, seed=4)
    fs = tmp_path / "feature_set"
    fs.mkdir()
    k = 0
    for w in words:
        for x in by_word[w]:
            np.save(fs / f"s{k:03d}_{w}.npy", x)
            k += 1
    monkeypatch.chdir(tmp_path)
    models = {}
    for w in words:
        h = HMMLearnModel(num_states=8, model_name=w, n_iter=4, min_covar=0.01)
        assert "Self-transition probability" in capsys.readouterr().out
        assert h.model.means_.shape == (10, 13) and h.model.transmat_.shape == (10, 10)
        # flat start on the GPU == numpy's float32 np.mean / np.var over the concatenated frames, bit for bit
        # (hmmlearn_hmm.py:83-94); os.listdir order decides the concatenation order, so rebuild it the same way
        X = np.concatenate([f.T for f in h.all_features], axis=0)
        assert X.dtype == np.float32 and h.global_mean.dtype == np.float32 and h.global_cov.dtype == np.float32
        m_seq, v_seq = _np_mean_var_sequential_f32(X)
        np.testing.assert_array_equal(h.global_mean, m_seq)
        np.testing.assert_array_equal(h.global_cov, v_seq)
        # numpy's own float32 order on this host may differ (and vary with buffer alignment): sanity only
        np.testing.assert_allclose(h.global_mean, np.mean(X, axis=0), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(h.global_cov, np.var(X, axis=0), rtol=1e-3)
        trained, ll = h.fit(by_word[w])
        assert trained is h.model and np.isfinite(ll) and 1 <= len(h.model.monitor_.history) <= 4
        hist = list(h.model.monitor_.history)
        assert all(b >= a - 1e-6 for a, b in zip(hist, hist[1:]))
        d = tmp_path / "trained_models" / "hmmlearn"
        d.mkdir(parents=True, exist_ok=True)
        with open(d / f"{w}_hmmlearn_4.pkl", "wb") as f:
            pickle.dump(trained, f)
        models[w] = trained
    dec = Decoder(implementation="hmmlearn", n_iter=4)
    res = dec.decode_vocabulary("feature_set", verbose=False)
    capsys.readouterr()
    res_v = dec.decode_vocabulary("feature_set", verbose=True)     # one batch for the whole vocabulary
    text = capsys.readouterr().out
    assert text.count("Results for '") == len(words) and text.count("\nSample ") == 24 and "Accuracy: " in text
    for w in words:
        assert [r["predicted_word"] for r in res_v[w]] == [r["predicted_word"] for r in res[w]]
        assert [r["sample_index"] for r in res[w]] == list(range(1, 7))
        for a, b in zip(dec.decode_word_samples(w, "feature_set"), res[w]):   # per-word call == slice of the batch
            assert a["log_likelihood"] == b["log_likelihood"] and a["predicted_word"] == b["predicted_word"]
            np.testing.assert_array_equal(a["state_sequence"], b["state_sequence"])
    n_ok = 0
    for w in words:
        feats = by_word[w]
        # the reference's os.listdir order is arbitrary: match results to utterances by score
        for r in res[w]:
            assert r["true_word"] == w and len(r["state_sequence"]) > 0
            n_ok += r["correct"]
    assert n_ok >= 20  # well-separated synthetic words
    # per-utterance oracle check of decode_sequence semantics
    order = list(dec.vocab)
    x = by_word[words[1]][2]
    word, score, states = dec.decode_sequence(x.T)
    best, bw, bst = -np.inf, None, None
    for w in order:
        m = dec.models[w]
        lp, st = ho.decode(x.T, m.startprob_, m.transmat_, m.means_, m._covars_, tie="high")
        if lp > best:
            best, bw, bst = lp, w, st
    assert word == bw and score == best
    np.testing.assert_array_equal(states, bst)
    # eval.py's harness (metrics only) on top of the batched decoder
    from sapr_amd.eval import eval_hmm
    ev = eval_hmm("hmmlearn", "feature_set", model_iter=4)
    assert ev["accuracy"] == pytest.approx(n_ok / 24) and len(ev["true_labels"]) == 24
    assert list(ev["confusion_matrix"].index) == list(dec.vocab) and int(ev["confusion_matrix"].values.sum()) == 24
    assert int(np.trace(ev["confusion_matrix"].values)) == n_ok
    # packed store path (store.FeatureStore): same answers as the per-file path, and fit_models can
    # train from it
    from sapr_amd.store import FeatureStore
    st = FeatureStore.pack_directory("feature_set", str(tmp_path / "store"))
    per_file = dec.decode_batch([st.utterance(i) for i in range(len(st))])
    packed = dec.decode_store(st)
    assert len(packed) == len(st) == 24
    for a, b in zip(per_file, packed):
        assert a[0] == b[0] and a[1] == b[1]
        np.testing.assert_array_equal(a[2], b[2])
    from sapr_amd.hmmlearn_hmm import fit_models
    fresh = [HMMLearnModel(num_states=8, model_name=w, n_iter=4, min_covar=0.01) for w in words]
    fit_models([h.model for h in fresh], st.training_data(words))
    for h, w in zip(fresh, words):
        np.testing.assert_allclose(h.model.means_, models[w].means_, rtol=1e-9, atol=1e-9)
        assert list(h.model.monitor_.history) == pytest.approx(list(models[w].monitor_.history), rel=1e-9)


def test_flat_start_column_sums_are_sequential_float32_chains():
    """sapr_colsum_f32 at a size where float32 accumulation visibly loses digits (300 k frames): one sequential
    float32 chain per coefficient, which is what np.mean / np.var over axis 0 of a float32 array compute on the
    reference's (and the golden build's) CPU."""
    from sapr_amd.hmmlearn_hmm import HMMLearnModel
    rng = np.random.default_rng(3)
    feats = [(rng.normal(0, 20, (13, int(t))) - np.r_[300, np.zeros(12)][:, None]).astype(np.float32)
             for t in rng.integers(900, 1100, 300)]
    X = np.concatenate([f.T for f in feats], axis=0)
    sums, n = HMMLearnModel._column_sums(feats)
    assert n == X.shape[0]
    np.testing.assert_array_equal(sums, np.cumsum(X, axis=0, dtype=np.float32)[-1])
    assert not np.array_equal(sums, np.sum(X.astype(np.float64), axis=0).astype(np.float32))   # it IS the float32 chain
    mean, var = _np_mean_var_sequential_f32(X)
    sq, _ = HMMLearnModel._column_sums(feats, center=mean)
    np.testing.assert_array_equal(np.true_divide(sq, n, out=sq, casting="unsafe"), var)
    np.testing.assert_allclose(var, np.var(X.astype(np.float64), axis=0), rtol=2e-2)   # float32 chain: sanity only
    # 39-dimensional features (tile geometry changes)
    f39 = [rng.normal(0, 5, (39, 57)).astype(np.float32) for _ in range(40)]
    X39 = np.concatenate([f.T for f in f39], axis=0)
    s39, _ = HMMLearnModel._column_sums(f39)
    np.testing.assert_array_equal(s39, np.cumsum(X39, axis=0, dtype=np.float32)[-1])


SPLIT_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
from tests._synth import VOCAB, synth_feature_set, trained_like_models
W, S, D = 3, 10, 13
sp, A, mu, cv = trained_like_models(W, 8, D, seed=5)
_, flat = synth_feature_set(VOCAB[:W], 120, D=D, seed=8, tmin=1, tmax=70)
utts = [np.ascontiguousarray(f.T) for f in flat]
packed = np.ascontiguousarray(np.concatenate(utts, axis=0), dtype=np.float32)
batch = FeatureBatch.from_packed(torch.from_numpy(packed).cuda(), np.asarray([u.shape[0] for u in utts]))
es = EStep(batch, np.repeat(np.arange(W), 120), W, S)
pack = DiagModelPack.from_params(sp, A, mu, cv)
a = es.run(pack).cpu().numpy().copy()
b = es.run(pack).cpu().numpy().copy()      # second call: staged features
assert np.array_equal(a, b)
np.savez(sys.argv[2], stats=a, loglik=es.loglik.cpu().numpy())
print("ok")
'''


def test_fused_second_pass_equals_the_split_pair(tmp_path):
    """Round 4's default (smoothing recursion + observation sums in one pass on the float64 matrix cores,
    fb_smooth_obs_kernel) against round 3's pair (posterior lattice written by the smoothing pass, fb_obs_kernel over it:
    SAPR_ESTEP_OBS=split, read once per process): the same statistics to the re-association of the sums."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(SPLIT_WORKER)
    got = {}
    for mode in ("fused", "split"):
        env = dict(os.environ)
        env.pop("SAPR_ESTEP_OBS", None)
        if mode == "split":
            env["SAPR_ESTEP_OBS"] = "split"
        out = str(tmp_path / f"{mode}.npz")
        p = subprocess.run([sys.executable, str(script), root, out], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ok" in p.stdout, (p.stdout + p.stderr)[-3000:]
        got[mode] = dict(np.load(out))
    np.testing.assert_array_equal(got["fused"]["loglik"], got["split"]["loglik"])
    np.testing.assert_allclose(got["fused"]["stats"], got["split"]["stats"], rtol=1e-12, atol=1e-10)


def test_fit_and_score_with_26_dimensional_features():
    """n_features = 26 (MFCC + delta): the E-step, fit, score and decode run on the 39-wide kernels with zero columns
    (trellis.kernel_dims); statistics, log-likelihoods and the EM trajectory are those of the numpy restatement on
    the 26 real dimensions."""
    from sapr_amd.hmmlearn_hmm import GaussianHMM
    from sapr_amd.trellis import DiagModelPack, EStep
    D, ns, S = 26, 8, 10
    by_word, flat = synth_feature_set(VOCAB[:2], 7, D=D, seed=27, tmin=1, tmax=50)
    sp, A, mu, cv = trained_like_models(2, ns, D, seed=29)
    utts = [np.ascontiguousarray(f.T) for f in flat]
    utt_model = np.repeat(np.arange(2), 7)
    es = EStep(_batch(utts), utt_model, 2, S)
    assert es.D == 39 and es.batch.D_model == 26
    stats = es.run(DiagModelPack.from_params(sp, A, mu, cv)).cpu().numpy()
    for w in range(2):
        ref = ho.new_stats(S, D)
        lps = [ho.accumulate(ref, utts[u], sp[w], A[w], mu[w], cv[w]) for u in range(len(utts)) if utt_model[u] == w]
        got = es.split(stats[w])
        assert got["obs"].shape == (S, D)
        np.testing.assert_allclose(got["logprob"], sum(lps), rtol=1e-11)
        for k_ref, k_got in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                             ("obs2", "obs**2")):
            np.testing.assert_allclose(got[k_got], ref[k_ref], rtol=1e-9, atol=1e-9, err_msg=k_ref)
    X = np.concatenate(utts[:7], axis=0)
    lengths = [u.shape[0] for u in utts[:7]]
    m = GaussianHMM(n_components=S, covariance_type="diag", n_iter=4, params="stmc", implementation="log",
                    min_covar=0.01, init_params="")
    m.means_, m.covars_, m.transmat_, m.startprob_ = mu[0].copy(), cv[0].copy(), A[0].copy(), sp[0].copy()
    m.fit(X, lengths)
    rsp, rA, rmu, rcv, hist = ho.fit(X, lengths, sp[0], A[0], mu[0], cv[0], n_iter=4)
    np.testing.assert_allclose(list(m.monitor_.history), hist, rtol=1e-9)
    assert m.means_.shape == (S, D) and m._covars_.shape == (S, D)
    np.testing.assert_allclose(m.means_, rmu, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m._covars_, rcv, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(m.score(X, lengths), ho.score(X, lengths, rsp, rA, m.means_, m._covars_), rtol=1e-10)
    # decode on the decoder.py:59 view: the left-to-right summation order keeps the oracle's bits when padded
    Xd = _tview26(utts[0])
    lp, st = m.decode(Xd)
    rlp, rst = ho.decode(Xd, m.startprob_, m.transmat_, m.means_, m._covars_, tie="high")
    assert lp == rlp and np.array_equal(st, rst)


def _tview26(u):
    return np.ascontiguousarray(u.T).T


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_estep_random_ragged_layouts_match_oracle(seed):
    """Random vocabularies, uneven utterance counts per word (quarter-tiles with 1..64 live slots, empty tails) and
    lengths from 1 frame up: the per-wavefront partial rows of the fused second pass must add up to the oracle's
    statistics whatever the tile layout."""
    from sapr_amd.trellis import DiagModelPack, EStep
    rng = np.random.default_rng(seed)
    W, D, ns = int(rng.integers(1, 5)), 13, 8
    S = ns + 2
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=seed)
    counts = [int(rng.choice([1, 3, 63, 64, 65, 130, 257])) for _ in range(W)]
    utts, utt_model = [], []
    for w, n in enumerate(counts):
        _, flat = synth_feature_set([VOCAB[w]], n, D=D, seed=seed + w, tmin=1, tmax=int(rng.integers(3, 40)))
        utts += [np.ascontiguousarray(f.T) for f in flat]
        utt_model += [w] * n
    perm = rng.permutation(len(utts))
    utts, utt_model = [utts[i] for i in perm], np.asarray(utt_model)[perm]
    es = EStep(_batch(utts), utt_model, W, S)
    stats = es.run(DiagModelPack.from_params(sp, A, mu, cv)).cpu().numpy()
    for w in range(W):
        ref = ho.new_stats(S, D)
        lps = [ho.accumulate(ref, utts[u], sp[w], A[w], mu[w], cv[w]) for u in range(len(utts)) if utt_model[u] == w]
        got = es.split(stats[w])
        assert got["nobs"] == counts[w]
        np.testing.assert_allclose(got["logprob"], sum(lps), rtol=1e-11)
        for k_ref, k_got in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                             ("obs2", "obs**2")):
            np.testing.assert_allclose(got[k_got], ref[k_ref], rtol=1e-9, atol=1e-9, err_msg=k_ref)
