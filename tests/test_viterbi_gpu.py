"""GPU parity: batched HIP Viterbi (through the C ABI) vs the CPU oracle.

Bar: scores bit-identical (float64 ``==``), state paths identical, arg-max word
identical — the kernel performs the same individually rounded IEEE operations in the
same order as numpy/hmmlearn (oracle/hmmlearn_oracle.py)."""
import numpy as np
import pytest

from tests._synth import synth_batch, synth_utterance, trained_like_models

pytestmark = pytest.mark.gpu


def _ragged(n, D, seed, tmin=1, tmax=120):
    """Utterances as C-contiguous (T,D) float32; tests take ``u.T.copy().T``-style views when they
    need the reference's (D,T)-storage-plus-transposed-view layout."""
    rng = np.random.default_rng(seed)
    utts = []
    for _ in range(n):
        short = tmin < 9 and rng.uniform() < 0.1
        T = int(rng.integers(tmin, 9)) if short else int(rng.integers(max(tmin, 9), tmax))
        proto = rng.normal(0, 20, (8, D))
        if T >= 9:
            x = synth_utterance(rng, proto, T, silence=3 if rng.uniform() < 0.3 else 0).T
        else:
            x = (rng.normal(0, 20, (T, D)) + np.r_[-300, np.zeros(D - 1)]).astype(np.float32)
        utts.append(np.ascontiguousarray(x, dtype=np.float32))
    return utts


def _tview(u):
    """(T,D) transposed view of (D,T) storage — what decoder.py:59 hands to model.decode."""
    return np.ascontiguousarray(u.T).T


def _run_gpu(utts, sp, A, mu, cv, tie, word_sel=None, sum_order=1):
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, viterbi_decode
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    res = viterbi_decode(batch, pack, tie=_lib.TIE_HIGH if tie == "high" else _lib.TIE_LOW,
                         sum_order=sum_order, word_sel=word_sel)
    import torch
    torch.cuda.synchronize()
    return (res.scores.cpu().numpy(), res.best_word.cpu().numpy(), res.best_score.cpu().numpy(),
            res.path.cpu().numpy(), batch, pack)


def _oracle(utts, sp, A, mu, cv, tie, sum_order=1):
    from oracle import c_oracle
    feats = np.concatenate(utts, axis=0)
    offs = np.r_[0, np.cumsum([u.shape[0] for u in utts])].astype(np.int64)
    return c_oracle.decode_batch(feats, offs, sp, A, mu, cv, tie=1 if tie == "high" else 0,
                                 sum_order=sum_order)


@pytest.mark.parametrize("sum_order", [1, 0])
@pytest.mark.parametrize("tie", ["high", "low"])
@pytest.mark.parametrize("D,ns", [(13, 8), (39, 16), (13, 16), (39, 8)])
def test_bidiag_ragged_matches_oracle(D, ns, tie, sum_order):
    from sapr_amd import _lib
    W = 11 if D == 13 else 3
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=3)
    utts = _ragged(700 if D == 13 else 150, D, seed=1)
    sc, bw, bs, path, batch, pack = _run_gpu(utts, sp, A, mu, cv, tie, sum_order=sum_order)
    assert pack.topology == _lib.TOPO_BIDIAG
    osc, obw, opath = _oracle(utts, sp, A, mu, cv, tie, sum_order=sum_order)
    np.testing.assert_array_equal(sc, osc)
    np.testing.assert_array_equal(bw, obw)
    np.testing.assert_array_equal(path, opath)
    np.testing.assert_array_equal(bs, osc[np.arange(len(utts)), obw])


@pytest.mark.parametrize("sum_order", [1, 0])
def test_numpy_oracle_spot_check(sum_order):
    """Same comparison against the readable numpy restatement, which gets its summation order
    from numpy itself: the transposed view of (D,T) storage (decoder.py:59) for sum_order=1,
    a C-contiguous (T,D) array for sum_order=0."""
    from oracle import hmmlearn_oracle as ho
    sp, A, mu, cv = trained_like_models(11, 8, 13, seed=3)
    utts = _ragged(40, 13, seed=7)
    for tie in ("high", "low"):
        sc, bw, bs, path, batch, _ = _run_gpu(utts, sp, A, mu, cv, tie, sum_order=sum_order)
        offs = np.r_[0, np.cumsum([u.shape[0] for u in utts])]
        for u in range(0, 40, 3):
            X = _tview(utts[u]) if sum_order else utts[u]
            for w in range(11):
                lp, st = ho.decode(X, sp[w], A[w], mu[w], cv[w], tie=tie)
                assert lp == sc[u, w]
                if w == bw[u]:
                    np.testing.assert_array_equal(st, path[offs[u]:offs[u + 1]])


@pytest.mark.parametrize("tie", ["high", "low"])
def test_flat_start_identical_states_exact_ties(tie):
    """All states share one Gaussian (flat start, hmmlearn_hmm.py:38-39): every lattice cell
    has mathematically tied predecessors, so paths depend on exact rounding and tie-break."""
    from oracle import hmmlearn_oracle as ho
    from tests._synth import VOCAB, synth_feature_set
    by_word, flat = synth_feature_set(VOCAB[:4], 6, D=13, seed=2)
    sp, A, mu, cv = ho.flat_start(flat, 8)
    spW, AW = sp[None].repeat(2, 0), A[None].repeat(2, 0)
    muW = np.stack([mu, mu + 1.0]).astype(np.float64)
    cvW = np.stack([cv, cv * 1.5]).astype(np.float64)
    utts = [f.T.copy() for f in flat]
    sc, bw, bs, path, _, _ = _run_gpu(utts, spW, AW, muW, cvW, tie)
    osc, obw, opath = _oracle(utts, spW, AW, muW, cvW, tie)
    np.testing.assert_array_equal(sc, osc)
    np.testing.assert_array_equal(bw, obw)
    np.testing.assert_array_equal(path, opath)


@pytest.mark.parametrize("tie", ["high", "low"])
def test_dense_topology(tie):
    from sapr_amd import _lib
    rng = np.random.default_rng(8)
    W, S, D = 3, 10, 13
    sp = rng.dirichlet(np.ones(S), W)
    A = rng.dirichlet(np.ones(S), (W, S))
    A[:, 2, 5] = 0.0  # a structural zero or two
    A /= A.sum(axis=2, keepdims=True)
    _, _, mu, cv = trained_like_models(W, 8, D, seed=4)
    utts = _ragged(200, D, seed=5)
    sc, bw, bs, path, batch, pack = _run_gpu(utts, sp, A, mu, cv, tie)
    assert pack.topology == _lib.TOPO_DENSE
    osc, obw, opath = _oracle(utts, sp, A, mu, cv, tie)
    np.testing.assert_array_equal(sc, osc)
    np.testing.assert_array_equal(bw, obw)
    np.testing.assert_array_equal(path, opath)


def test_word_sel_returns_each_models_path():
    """HMM-per-word decode (model.decode for a chosen model, decoder.py:43)."""
    from oracle import hmmlearn_oracle as ho
    sp, A, mu, cv = trained_like_models(5, 8, 13, seed=6)
    utts = _ragged(30, 13, seed=9, tmin=10)
    sel = np.arange(30) % 5
    sc, bw, bs, path, _, _ = _run_gpu(utts, sp, A, mu, cv, "high", word_sel=sel)
    offs = np.r_[0, np.cumsum([u.shape[0] for u in utts])]
    np.testing.assert_array_equal(bw, sel)
    for u in range(30):
        w = sel[u]
        lp, st = ho.decode(_tview(utts[u]), sp[w], A[w], mu[w], cv[w], tie="high")
        assert lp == bs[u]
        np.testing.assert_array_equal(st, path[offs[u]:offs[u + 1]])


def test_fixed_length_batch_config3_shape():
    """BASELINE config 3 shape at reduced N: T=101, D=13, W=11, 8 emitting states."""
    sp, A, mu, cv = trained_like_models(11, 8, 13, seed=3)
    X = synth_batch(2048, T=101, D=13, seed=0)
    utts = [x for x in X]
    sc, bw, bs, path, _, _ = _run_gpu(utts, sp, A, mu, cv, "high")
    osc, obw, opath = _oracle(utts, sp, A, mu, cv, "high")
    np.testing.assert_array_equal(sc, osc)
    np.testing.assert_array_equal(bw, obw)
    np.testing.assert_array_equal(path, opath)


def test_unsupported_shape_fails_loudly():
    from sapr_amd._lib import SaprHipError
    sp, A, mu, cv = trained_like_models(2, 5, 45, seed=1)      # 45-dim features: wider than the widest kernel
    utts = _ragged(4, 45, seed=1, tmin=10)
    with pytest.raises(SaprHipError):
        _run_gpu(utts, sp, A, mu, cv, "high")
    sp, A, mu, cv = trained_like_models(2, 17, 13, seed=1)     # 19 states: beyond the largest kernel
    with pytest.raises(SaprHipError):
        _run_gpu(_ragged(4, 13, seed=1, tmin=10), sp, A, mu, cv, "high")


@pytest.mark.parametrize("sum_order", [1, 0])
@pytest.mark.parametrize("tie", ["high", "low"])
@pytest.mark.parametrize("D,ns", [(13, 1), (13, 5), (39, 7), (13, 12), (39, 15)])
def test_other_state_counts_are_padded_bit_exactly(D, ns, tie, sum_order):
    """HMMLearnModel(num_states=...) is a free parameter in the reference (hmmlearn_hmm.py:12).  Models
    with fewer states than an instantiated kernel run padded with unreachable states; scores, words
    and paths must not change by a bit."""
    sp, A, mu, cv = trained_like_models(4, ns, D, seed=40 + ns)
    utts = _ragged(120, D, seed=ns, tmin=1, tmax=60)
    sc_g, bw_g, bs_g, path_g, batch, pack = _run_gpu(utts, sp, A, mu, cv, tie, sum_order=sum_order)
    assert pack.S_model == ns + 2 and pack.S in (10, 18) and sc_g.shape == (120, 4)
    sc, bw, path = _oracle(utts, sp, A, mu, cv, tie, sum_order=sum_order)
    np.testing.assert_array_equal(sc_g, sc)
    np.testing.assert_array_equal(bw_g, bw)
    np.testing.assert_array_equal(path_g, path)
    assert path_g.max() < ns + 2


def test_division_variants_bit_identical():
    """The 4-instruction exactly-rounded division (emission.h quad_term<true>) and the IEEE-division
    instantiation must produce the same bits; both must equal the oracle's ``/``."""
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, viterbi_decode
    import torch
    sp, A, mu, cv = trained_like_models(5, 8, 13, seed=21)
    rng = np.random.default_rng(5)
    # awkward variances: tiny, huge-ish, powers of two, values one ulp off a power of two
    cv[0, 1, :] = 2.0 ** rng.integers(-20, 20, 13)
    cv[1, 2, :] = np.nextafter(2.0 ** rng.integers(-8, 8, 13).astype(np.float64), np.inf)
    cv[2, 3, :] = rng.uniform(1e-6, 1e-5, 13)
    cv[3, 4, :] = rng.uniform(1e5, 1e6, 13)
    utts = _ragged(300, 13, seed=77)
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    assert pack.fast_div == 1
    fast = viterbi_decode(batch, pack, fast_div=1)
    ieee = viterbi_decode(batch, pack, fast_div=0)
    torch.cuda.synchronize()
    sc, bw, path = _oracle(utts, sp, A, mu, cv, "high")
    for res in (fast, ieee):
        np.testing.assert_array_equal(res.scores.cpu().numpy(), sc)
        np.testing.assert_array_equal(res.path.cpu().numpy(), path)
        np.testing.assert_array_equal(res.best_word.cpu().numpy(), bw)


@pytest.mark.parametrize("case", ["all_ones_significand", "huge_variance", "tiny_mean"])
def test_models_outside_the_fast_division_domain_use_ieee_division(case):
    """sapr_diag_pack reports fast_div_ok = 0 for operands outside the proven domain; the decode then
    runs the IEEE-division kernels and is still bit-identical to the oracle."""
    from sapr_amd.trellis import DiagModelPack
    sp, A, mu, cv = trained_like_models(3, 8, 13, seed=22)
    if case == "all_ones_significand":
        cv[1, 4, 7] = np.nextafter(32.0, 0.0)
    elif case == "huge_variance":
        cv[2, 5, 0] = 3.0e31
    else:
        mu[0, 3, 2] = 1.0e-40
    assert DiagModelPack.from_params(sp, A, mu, cv).fast_div == 0
    utts = _ragged(64, 13, seed=78)
    sc_g, bw_g, _, path_g, _, pack = _run_gpu(utts, sp, A, mu, cv, "high")
    assert pack.fast_div == 0
    sc, bw, path = _oracle(utts, sp, A, mu, cv, "high")
    np.testing.assert_array_equal(sc_g, sc)
    np.testing.assert_array_equal(path_g, path)
    np.testing.assert_array_equal(bw_g, bw)


# ------------------------------------------------------------------------------------------------------
# pruned decoder (sapr_viterbi_decode_pruned): same best word / score / path bits as the all-vocabulary
# evaluation, and every float32 interval must contain the exact score
# ------------------------------------------------------------------------------------------------------
def _run_pruned(utts, sp, A, mu, cv, tie="high", sum_order=1, approx="auto"):
    import torch
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, PrunedDecoder, viterbi_decode
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    t = _lib.TIE_HIGH if tie == "high" else _lib.TIE_LOW
    full = viterbi_decode(batch, pack, tie=t, sum_order=sum_order)
    dec = PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, pack, batch.feats.device, approx=approx)
    dec.launch(batch.feats, batch.offsets, batch.order, t, sum_order, _lib.current_stream())
    torch.cuda.synchronize()
    return full, dec, batch, pack


def _assert_pruned_equals_full(full, dec):
    import torch
    assert torch.equal(dec.best_word, full.best_word)
    # NaN scores (non-finite features) compare equal bit-wise
    assert torch.equal(dec.best_score.view(torch.int64), full.best_score.view(torch.int64))
    assert torch.equal(dec.path, full.path)
    asc, aeps, exs, cslot, ccnt = dec.views()
    sc = full.scores
    inside = ((asc - sc).abs() <= aeps) | ~torch.isfinite(aeps) | (torch.isinf(sc) & (asc == sc))
    assert bool(inside.all()), "a float32 interval misses the exact score"
    kept = cslot >= 0
    assert torch.equal(exs[kept].view(torch.int64), sc[kept].view(torch.int64))
    assert int(ccnt.sum()) == int(kept.sum())
    return kept


# the bounding pass has two implementations: matrix cores ("auto", where the pack carries PACK_GEMM_OK and
# S <= 16) and vector ALU ("valu"); both must bracket the exact scores and give the same final outputs
APPROX = ["auto", "valu"]


@pytest.mark.parametrize("approx", APPROX)
@pytest.mark.parametrize("tie", ["high", "low"])
@pytest.mark.parametrize("D,ns", [(13, 8), (39, 16), (13, 16), (39, 8)])
def test_pruned_decoder_matches_all_vocabulary_evaluation(D, ns, tie, approx):
    from sapr_amd import _lib
    W = 11 if D == 13 else 4
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=3)
    utts = _ragged(900 if D == 13 else 200, D, seed=21)
    full, dec, batch, pack = _run_pruned(utts, sp, A, mu, cv, tie=tie, approx=approx)
    assert pack.flags & _lib.PACK_GEMM_OK
    kept = _assert_pruned_equals_full(full, dec)
    # ... and against the oracle directly
    osc, obw, opath = _oracle(utts, sp, A, mu, cv, tie)
    np.testing.assert_array_equal(dec.best_word.cpu().numpy(), obw)
    np.testing.assert_array_equal(dec.path.cpu().numpy(), opath)
    np.testing.assert_array_equal(dec.best_score.cpu().numpy(), osc[np.arange(len(utts)), obw])
    assert float(kept.double().mean()) < 0.6   # distinct word models: most words are dropped


@pytest.mark.parametrize("D,ns,W", [(13, 8, 13), (13, 8, 25), (13, 16, 12), (39, 16, 7), (39, 8, 12), (13, 8, 1), (13, 8, 2)])
def test_bounding_pass_vocabularies_of_any_size(D, ns, W):
    """The matrix-core bounding pass packs the states of up to 11 words back to back
    along the MFMA tiles of one pass; larger vocabularies take several passes, the last one partly filled, smaller
    ones leave tiles empty.  Same outputs as the all-vocabulary evaluation, every interval holds."""
    from sapr_amd import _lib
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=17)
    utts = _ragged(150, D, seed=23)
    full, dec, batch, pack = _run_pruned(utts, sp, A, mu, cv)
    assert pack.flags & _lib.PACK_GEMM_OK
    _assert_pruned_equals_full(full, dec)


def test_bounding_pass_in_two_passes_at_39_dimensions(monkeypatch):
    """(39, 18) runs the whole vocabulary in one pass with the feature rows prefetched into LDS; SAPR_BOUND_WC=6 selects
    the other instantiation (two passes of 6 + 5 words, register prefetch).  Same outputs, every interval holds."""
    monkeypatch.setenv("SAPR_BOUND_WC", "6")
    sp, A, mu, cv = trained_like_models(11, 16, 39, seed=19)
    utts = _ragged(120, 39, seed=29)
    full, dec, batch, pack = _run_pruned(utts, sp, A, mu, cv)
    _assert_pruned_equals_full(full, dec)


@pytest.mark.parametrize("approx", APPROX)
def test_pruned_decoder_keeps_every_word_that_ties(approx):
    """Identical word models: every score ties exactly, nothing may be dropped, and the winner is the FIRST
    model (decoder.py:42-47 strict '>'); near-identical ones (1e-9 apart) must be resolved by the exact pass."""
    sp, A, mu, cv = trained_like_models(1, 8, 13, seed=5)
    W = 6
    spW, AW = np.repeat(sp, W, 0), np.repeat(A, W, 0)
    muW, cvW = np.repeat(mu, W, 0).copy(), np.repeat(cv, W, 0).copy()
    muW[4] += 1e-9
    muW[5, :, 3] -= 3e-10
    utts = _ragged(300, 13, seed=4)
    full, dec, _, _ = _run_pruned(utts, spW, AW, muW, cvW, approx=approx)
    kept = _assert_pruned_equals_full(full, dec)
    assert bool(kept.all())
    sc = full.scores.cpu().numpy()
    assert np.array_equal(sc[:, 0], sc[:, 1]) and np.array_equal(sc[:, 0], sc[:, 3])


@pytest.mark.parametrize("approx", APPROX)
def test_pruned_decoder_hard_numerics_and_edge_cases(approx):
    """Large means against small variances (the float32 pass loses digits: wide intervals, still valid),
    non-finite features, one-frame and empty utterances."""
    rng = np.random.default_rng(11)
    W, ns, D = 5, 8, 13
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=9)
    mu[1] *= 50.0                      # |mean| ~ 1.5e4 on c0
    cv[1] *= 1e-3
    mu[2] += 1e5                       # hopeless word: float32 cancellation, interval ~1e3 wide
    cv[3] = rng.uniform(1e-6, 1e-4, cv[3].shape)
    utts = _ragged(200, D, seed=6)
    utts[3] = utts[3].copy()
    utts[3][1, 2] = np.inf
    utts[7] = utts[7].copy()
    utts[7][0, 0] = np.nan
    utts[9] = (utts[9] * 1e30).astype(np.float32)      # squares overflow float32, not float64
    utts[11] = np.zeros((0, D), np.float32)
    utts[12] = utts[12][:1]
    full, dec, _, _ = _run_pruned(utts, sp, A, mu, cv, approx=approx)
    _assert_pruned_equals_full(full, dec)


def test_vector_alu_bound_holds_at_39_dimensions_with_adversarial_magnitudes():
    """The vector-ALU pass accumulates a D-term float32 fma chain per state: its worst-case relative error grows
    with D ((D + 6) u32), so the interval's first term is (D + 7) / 2 * M, not the 10 M of 13 dimensions.  Large
    same-sign per-dimension terms (features far from the means, everything the same order of magnitude) are where
    the chain's roundings line up; every exact score must still sit inside its interval."""
    rng = np.random.default_rng(5)
    W, ns, D = 4, 16, 39
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=13)
    mu[1] = 3.0e3 + rng.uniform(-1, 1, mu[1].shape)            # every dimension ~3e3 away from the features
    cv[1] = rng.uniform(0.9, 1.1, cv[1].shape)
    mu[2] = -7.7e2 + 1e-3 * rng.standard_normal(mu[2].shape)
    cv[2] = rng.uniform(1e-2, 2e-2, cv[2].shape)
    mu[3] *= 1.0 + 2.0 ** -12                                  # means that do not round to float32 exactly
    utts = _ragged(160, D, seed=31, tmin=9)
    for k in range(0, 160, 5):                                  # constant-sign, equal-magnitude frames
        utts[k] = np.full_like(utts[k], np.float32(1.0 / 3.0)) * np.float32(1 + (k % 7))
    full, dec, _, _ = _run_pruned(utts, sp, A, mu, cv, approx="valu")
    _assert_pruned_equals_full(full, dec)
    asc, aeps, _, _, _ = dec.views()
    err = (asc - full.scores).abs()
    assert bool((err <= aeps).all()) and float((err / aeps).max()) < 0.6   # the bound holds with its safety margin


def test_matrix_core_bounding_pass_domain_flag():
    """A non-finite coefficient of the expanded quadratic, or a state without a self-loop at a chain position the
    kernel has no mask for, clears PACK_GEMM_OK; the decoder then bounds on the vector ALU, same outputs.  States without a self-loop are inside at chain positions 0, 4, 8, 12
    (position 0 is the reference's entry state)."""
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack
    sp, A, mu, cv = trained_like_models(3, 8, 13, seed=2)
    assert DiagModelPack.from_params(sp, A, mu, cv).flags & _lib.PACK_GEMM_OK
    utts = _ragged(120, 13, seed=8)
    A2 = A.copy()
    A2[1, 4, 4], A2[1, 4, 5] = 0.0, 1.0         # chain position 4: inside (one lane mask per quarter)
    A2[2, 8, 8], A2[2, 8, 9] = 0.0, 1.0
    assert DiagModelPack.from_params(sp, A2, mu, cv).flags & _lib.PACK_GEMM_OK
    full, dec, _, _ = _run_pruned(utts, sp, A2, mu, cv)
    _assert_pruned_equals_full(full, dec)
    A4 = A.copy()
    A4[2, 7, 7], A4[2, 7, 8] = 0.0, 1.0         # position 7: outside, vector-ALU bounds
    p4 = DiagModelPack.from_params(sp, A4, mu, cv)
    assert p4.prunable and not (p4.flags & _lib.PACK_GEMM_OK)
    full, dec, _, _ = _run_pruned(utts, sp, A4, mu, cv)
    _assert_pruned_equals_full(full, dec)
    cv3 = cv.copy()
    cv3[2, 1, 7] = 1e-19                        # a tiny variance only changes the scaling of the half operands
    p3 = DiagModelPack.from_params(sp, A, mu, cv3)
    assert p3.prunable and (p3.flags & _lib.PACK_GEMM_OK)
    full, dec, _, _ = _run_pruned(utts, sp, A, mu, cv3)
    _assert_pruned_equals_full(full, dec)
    mu5 = mu.copy()
    mu5[0, 3, 2] = np.inf                       # a non-finite coefficient: outside
    p5 = DiagModelPack.from_params(sp, A, mu5, cv)
    assert not (p5.flags & _lib.PACK_GEMM_OK)


def test_pruned_decoder_refuses_models_outside_the_bound_domain():
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, PrunedDecoder, viterbi_decode, viterbi_decode_best
    import torch
    sp, A, mu, cv = trained_like_models(3, 8, 13, seed=2)
    cv[1, 4, 2] = 1e-25
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    assert pack.fast_div and not pack.prunable
    utts = _ragged(50, 13, seed=2)
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    with pytest.raises(_lib.SaprHipError):
        PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, pack, batch.feats.device)
    bw, bs, path = viterbi_decode_best(batch, pack)       # falls back to the all-vocabulary evaluation
    full = viterbi_decode(batch, pack)
    assert torch.equal(bw, full.best_word) and torch.equal(path, full.path)


@pytest.mark.parametrize("D, ns", [(5, 6), (7, 8), (20, 8), (26, 12)])
def test_feature_widths_other_than_13_and_39_run_padded(D, ns):
    """hmmlearn's GaussianHMM takes any n_features (the reference's HMM(num_obs=...) is free as well).  The trellis
    kernels are instantiated for 13 and 39 dimensions: other widths up to 39 run padded with (mean 0, variance 1,
    feature 0) dimensions, each adding +0.0 to the quadratic form (trellis.kernel_dims).  In the decoder's
    left-to-right summation order — and below 8 dimensions, where numpy's pair-wise order is the same loop — scores,
    words and paths keep the oracle's bits; the pair-wise order of wider C-contiguous arrays associates differently
    (checked to 1e-13 relative)."""
    import torch
    from oracle import c_oracle
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, viterbi_decode, viterbi_decode_best
    W = 4
    sp, A, mu, cv = trained_like_models(W, ns, D, seed=41 + D)
    rng = np.random.default_rng(D)
    # (one-frame utterances only below 8 dimensions: numpy reduces a (1, D) row pair-wise in either layout, which for
    # 8 <= D the padded row associates differently)
    lens = rng.integers(1 if D < 8 else 2, 60, 90)
    lens[:3] = 1 if D < 8 else 2
    utts = [(mu[u % W, 1 + (np.arange(t) * ns) // t] + rng.normal(0, 3.0, (t, D))).astype(np.float32) for u, t in enumerate(lens)]
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    assert batch.D in (13, 39) and batch.D_model == D
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    assert pack.D == batch.D and pack.D_model == D
    feats = np.concatenate(utts, axis=0)
    offs = np.r_[0, np.cumsum(lens)].astype(np.int64)
    for sum_order in (1, 0):     # SUM_TVIEW, SUM_PAIRWISE
        osc, obw, opath = c_oracle.decode_batch(feats, offs, sp, A, mu, cv, tie=1, sum_order=sum_order)
        r = viterbi_decode(batch, pack, tie=1, sum_order=sum_order)
        if sum_order == 1 or D < 8:
            assert np.array_equal(r.scores.cpu().numpy(), osc)
            assert np.array_equal(r.best_word.cpu().numpy(), obw) and np.array_equal(r.path.cpu().numpy(), opath)
            bw, bs, path = viterbi_decode_best(batch, pack, tie=1, sum_order=sum_order)   # pruned where the pack allows
            assert np.array_equal(bw.cpu().numpy(), obw) and np.array_equal(path.cpu().numpy(), opath)
            assert np.array_equal(bs.cpu().numpy(), osc[np.arange(len(lens)), obw])
        else:
            np.testing.assert_allclose(r.scores.cpu().numpy(), osc, rtol=1e-13)
    # a model of another width is refused even when both pad to the same kernel width
    other = DiagModelPack.from_params(*trained_like_models(W, ns, D + 1, seed=3))
    if other.D == pack.D:
        with pytest.raises(ValueError):
            viterbi_decode(batch, other)
    torch.cuda.synchronize()
