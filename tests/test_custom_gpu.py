"""GPU parity of the custom-HMM mirror (sapr_amd.custom_hmm.HMM, kernels in custom.hip) against the
golden vectors produced by the IMPORTED reference (tests/golden/make_golden.py) and the oracle.

Floating point: float64 on both sides.  The emission matrix (``compute_emission_matrix``) and ``decode``
are evaluated in the reference's own operation order and must be BIT-IDENTICAL to the goldens (flat-start
ties included); device exp/log1p differ from numpy's in the last ulps and the batched E-step uses the
algebraically equal row-sum emission, so the recurrences and Baum-Welch are compared at rtol 1e-9
(absolute 1e-9 near zero)."""
import numpy as np
import pytest

from tests._synth import VOCAB, synth_feature_set

pytestmark = pytest.mark.gpu
RT = dict(rtol=1e-9, atol=1e-9)


def _close(a, b, **kw):
    kw = {**RT, **kw}
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    np.testing.assert_array_equal(np.isneginf(a), np.isneginf(b))
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    m = np.isfinite(b)
    np.testing.assert_allclose(a[m], b[m], **kw)


def _hmm(golden, prefix, flat, n_states=8, D=13):
    from sapr_amd.custom_hmm import HMM
    h = HMM(n_states, D)
    h.A = golden[f"{prefix}_A"].copy()
    h.B = {"mean": golden[f"{prefix}_mean"].copy(), "covariance": golden[f"{prefix}_cov"].copy()}
    h.global_mean = golden[f"{prefix}_gmean"].copy()
    h.global_covariance = golden[f"{prefix}_gcov"].copy()
    return h


def test_g1_flat_start_on_gpu(golden, feature_set):
    from sapr_amd.custom_hmm import HMM
    _, flat = feature_set
    h = HMM(8, 13, flat)
    np.testing.assert_array_equal(h.global_mean, golden["g1_gmean"])  # float32 pair-wise row sums reproduced
    np.testing.assert_allclose(h.global_covariance, golden["g1_gcov"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(h.A, golden["g1_A"])
    np.testing.assert_array_equal(h.B["mean"], golden["g1_mean"])
    np.testing.assert_allclose(h.B["covariance"], golden["g1_cov"], rtol=1e-12, atol=1e-12)
    assert h.pi[0] == 1.0 and h.pi[1:].sum() == 0.0 and h.total_states == 10


@pytest.mark.parametrize("n_utts, D", [(1, 13), (471, 13), (945, 13), (2501, 13), (700, 39)])
def test_global_mean_list_order_fold_is_bit_exact_over_many_tiles(n_utts, D):
    """custom_hmm.py:70-80 adds the per-utterance float32 row sums one utterance after another in float64; the device
    fold stages the rows through LDS tile by tile (472 rows of 13 sums per tile) and must reproduce that chain bit for
    bit at every tile count and remainder — with values whose low bits make every addition round."""
    from oracle import custom_hmm_oracle as co
    from sapr_amd.custom_hmm import HMM
    rng = np.random.default_rng(n_utts)
    feats = []
    for u in range(n_utts):
        T = int(rng.integers(9, 40))
        scale = 10.0 ** rng.integers(-6, 4)          # row sums from 1e-6 to 1e4: the chain rounds all the time
        feats.append((scale * rng.standard_normal((D, T))).astype(np.float32))
    h = HMM(8, D, feats)
    np.testing.assert_array_equal(h.global_mean, co.global_mean(feats))


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_g2_g3_per_method_api(golden, feature_set, stage):
    by_word, flat = feature_set
    probe = [by_word["heed"][0], by_word["heed"][2], by_word["hood"][1]]
    h = _hmm(golden, f"g2_s{stage}", flat)
    from oracle import custom_hmm_oracle as co
    for u, f in enumerate(probe):
        E = h.compute_emission_matrix(f)
        Eg = golden[f"g2_s{stage}_u{u}_E"]
        # bit-identical to the oracle's explicit evaluation order (pinned to the goldens on the CPU,
        # tests/test_oracle_custom.py) — and therefore to the reference itself wherever this host's LAPACK
        # returns the golden build's inverse (always at flat start: diagonal covariances)
        np.testing.assert_array_equal(E, co.emission(f, h.B["mean"], h.B["covariance"], gram="chain"))
        if stage == 0:
            np.testing.assert_array_equal(E, Eg)
        _close(E, Eg, rtol=1e-12)
        al, sc = h.forward(Eg)
        _close(al, golden[f"g3_s{stage}_u{u}_alpha"])
        _close(sc, golden[f"g3_s{stage}_u{u}_scale"])
        be = h.backward(Eg, golden[f"g3_s{stage}_u{u}_scale"])
        _close(be, golden[f"g3_s{stage}_u{u}_beta"])
        ga = h.compute_gamma(golden[f"g3_s{stage}_u{u}_alpha"], golden[f"g3_s{stage}_u{u}_beta"])
        _close(ga, golden[f"g3_s{stage}_u{u}_gamma"])
        xi = h.compute_xi(golden[f"g3_s{stage}_u{u}_alpha"], golden[f"g3_s{stage}_u{u}_beta"], Eg)
        _close(xi, golden[f"g3_s{stage}_u{u}_xi"])


def test_reference_test_suite_properties(feature_set):
    """The properties the reference's own tests assert (tests/test_foward_backward.py:17-132,
    tests/test_training.py:111-275), on the mirror."""
    from sapr_amd.custom_hmm import HMM
    by_word, flat = feature_set
    h = HMM(8, 13, flat)
    f = flat[0]
    B = h.compute_emission_matrix(f)
    assert B.shape == (f.shape[1], 10)  # (the reference's `B <= 0` holds on its dataset, not in general: SURVEY §8a)
    assert np.all(B[:, 0] == -np.inf) and np.all(B[:, -1] == -np.inf)
    assert np.std(B[0, 1:-1]) < 1e-10
    al, sc = h.forward(B)
    be = h.backward(B, sc)
    assert np.all(al[1:, 0] == -np.inf) and np.all(be[:-1, -1] == -np.inf) and be[-1, -1] == 0
    assert np.all(al[al != -np.inf] <= 0)
    ga = h.compute_gamma(al, be)
    xi = h.compute_xi(al, be, B)
    assert ga.shape == (f.shape[1], 10) and xi.shape == (f.shape[1] - 1, 10, 10)
    assert np.all((ga >= 0) & (ga <= 1)) and np.all((xi >= 0) & (xi <= 1))
    np.testing.assert_allclose(ga.sum(axis=1), 1.0, atol=1e-9)
    # update_A keeps the left-right structure, rows sum to one (test_training.py:136-203)
    h.update_A(xi.sum(axis=0), ga[:-1].sum(axis=0))
    assert h.A[0, 1] == 1.0 and h.A[-1, -1] == 1.0
    np.testing.assert_allclose(h.A[1:-1].sum(axis=1), 1.0, atol=1e-10)
    # update_B: shapes, zero entry/exit rows, symmetric, floored diagonal (test_training.py:206-275)
    feats = by_word["heed"]
    gammas = []
    for x in feats:
        e = h.compute_emission_matrix(x)
        a, s = h.forward(e)
        gammas.append(h.compute_gamma(a, h.backward(e, s)))
    h.update_B(feats, gammas)
    assert h.B["mean"].shape == (10, 13) and h.B["covariance"].shape == (10, 13, 13)
    assert np.all(h.B["mean"][[0, -1]] == 0) and np.all(h.B["covariance"][[0, -1]] == 0)
    floor = h.var_floor_factor * np.mean(np.diagonal(h.global_covariance))
    for j in range(1, 9):
        c = h.B["covariance"][j]
        np.testing.assert_allclose(c, c.T, atol=1e-12)
        assert np.all(np.diag(c) >= floor - 1e-15) and np.all(np.linalg.eigvalsh(c) > -1e-10)


def test_stage_features_and_weighted_moments_through_the_c_abi():
    """sapr_custom_stage_features: feat_t[t][d][slot] is the transpose of the ragged frame-major features, zero past
    each utterance.  sapr_custom_update_b_moments: row s of its output holds sum g x'x'^T (upper triangle), sum g x'
    and sum g over all frames, x' = x - centre, for the emitting states — against numpy on ragged utterances with a
    random posterior lattice in the slot layout."""
    import ctypes as C
    import torch
    from sapr_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    D, S, lens = 13, 10, np.array([5, 17, 1, 33, 12, 64, 2, 40, 9, 21, 7, 3, 28, 16, 11, 6, 19, 50])
    N, max_T = len(lens), int(lens.max())
    offs = np.r_[0, np.cumsum(lens)].astype(np.int64)
    x = (rng.standard_normal((int(lens.sum()), D)) * 5 + 100).astype(np.float32)
    slots = 64
    dev = torch.device("cuda", 0)
    feats, offsets = torch.from_numpy(x).to(dev), torch.from_numpy(offs).to(dev)
    st = _lib.current_stream()
    feat_t = torch.full((max_T * D * slots,), float("nan"), dtype=torch.float32, device=dev)
    fsum = torch.full((D * slots,), float("nan"), dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_custom_stage_features(_lib.ptr(feats), _lib.ptr(offsets), N, D, max_T, slots, _lib.ptr(feat_t),
                                              _lib.ptr(fsum), st), "sapr_custom_stage_features")
    got = feat_t.cpu().numpy().reshape(max_T, D, slots)
    want = np.zeros((max_T, D, slots), np.float32)
    for u in range(N):
        want[:lens[u], :, u] = x[offs[u]:offs[u + 1]]
    np.testing.assert_array_equal(got, want)
    # frame_sums[d][slot]: the utterance's frames added one after another in float64 (zero for the empty slots)
    want_sum = np.zeros((D, slots))
    for u in range(N):
        acc = np.zeros(D)
        for row in x[offs[u]:offs[u + 1]]:
            acc += row.astype(np.float64)
        want_sum[:, u] = acc
    np.testing.assert_array_equal(fsum.cpu().numpy().reshape(D, slots), want_sum)

    gamma = np.zeros((max_T, S, slots))
    for u in range(N):
        g = rng.random((lens[u], S))
        g[:, 0] = g[:, -1] = 0.0  # non-emitting entry / exit states
        gamma[:lens[u], :, u] = g / g.sum(axis=1, keepdims=True)
    centre = x.astype(np.float64).mean(axis=0)
    nb = C.c_size_t(0)
    _lib.check(lib.sapr_custom_update_b_workspace_bytes(N, 1, D, S, C.byref(nb)), "workspace_bytes")
    ws = torch.empty(int(nb.value), dtype=torch.uint8, device=dev)
    out = torch.zeros(16 * 112, dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_custom_update_b_moments(_lib.ptr(feats), _lib.ptr(offsets), N, D, S,
                                                _lib.ptr(torch.from_numpy(gamma.reshape(-1)).to(dev)), slots,
                                                _lib.ptr(torch.from_numpy(centre).to(dev)), _lib.ptr(out), _lib.ptr(ws),
                                                int(nb.value), st), "sapr_custom_update_b_moments")
    m = out.cpu().numpy().reshape(16, 112)
    iu = np.triu_indices(D)
    for s in range(16):
        if 1 <= s <= S - 2:
            occ, s1, s2 = 0.0, np.zeros(D), np.zeros((D, D))
            for u in range(N):
                xc = x[offs[u]:offs[u + 1]].astype(np.float64) - centre
                g = gamma[:lens[u], s, u]
                occ += g.sum()
                s1 += g @ xc
                s2 += (xc * g[:, None]).T @ xc
            np.testing.assert_allclose(m[s, 104], occ, rtol=1e-12)
            np.testing.assert_allclose(m[s, 91:104], s1, rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(m[s, :91], s2[iu], rtol=1e-10, atol=1e-8)
        else:
            assert not m[s].any()


@pytest.mark.parametrize("ns,D", [(8, 13), (16, 39), (8, 39), (16, 13)])
def test_batched_estep_staged_unstaged_and_reference_order_agree(ns, D):
    """The batched E-step of every instantiated shape through the C ABI, two word models interleaved utterance by
    utterance (the lanes of a wavefront hold different models) on ragged utterances: reading the features from the
    slot-major copy or from the frame-major array gives the same bits (the same operations), and both agree with the
    run-time-shaped kernel in the reference's row layout and operation order at 1e-9 — per-utterance log-likelihood,
    scale, posterior sums, transition sums and the posterior lattice itself."""
    import torch
    from sapr_amd import _lib
    from sapr_amd.custom_hmm import HMM, model_arrays
    lib = _lib.load()
    S = ns + 2
    by_word, flat = synth_feature_set(VOCAB[:2], 45, D=D, seed=31, tmin=ns + 4, tmax=70)
    rng = np.random.default_rng(5)
    models = []
    for w in VOCAB[:2]:
        h = HMM(ns, D, by_word[w], w)
        # distinct states: means spread around the global mean, full covariances scaled per state
        for j in range(1, S - 1):
            h.B["mean"][j] = h.global_mean + rng.normal(0, 1.0, D) * np.sqrt(np.diag(h.global_covariance))
            a = rng.normal(0, 0.05, (D, D))
            h.B["covariance"][j] = h.global_covariance * (0.6 + 0.1 * j) + a @ a.T
        models.append(h)
    order = [by_word[VOCAB[k % 2]][k // 2] for k in range(90)]  # word 0, word 1, word 0, ...
    utt_model = np.arange(90, dtype=np.int32) % 2
    lens = np.array([f.shape[1] for f in order])
    N, max_T, total = len(order), int(lens.max()), int(lens.sum())
    offs = np.r_[0, np.cumsum(lens)].astype(np.int64)
    x = np.ascontiguousarray(np.concatenate([f.T for f in order], axis=0), dtype=np.float32)
    dev = torch.device("cuda", 0)
    feats, offsets, um = (torch.from_numpy(a).to(dev) for a in (x, offs, utt_model))
    arrs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in model_arrays(models)]
    st = _lib.current_stream()
    K = 2 + S + S * S

    def lattices(n):
        return [torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for _ in range(4)]

    # reference order, row layout [total_frames][S]
    E0, a0, b0, g0 = lattices(total * S)
    out0 = torch.zeros(N * K, dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_custom_estep(_lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(um), N, D, S, 2,
                                     *[_lib.ptr(a) for a in arrs], 0, _lib.ptr(E0), _lib.ptr(a0), _lib.ptr(b0),
                                     _lib.ptr(g0), None, _lib.ptr(out0), st), "sapr_custom_estep")
    slots = 128
    outs, gammas = [], []
    feat_t = torch.empty(max_T * D * slots, dtype=torch.float32, device=dev)
    fsum = torch.empty(D * slots, dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_custom_stage_features(_lib.ptr(feats), _lib.ptr(offsets), N, D, max_T, slots, _lib.ptr(feat_t),
                                              _lib.ptr(fsum), st), "sapr_custom_stage_features")
    # features from the frame-major array / from the slot-major copy / with the precomputed frame sums as well:
    # the same operations in the same order, identical bits
    for staged in (0, 1, 2):
        E1, a1, b1, g1 = lattices(max_T * S * slots)
        out1 = torch.zeros(N * K, dtype=torch.float64, device=dev)
        _lib.check(lib.sapr_custom_estep_staged(_lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(um), N, D, S, 2,
                                                *[_lib.ptr(a) for a in arrs], slots, _lib.ptr(E1), _lib.ptr(a1),
                                                _lib.ptr(b1), _lib.ptr(g1), None, _lib.ptr(out1),
                                                _lib.ptr(feat_t) if staged else None,
                                                _lib.ptr(fsum) if staged == 2 else None, st), "sapr_custom_estep_staged")
        outs.append(out1.cpu().numpy().reshape(N, K))
        gammas.append(g1.cpu().numpy().reshape(max_T, S, slots))
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])
    np.testing.assert_array_equal(gammas[1][~np.isnan(gammas[1])], gammas[2][~np.isnan(gammas[2])])
    ref = out0.cpu().numpy().reshape(N, K)
    gref = g0.cpu().numpy().reshape(total, S)
    assert np.isfinite(ref[:, 0]).all()
    _close(outs[1], ref)
    for u in range(N):
        for g in gammas:
            _close(g[:lens[u], :, u], gref[offs[u]:offs[u + 1]])
    np.testing.assert_array_equal(gammas[0][~np.isnan(gammas[0])], gammas[1][~np.isnan(gammas[1])])


def test_update_b_moments_match_the_two_pass_reference_order(feature_set, monkeypatch):
    """Default training path: update_B from one pass of posterior-weighted moments about the global mean
    (sapr_custom_update_b_moments, float64 matrix cores), E-step on the staged slot-major features with the
    smoothing / reference-order split.  SAPR_CUSTOM_FOLD=ordered keeps custom_hmm.py:366-400's two passes and the
    list-order sums: both must train the same model (the goldens pin each of them separately)."""
    from sapr_amd.custom_hmm import HMM
    by_word, flat = feature_set
    out = {}
    for mode in ("", "ordered"):
        if mode:
            monkeypatch.setenv("SAPR_CUSTOM_FOLD", mode)
        else:
            monkeypatch.delenv("SAPR_CUSTOM_FOLD", raising=False)
        h = HMM(8, 13, flat, model_name="heed")
        with np.errstate(all="ignore"):
            hist = h.baum_welch(by_word["heed"], 3)
        out[mode] = (np.asarray(hist), h.A.copy(), h.B["mean"].copy(), h.B["covariance"].copy())
    for a, b, rtol, atol in zip(out[""], out["ordered"], (1e-9, 1e-8, 1e-8, 1e-7), (0, 1e-12, 1e-10, 1e-9)):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_update_b_tight_state_far_from_the_global_mean_takes_the_two_pass_route(monkeypatch):
    """The one-pass covariance S2/occ - d d^T about the global mean cancels ~log10(d^2 / var) digits.  A state with
    sigma = 0.01 sitting 500 away from the centre (d^2 / var = 2.5e9) would keep 6 of 16: update_B detects it from the
    moments and runs custom_hmm.py:366-400's own two passes instead.  Ground truth: the two passes in numpy float64."""
    from sapr_amd.custom_hmm import HMM
    rng = np.random.default_rng(11)
    n_utts, T, D, n_states = 12, 48, 13, 8
    seg = np.minimum(np.arange(T) * n_states // T, n_states - 1) + 1        # emitting state of every frame
    mu = rng.normal(0, 5, (n_states + 2, D))
    sd = np.full(n_states + 2, 3.0)
    mu[4] += 500.0
    sd[4] = 0.01
    feats, gammas = [], []
    for _ in range(n_utts):
        x = (mu[seg] + sd[seg, None] * rng.standard_normal((T, D))).astype(np.float32)
        g = np.zeros((T, n_states + 2))
        g[np.arange(T), seg] = 1.0            # hard posteriors: the tight cluster must not mix with its neighbours
        soft = (seg != 4) & (np.minimum(seg + 1, n_states) != 4) & (seg < n_states)
        g[np.arange(T)[soft], seg[soft]] = 0.75                               # elsewhere 3/4 on the segment's state,
        g[np.arange(T)[soft], seg[soft] + 1] = 0.25                           # 1/4 on its right neighbour
        feats.append(np.ascontiguousarray(x.T))
        gammas.append(g)
    X = np.concatenate([f.T for f in feats]).astype(np.float64)
    G = np.concatenate(gammas)
    occ = G.sum(0)
    want_mean = np.zeros((n_states + 2, D))
    want_cov = np.zeros((n_states + 2, D, D))
    for j in range(1, n_states + 1):
        want_mean[j] = G[:, j] @ X / occ[j]
        dx = X - want_mean[j]
        want_cov[j] = (G[:, j, None] * dx).T @ dx / occ[j]
    got = {}
    for mode in ("", "ordered", "moments"):
        if mode:
            monkeypatch.setenv("SAPR_CUSTOM_FOLD", mode)
        else:
            monkeypatch.delenv("SAPR_CUSTOM_FOLD", raising=False)
        h = HMM(n_states, D, feats, model_name="tight", var_floor_factor=1e-12)
        h.update_B(feats, gammas)
        got[mode] = (h.B["mean"].copy(), h.B["covariance"].copy())
    for mode in ("", "ordered"):
        np.testing.assert_allclose(got[mode][0][1:-1], want_mean[1:-1], rtol=1e-12, atol=1e-12, err_msg=mode)
        # (atol: 1e-7 of the tight state's variance 1e-4 — its small off-diagonal entries carry the rounding of the mean)
        np.testing.assert_allclose(got[mode][1][1:-1], want_cov[1:-1], rtol=1e-8, atol=1e-11, err_msg=mode)
    # what the guard prevents: the forced one-pass form is off by far more than that on the tight state ...
    err = np.abs(got["moments"][1][4] - want_cov[4]).max() / np.abs(want_cov[4]).max()
    assert err > 1e-9, err
    # ... while the well-conditioned states agree in every mode
    for j in (1, 2, 3, 5, 6, 7, 8):
        np.testing.assert_allclose(got["moments"][1][j], want_cov[j], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n_it", [1, 2, 3])
def test_g4_baum_welch(golden, feature_set, n_it, capsys):
    from sapr_amd.custom_hmm import HMM
    by_word, flat = feature_set
    h = HMM(8, 13, flat, model_name="heed")
    with np.errstate(all="ignore"):
        hist = h.baum_welch(by_word["heed"], n_it)
    assert "Training complete!" in capsys.readouterr().out
    np.testing.assert_allclose(hist, golden[f"g4_it{n_it}_hist"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(h.A, golden[f"g4_it{n_it}_A"], rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(h.B["mean"], golden[f"g4_it{n_it}_mean"], rtol=1e-7, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(h.B["covariance"], golden[f"g4_it{n_it}_cov"], rtol=1e-6, atol=1e-8, equal_nan=True)


def test_g5_decode_paths_identical(golden, feature_set):
    from sapr_amd.custom_hmm import decode_batch
    _, flat = feature_set
    from oracle import custom_hmm_oracle as co
    models = [_hmm(golden, f"g5_model_{w}", flat) for w in VOCAB]
    scores, paths, bw, bs, bp = decode_batch(models, flat, with_best=True)
    np.testing.assert_array_equal(paths, golden["g5_paths"])
    _close(scores, golden["g5_scores"], rtol=1e-12)
    # bit-identical to the oracle's explicit evaluation order on this host (same LAPACK inverse)
    for u in (0, 7, 65):
        for w in (0, 5, 10):
            with np.errstate(all="ignore"):
                lp, p = co.decode(flat[u], models[w].A, models[w].B["mean"], models[w].B["covariance"], 8, gram="chain")
            np.testing.assert_equal(scores[u, w], lp)
            assert list(paths[u, w]) == p
    # decoder.py:42-47 arg-max (first strict maximum in model order), evaluated on the device
    np.testing.assert_array_equal(bw, golden["g6_best_word"])
    for u in range(len(flat)):
        assert bs[u] == scores[u, bw[u]] and list(bp[u]) == list(paths[u, bw[u]])


def test_g5_flat_start_ties_and_quirk(golden, feature_set):
    """tests/test_decode.py:32-38: the path has features.shape[0] (= 13) entries.

    At flat start every state has the same Gaussian, so all left-to-right paths into a cell tie
    mathematically and the reference's choice hangs on the last bit of its emission values.  The kernel
    evaluates them in the reference's order (custom.hip, pinned in tests/test_oracle_custom.py), so paths
    AND scores are the reference's, bit for bit, for all 66 utterances."""
    from sapr_amd.custom_hmm import decode_batch
    _, flat = feature_set
    h = _hmm(golden, "g1", flat)
    for u, f in enumerate(flat[:4]):  # the per-utterance API
        lp, p = h.decode(f)
        assert len(p) == f.shape[0] == 13 and isinstance(lp, float) and all(isinstance(x, int) for x in p)
        assert p == list(golden["g5_flat_paths"][u]) and lp == golden["g5_flat_scores"][u]
    scores, paths = decode_batch([h], flat)
    np.testing.assert_array_equal(paths[:, 0, :], golden["g5_flat_paths"])
    np.testing.assert_array_equal(scores[:, 0], golden["g5_flat_scores"])


def test_g5_sixteen_states(golden, feature_set):
    from sapr_amd.custom_hmm import HMM
    _, flat = feature_set
    h16 = HMM(16, 13, flat)
    lp, p = h16.decode(flat[0])
    assert lp == -np.inf and p == list(golden["g5_s16_d13_path"])
    by39, flat39 = synth_feature_set(VOCAB[:3], 4, D=39, seed=5)
    h39 = _hmm(golden, "g5_s16_d39", flat39, n_states=16, D=39)
    from oracle import custom_hmm_oracle as co
    for u, f in enumerate(flat39):
        lp, p = h39.decode(f)
        assert p == list(golden["g5_s16_d39_paths"][u]) and len(p) == 39
        # 39-dim full covariances estimated from 4 utterances are nearly singular (|E| ~ 1e10): the last
        # bits of LAPACK's inverse matter, so the golden comparison is loose and the exact one is against
        # the oracle's chain order evaluated with this host's inverse
        _close(lp, golden["g5_s16_d39_scores"][u], rtol=1e-6)
        with np.errstate(all="ignore"):
            olp, op = co.decode(f, h39.A, h39.B["mean"], h39.B["covariance"], 16, gram="chain")
        assert lp == olp and p == op
    E = h39.compute_emission_matrix(flat39[1])
    np.testing.assert_array_equal(E, co.emission(flat39[1], h39.B["mean"], h39.B["covariance"], gram="chain"))
    _close(E, golden["g5_s16_d39_E"], rtol=1e-6)


@pytest.mark.parametrize("n_states,D", [(8, 13), (16, 13), (8, 39), (16, 39)])
def test_emission_rows_of_every_leaf_shape_keep_the_chain_order_bits(golden, feature_set, n_states, D):
    """custom_emission_bcast_kernel (13 / 39 dimensions, one row per lane, differences broadcast over the DPP row):
    utterances shorter than one numpy block, with every remainder of a block, exactly one leaf (128), and the
    recursive halving above it (129 .. 700 frames: up to three levels) — all frames through compute_emission_matrix,
    the first D rows of a mixed batch through decode — against the oracle's explicit evaluation order, bit for bit."""
    from sapr_amd.custom_hmm import HMM, decode_batch
    from oracle import custom_hmm_oracle as co
    rng = np.random.default_rng(100 + n_states + D)
    S = n_states + 2
    h = HMM(n_states, D)
    A = np.zeros((S, S))
    A[0, 1] = 1.0
    for i in range(1, S - 1):
        A[i, i], A[i, i + 1] = 0.8, 0.2
    A[S - 1, S - 1] = 1.0
    h.A = A
    mean = rng.normal(0, 4, (S, D))
    cov = np.stack([np.cov(rng.normal(0, 3, (D, 50))) + np.eye(D) for _ in range(S)])
    mean[[0, -1]] = 0
    cov[[0, -1]] = 0
    h.B = {"mean": mean, "covariance": cov}
    lens = [1, 2, 7, 8, 9, 13, 15, 16, 23, 100, 127, 128, 129, 136, 257, 700]
    feats = [(5 * rng.standard_normal((D, T))).astype(np.float32) for T in lens]
    for f in feats:
        E = h.compute_emission_matrix(f)
        np.testing.assert_array_equal(E, co.emission(f, mean, cov, gram="chain"))
    batch = [f for f in feats if f.shape[1] >= D]
    h2 = HMM(n_states, D)
    h2.A = A
    h2.B = {"mean": mean + 1.0, "covariance": cov}
    scores, paths = decode_batch([h, h2, h], batch)
    for u, f in enumerate(batch):
        for w, m in enumerate((h, h2, h)):
            with np.errstate(all="ignore"):
                lp, pth = co.decode(f, m.A, m.B["mean"], m.B["covariance"], n_states, gram="chain")
            np.testing.assert_equal(scores[u, w], lp)
            assert list(paths[u, w]) == pth


DECODE_WORKER = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from sapr_amd.custom_hmm import HMM, decode_batch
D, n_states, n_utts, W = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), 5
rng = np.random.default_rng(2024 + D)
S = n_states + 2
models = []
for w in range(W):
    h = HMM(n_states, D)
    A = np.zeros((S, S)); A[0, 1] = 1.0
    for i in range(1, S - 1):
        A[i, i], A[i, i + 1] = 0.7 + 0.02 * w, 0.3 - 0.02 * w
    A[S - 1, S - 1] = 1.0
    h.A = A
    mean = rng.normal(0, 4, (S, D)); mean[[0, -1]] = 0
    cov = np.stack([np.cov(rng.normal(0, 3, (D, 3 * D))) + np.eye(D) for _ in range(S)]); cov[[0, -1]] = 0
    h.B = {"mean": mean, "covariance": cov}
    models.append(h)
lens = rng.integers(D, 4 * D + 140, n_utts)
lens[:4] = [D, 127, 128, 129 + D]
feats = [(5 * rng.standard_normal((D, int(T)))).astype(np.float32) for T in lens]
sc, paths, bw, bs, bp = decode_batch(models, feats, with_best=True)
np.savez(sys.argv[2], sc=sc, paths=paths, bw=bw, bs=bs, bp=bp)
print("ok")
"""


@pytest.mark.parametrize("D,n_states,n_utts", [(13, 8, 1500), (13, 16, 300), (39, 8, 200)])
def test_decode_kernels_agree_bit_for_bit_with_the_generic_ones(tmp_path, D, n_states, n_utts):
    """The DPP-row emission kernel and the register-resident left-to-right trellis (defaults) against the
    one-thread-per-row emission kernel and the generic trellis (SAPR_CUSTOM_EMISSION_ROWS1=1 /
    SAPR_CUSTOM_DECODE_GENERIC=1, read once per process): ragged utterances from D frames (the shortest decode
    accepts) up to several leaves of the pair-wise sum, five models — scores, paths and the decoder's arg-max
    identical to the last bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(DECODE_WORKER)
    got = {}
    for mode in ("default", "generic"):
        env = dict(os.environ)
        for k in ("SAPR_CUSTOM_EMISSION_ROWS1", "SAPR_CUSTOM_DECODE_GENERIC"):
            env.pop(k, None)
            if mode == "generic":
                env[k] = "1"
        out = str(tmp_path / f"{mode}.npz")
        p = subprocess.run([sys.executable, str(script), root, out, str(D), str(n_states), str(n_utts)], env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "ok" in p.stdout, (p.stdout + p.stderr)[-3000:]
        got[mode] = dict(np.load(out))
    for k in ("sc", "paths", "bw", "bs", "bp"):
        np.testing.assert_array_equal(got["default"][k], got["generic"][k], err_msg=k)
    # (16 states in the 13 frames decode walks: the exit state is out of reach — every score is -inf, as in the reference)
    assert np.isfinite(got["default"]["sc"]).any() == (n_states < D)


def test_reference_error_behaviour(feature_set):
    from sapr_amd.custom_hmm import HMM
    _, flat = feature_set
    with pytest.raises(AssertionError):
        HMM(0, 13)
    with pytest.raises(AssertionError):
        HMM(8, 12, flat)
    h = HMM(8, 13, flat)
    with pytest.raises(ValueError):
        h.decode(flat[0].T)          # (T,13) input: the reference's broadcast error (custom_hmm.py:157)
    with pytest.raises(IndexError):
        h.decode(flat[0][:, :9])     # shorter than D frames (custom_hmm.py:500)


def test_g0_known_answers_from_reference_logs(golden):
    """gamma/xi values printed in the reference's pytest_results (flat start, T = 47)."""
    from sapr_amd.custom_hmm import HMM
    aii, T, S = float(golden["g0_aii"]), int(golden["g0_T"]), 10
    h = HMM(8, 13)
    A = np.zeros((S, S))
    A[0, 1] = 1
    for i in range(1, 9):
        A[i, i], A[i, i + 1] = aii, 1 - aii
    A[-1, -1] = 1
    h.A = A
    h.B = {"mean": np.zeros((S, 13)), "covariance": np.tile(np.eye(13), (S, 1, 1))}
    E = np.full((T, S), -np.inf)
    E[:, 1:-1] = -40.0
    al, sc = h.forward(E)
    be = h.backward(E, sc)
    ga = h.compute_gamma(al, be)
    xi = h.compute_xi(al, be, E)
    np.testing.assert_allclose(ga[2, :3], golden["g0_gamma2"], atol=5e-9)
    np.testing.assert_allclose(ga[23, 1:-1], golden["g0_gamma23"], atol=5e-9)
    np.testing.assert_allclose(xi[1, 1, 1], golden["g0_xi_1_1_1"], rtol=1e-10)


def test_pickle_roundtrip_and_custom_decoder(tmp_path, golden, feature_set):
    import pickle
    from sapr_amd.decoder import Decoder
    by_word, flat = feature_set
    d = tmp_path / "trained_models" / "custom"
    d.mkdir(parents=True)
    for w in VOCAB:
        with open(d / f"{w}_custom_2.pkl", "wb") as f:
            pickle.dump(_hmm(golden, f"g5_model_{w}", flat), f)
    fs = tmp_path / "feature_set"
    fs.mkdir()
    k = 0
    for w in VOCAB:
        for x in by_word[w]:
            np.save(fs / f"s{k:03d}_{w}.npy", x)
            k += 1
    dec = Decoder(models_dir=str(tmp_path / "trained_models"), implementation="custom", n_iter=2)
    assert sorted(dec.vocab) == sorted(VOCAB)
    res = dec.decode_vocabulary(str(fs), verbose=False)
    order = {w: i for i, w in enumerate(dec.vocab)}
    sc = golden["g5_scores"]
    for w in VOCAB:
        assert len(res[w]) == 6
        for r in res[w]:
            assert set(r) == {"sample_index", "true_word", "predicted_word", "log_likelihood", "correct",
                              "state_sequence"}
            assert r["true_word"] == w and r["correct"] == (r["predicted_word"] == w)
            assert len(r["state_sequence"]) == 13
    # the winner of every utterance = first strict maximum over the models in LOAD order
    flat_idx = {id(x): i for i, x in enumerate(flat)}
    with pytest.raises(ValueError):
        dec.decode_word_samples("nosuchword", str(fs))
    with pytest.raises(ValueError):
        Decoder(models_dir=str(tmp_path / "trained_models"), implementation="custom", n_iter=99)
