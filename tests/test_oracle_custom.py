"""The custom-HMM oracle (numpy restatement) against golden vectors produced by the
imported reference (tests/golden/make_golden.py) and the reference's own logged
known answers (pytest_results/*.txt).  CPU only."""
import numpy as np
import pytest

from oracle import custom_hmm_oracle as co
from tests._synth import VOCAB, synth_feature_set


def _model(g, prefix):
    return g[f"{prefix}_A"], g[f"{prefix}_mean"], g[f"{prefix}_cov"]


def test_g1_flat_start(golden, feature_set):
    _, flat = feature_set
    fs = co.flat_start(flat, 8)
    np.testing.assert_array_equal(fs["global_mean"], golden["g1_gmean"])
    np.testing.assert_allclose(fs["global_covariance"], golden["g1_gcov"], rtol=1e-13, atol=0)
    np.testing.assert_array_equal(fs["A"], golden["g1_A"])
    np.testing.assert_array_equal(fs["mean"], golden["g1_mean"])
    np.testing.assert_allclose(fs["covariance"], golden["g1_cov"], rtol=1e-13)


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_g2_g3_estep_pieces(golden, feature_set, stage):
    by_word, _ = feature_set
    probe = [by_word["heed"][0], by_word["heed"][2], by_word["hood"][1]]
    A, mean, cov = _model(golden, f"g2_s{stage}")
    for u, f in enumerate(probe):
        with np.errstate(all="ignore"):
            E = co.emission(f, mean, cov)
        Eg = golden[f"g2_s{stage}_u{u}_E"]
        np.testing.assert_array_equal(E, Eg)  # same numpy/BLAS calls → same bits
        with np.errstate(all="ignore"):
            E2 = co.emission_rowsum_form(f, mean, cov)
        fin = np.isfinite(Eg)
        np.testing.assert_allclose(E2[fin], Eg[fin], rtol=1e-9)
        with np.errstate(all="ignore"):
            al, sc = co.forward(Eg, A)
            be = co.backward(Eg, A, sc)
            ga = co.gamma(al, be)
            xi = co.xi(al, be, Eg, A)
        np.testing.assert_array_equal(al, golden[f"g3_s{stage}_u{u}_alpha"])
        assert sc == golden[f"g3_s{stage}_u{u}_scale"]
        np.testing.assert_array_equal(be, golden[f"g3_s{stage}_u{u}_beta"])
        np.testing.assert_array_equal(ga, golden[f"g3_s{stage}_u{u}_gamma"])
        np.testing.assert_allclose(xi, golden[f"g3_s{stage}_u{u}_xi"], rtol=1e-15, atol=0, equal_nan=True)


@pytest.mark.parametrize("n_it", [1, 2, 3, 4])
def test_g4_baum_welch(golden, feature_set, n_it):
    by_word, flat = feature_set
    fs = co.flat_start(flat, 8)
    with np.errstate(all="ignore"):
        hist, A, mean, cov = co.baum_welch(by_word["heed"], fs["A"], fs["mean"], fs["covariance"],
                                           fs["global_covariance"], 0.001, max_iter=n_it)
    np.testing.assert_allclose(hist, golden[f"g4_it{n_it}_hist"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(A, golden[f"g4_it{n_it}_A"], rtol=1e-10, atol=1e-300, equal_nan=True)
    np.testing.assert_allclose(mean, golden[f"g4_it{n_it}_mean"], rtol=1e-10, equal_nan=True)
    np.testing.assert_allclose(cov, golden[f"g4_it{n_it}_cov"], rtol=1e-9, atol=1e-9, equal_nan=True)


def test_g5_decode_all_models(golden, feature_set):
    _, flat = feature_set
    for w, word in enumerate(VOCAB):
        A, mean, cov = _model(golden, f"g5_model_{word}")
        for u, f in enumerate(flat):
            with np.errstate(all="ignore"):
                lp, p = co.decode(f, A, mean, cov, 8)
            assert p == list(golden["g5_paths"][u, w]), (word, u)
            np.testing.assert_equal(lp, golden["g5_scores"][u, w])


def test_g5_decode_flat_start_ties(golden, feature_set):
    _, flat = feature_set
    A, mean, cov = golden["g1_A"], golden["g1_mean"], golden["g1_cov"]
    for u, f in enumerate(flat):
        lp, p = co.decode(f, A, mean, cov, 8)
        assert p == list(golden["g5_flat_paths"][u])
        assert lp == golden["g5_flat_scores"][u]


def test_g5_sixteen_states(golden, feature_set):
    _, flat = feature_set
    fs = co.flat_start(flat, 16)
    lp, p = co.decode(flat[0], fs["A"], fs["mean"], fs["covariance"], 16)
    assert lp == golden["g5_s16_d13_score"] == -np.inf
    assert p == list(golden["g5_s16_d13_path"])
    by39, flat39 = synth_feature_set(VOCAB[:3], 4, D=39, seed=5)
    A, mean, cov = _model(golden, "g5_s16_d39")
    for u, f in enumerate(flat39):
        with np.errstate(all="ignore"):
            lp, p = co.decode(f, A, mean, cov, 16)
        assert p == list(golden["g5_s16_d39_paths"][u])
        np.testing.assert_equal(lp, golden["g5_s16_d39_scores"][u])
    with np.errstate(all="ignore"):
        E = co.emission(flat39[1], mean, cov)
        al, sc = co.forward(E, A)
    np.testing.assert_array_equal(E, golden["g5_s16_d39_E"])
    np.testing.assert_array_equal(al, golden["g5_s16_d39_alpha"])


def test_g6_decoder_argmax(golden):
    sc = golden["g5_scores"]
    best = []
    for u in range(sc.shape[0]):
        b, bw = float("-inf"), -1
        for w in range(sc.shape[1]):
            if sc[u, w] > b:
                b, bw = sc[u, w], w
        best.append(bw)
    np.testing.assert_array_equal(best, golden["g6_best_word"])


def test_g0_known_answers_from_reference_logs(golden):
    """pytest_results/training_results.txt:12,41,46 and forward_backward_results.txt:229-232:
    flat start (state-independent emissions) → gamma/xi depend only on A and T."""
    aii, T = float(golden["g0_aii"]), int(golden["g0_T"])
    S = 10
    A = np.zeros((S, S))
    A[0, 1] = 1
    for i in range(1, 9):
        A[i, i], A[i, i + 1] = aii, 1 - aii
    A[-1, -1] = 1
    E = np.full((T, S), -np.inf)
    E[:, 1:-1] = -40.0  # any state-independent value
    al, sc = co.forward(E, A)
    be = co.backward(E, A, sc)
    ga = co.gamma(al, be)
    xi = co.xi(al, be, E, A)
    np.testing.assert_allclose(ga[2, :3], golden["g0_gamma2"], atol=5e-9)
    np.testing.assert_allclose(ga[23, 1:-1], golden["g0_gamma23"], atol=5e-9)
    np.testing.assert_allclose(xi[1, 1, 1], golden["g0_xi_1_1_1"], rtol=1e-12)
    e = np.eye(S)
    np.testing.assert_allclose(ga[0], e[0], atol=1e-12)
    np.testing.assert_allclose(ga[1], e[1], atol=1e-12)
    np.testing.assert_allclose(ga[45], e[8], atol=1e-12)
    np.testing.assert_allclose(ga[46], e[9], atol=1e-12)


# ---------------------------------------------------------------------------------------------------
# The evaluation ORDER of the emission term (custom_hmm.py:168-172).  gram="chain" states it explicitly
# (one k-ascending fused-multiply-add chain per element of both products, numpy's pair-wise row sum); the
# tests below pin that statement to the reference's own outputs, bit for bit, without calling a BLAS.
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_chain_order_reproduces_reference_emission_bits(golden, feature_set, stage):
    by_word, _ = feature_set
    probe = [by_word["heed"][0], by_word["heed"][2], by_word["hood"][1]]
    _, mean, cov = _model(golden, f"g2_s{stage}")
    for u, f in enumerate(probe):
        with np.errstate(all="ignore"):
            E = co.emission(f, mean, cov, gram="chain")
        np.testing.assert_array_equal(E, golden[f"g2_s{stage}_u{u}_E"])


def test_chain_order_reproduces_reference_emission_bits_39_dim(golden):
    _, flat39 = synth_feature_set(VOCAB[:3], 4, D=39, seed=5)
    E = co.emission(flat39[1], golden["g5_s16_d39_mean"], golden["g5_s16_d39_cov"], gram="chain")
    np.testing.assert_array_equal(E, golden["g5_s16_d39_E"])


def test_chain_order_reproduces_flat_start_and_trained_paths_and_scores(golden, feature_set):
    """Flat start: every state has the same Gaussian, all left-to-right paths into a cell tie
    mathematically and the reference's path is decided by the last bit of its emission values — the
    chain order reproduces all 66 paths AND scores exactly (tests/test_decode.py:32-38 pins the length)."""
    _, flat = feature_set
    A, mean, cov = golden["g1_A"], golden["g1_mean"], golden["g1_cov"]
    for u, f in enumerate(flat):
        lp, p = co.decode(f, A, mean, cov, 8, gram="chain")
        assert p == list(golden["g5_flat_paths"][u]), u
        assert lp == golden["g5_flat_scores"][u]
    for w, word in enumerate(VOCAB[:3]):
        A, mean, cov = _model(golden, f"g5_model_{word}")
        for u, f in enumerate(flat):
            with np.errstate(all="ignore"):
                lp, p = co.decode(f, A, mean, cov, 8, gram="chain")
            assert p == list(golden["g5_paths"][u, w])
            np.testing.assert_equal(lp, golden["g5_scores"][u, w])


def test_flat_start_paths_hang_on_the_evaluation_order(golden, feature_set):
    """Why the order matters: evaluating the SAME expression in the algebraically equal row-sum form
    d_t . (C^-1 sum_s d_s) (different rounding, |dE| ~ 1e-12) changes flat-start paths while the scores
    agree to 1e-12 — so a bit-faithful order is the only way to return the reference's paths there."""
    _, flat = feature_set
    A, mean, cov = golden["g1_A"], golden["g1_mean"], golden["g1_cov"]
    changed = 0
    orig = co.emission
    try:
        co.emission = lambda f, m, c, gram="blas": co.emission_rowsum_form(f, m, c)
        for u, f in enumerate(flat):
            lp, p = co.decode(f, A, mean, cov, 8)
            assert abs(lp - golden["g5_flat_scores"][u]) <= 1e-11 * abs(lp)
            changed += p != list(golden["g5_flat_paths"][u])
    finally:
        co.emission = orig
    assert changed > 0
