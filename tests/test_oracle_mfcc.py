"""Independent cross-checks of the MFCC oracle (oracle/mfcc_oracle.py).  CPU only.

librosa is absent from the image and the reference's tests pin only the output shape
(``tests/test_mfcc_extract.py:31-45``), so the chain stays "parity unpinned" against the reference
itself.  What CAN be pinned is pinned here, piece by piece, against code that shares nothing with the
oracle:

* the numbers librosa publishes in its own documentation (docstring examples of ``hz_to_mel``,
  ``mel_to_hz``, ``mel_frequencies``, ``filters.mel``);
* ``scipy.signal.stft`` and ``torch.stft`` (CPU) for the framing / centring / window / rFFT;
* ``scipy.signal.get_window``, ``scipy.fft.dct``, ``scipy.signal.savgol_filter`` for window, DCT, deltas;
* an end-to-end recomputation through ``torch.stft`` in float64.
"""
import numpy as np
import pytest
import scipy.fft
import scipy.signal

from oracle import mfcc_oracle as mo


def _signal(n=16000, sr=16000, seed=3):
    return mo.synth_utterances(1, n, sr, seed=seed)[0]


# ------------------------------------------------------------------ librosa's published examples
def test_mel_scale_matches_librosa_documentation_examples():
    # librosa.hz_to_mel / mel_to_hz docstrings
    assert mo.hz_to_mel(60) == pytest.approx(0.9, abs=1e-12)
    np.testing.assert_allclose(mo.hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], rtol=1e-12)
    assert mo.mel_to_hz(3) == pytest.approx(200.0, abs=1e-12)
    np.testing.assert_allclose(mo.mel_to_hz([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], atol=5e-4)
    # librosa.mel_frequencies(n_mels=40) docstring (fmin 0, fmax 11025)
    doc = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49,
           1024.856, 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393,
           2470.47, 2697.686, 2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621,
           5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028, 10096.408, 11025.]
    got = mo.mel_to_hz(np.linspace(mo.hz_to_mel(0.0), mo.hz_to_mel(11025.0), 40))
    np.testing.assert_allclose(got, doc, atol=5e-4)
    # Slaney break point: 1 kHz = 15 mel, and 6.4 kHz is 27 log-steps above it
    assert mo.hz_to_mel(1000.0) == pytest.approx(15.0) and mo.hz_to_mel(6400.0) == pytest.approx(42.0)


def test_mel_filterbank_matches_librosa_documentation_and_definition():
    M = mo.mel_filterbank(22050, 2048, 128)
    assert M.shape == (128, 1025) and M.dtype == np.float32
    # librosa.filters.mel(sr=22050, n_fft=2048) docstring: first row starts 0., 0.016, ...; last row ends 0., 0.
    np.testing.assert_allclose(M[0, :2], [0.0, 0.016], atol=5e-4)
    assert M[-1, -1] == 0 and M[1, 0] == 0
    for sr, n_fft, n_mels in [(22050, 2048, 128), (16000, 512, 40)]:
        M = mo.mel_filterbank(sr, n_fft, n_mels).astype(np.float64)
        freqs = np.arange(1 + n_fft // 2) * sr / n_fft
        edges = mo.mel_to_hz(np.linspace(mo.hz_to_mel(0.0), mo.hz_to_mel(sr / 2), n_mels + 2))
        assert np.all(M >= 0)
        for i in range(n_mels):
            nz = np.nonzero(M[i])[0]
            if nz.size == 0:  # librosa warns about empty filters; none at these settings
                pytest.fail(f"empty mel filter {i}")
            # support strictly inside (edge_i, edge_{i+2}), one peak next to the centre edge_{i+1}
            assert freqs[nz[0]] > edges[i] - 1e-9 and freqs[nz[-1]] < edges[i + 2] + 1e-9
            pk = freqs[np.argmax(M[i])]
            assert abs(pk - edges[i + 1]) <= sr / n_fft
            # rises then falls (triangle)
            k = np.argmax(M[i])
            assert np.all(np.diff(M[i, nz[0]:k + 1]) >= -1e-12) and np.all(np.diff(M[i, k:nz[-1] + 1]) <= 1e-12)
            # independent evaluation of the triangle, Slaney area normalisation 2 / (f[i+2] - f[i])
            tri = np.maximum(0, np.minimum((freqs - edges[i]) / (edges[i + 1] - edges[i]),
                                           (edges[i + 2] - freqs) / (edges[i + 2] - edges[i + 1])))
            np.testing.assert_allclose(M[i], tri * 2.0 / (edges[i + 2] - edges[i]), rtol=2e-6, atol=1e-9)
        # area normalisation: every filter wide enough to be sampled integrates to ~1 over frequency
        area = M.sum(axis=1) * sr / n_fft
        wide = (edges[2:] - edges[:-2]) > 8 * sr / n_fft
        np.testing.assert_allclose(area[wide], 1.0, rtol=0.02)


# ------------------------------------------------------------------------------- window, DCT, deltas
def test_window_is_periodic_hamming_centred_in_the_fft_frame():
    for win, n_fft in [(661, 2048), (400, 512)]:
        w = mo.padded_window(win, n_fft)
        n = np.arange(win)
        ham = 0.54 - 0.46 * np.cos(2 * np.pi * n / win)  # periodic (fftbins=True): denominator N, not N-1
        lpad = (n_fft - win) // 2
        np.testing.assert_allclose(w[lpad:lpad + win], ham, atol=1e-15)
        assert np.all(w[:lpad] == 0) and np.all(w[lpad + win:] == 0)
        np.testing.assert_allclose(w[lpad:lpad + win], scipy.signal.get_window("hamming", win, fftbins=True), atol=0)
        assert not np.allclose(w[lpad:lpad + win], scipy.signal.get_window("hamming", win, fftbins=False))


def test_dct_matrix_is_scipy_ortho_dct2():
    for n_mfcc, n_mels in [(13, 128), (13, 40)]:
        m = mo.dct_matrix(n_mfcc, n_mels)
        ref = scipy.fft.dct(np.eye(n_mels), type=2, norm="ortho", axis=0)[:n_mfcc]
        np.testing.assert_allclose(m, ref, atol=1e-14)
        np.testing.assert_allclose(m @ m.T, np.eye(n_mfcc), atol=1e-13)  # orthonormal rows


def test_delta_is_savgol_interp_and_edge_matrices_describe_it():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 37))
    for order in (1, 2):
        want = scipy.signal.savgol_filter(x, 9, deriv=order, polyorder=order, axis=-1, mode="interp")
        np.testing.assert_allclose(mo.delta(x, order), want, atol=0)
        taps, head, tail = mo.delta_edge_matrices(order)
        got = np.empty_like(x)
        for t in range(4, x.shape[1] - 4):
            got[:, t] = x[:, t - 4:t + 5] @ taps
        got[:, :4] = x[:, :9] @ head.T
        got[:, -4:] = x[:, -9:] @ tail.T
        np.testing.assert_allclose(got, want, atol=1e-12)
    # first difference of a straight line is its slope everywhere, edges included (polynomial fit)
    line = 3.0 * np.arange(20.0)[None]
    np.testing.assert_allclose(mo.delta(line, 1), 3.0, atol=1e-12)
    np.testing.assert_allclose(mo.delta(line, 2), 0.0, atol=1e-10)


# ------------------------------------------------------------------------------------------ STFT
@pytest.mark.parametrize("preset", ["REFERENCE", "BENCH"])
def test_power_spectrogram_matches_scipy_and_torch_stft(preset):
    import torch
    p = getattr(mo, preset)
    y = _signal(p["sr"], p["sr"])
    P = mo.power_spectrogram(y, p["n_fft"], p["win_length"], p["hop_length"]).astype(np.float64)
    T = mo.num_frames(len(y), p["hop_length"])
    assert P.shape == (1 + p["n_fft"] // 2, T) and T == 1 + len(y) // p["hop_length"]
    win = mo.padded_window(p["win_length"], p["n_fft"])
    # scipy: explicit zero padding of n_fft/2 (center=True, pad_mode="constant"), no extra padding
    yp = np.pad(y.astype(np.float64), p["n_fft"] // 2)
    _, _, Z = scipy.signal.stft(yp, window=win, nperseg=p["n_fft"], noverlap=p["n_fft"] - p["hop_length"],
                                boundary=None, padded=False, return_onesided=True, scaling="spectrum")
    Ps = np.abs(Z * win.sum()) ** 2
    assert Ps.shape[1] >= T
    np.testing.assert_allclose(P, Ps[:, :T], rtol=2e-5, atol=1e-9 * Ps.max())
    # torch.stft pads the short window to n_fft centred, like librosa's util.pad_center
    w_t = torch.from_numpy(scipy.signal.get_window("hamming", p["win_length"], fftbins=True))
    Zt = torch.stft(torch.from_numpy(y.astype(np.float64)), n_fft=p["n_fft"], hop_length=p["hop_length"],
                    win_length=p["win_length"], window=w_t, center=True, pad_mode="constant",
                    return_complex=True).numpy()
    assert Zt.shape == P.shape
    np.testing.assert_allclose(P, np.abs(Zt) ** 2, rtol=2e-5, atol=1e-9 * Ps.max())


def test_power_to_db_clips_against_the_whole_utterance():
    S = np.array([[1.0, 1e-3], [1e-12, 10.0]], dtype=np.float32)
    db = mo.power_to_db(S, top_db=80.0)
    np.testing.assert_allclose(db, [[0.0, -30.0], [-70.0, 10.0]], atol=1e-5)  # 1e-12 -> amin -> -100 -> clip at 10-80
    assert db.dtype == np.float32
    np.testing.assert_allclose(mo.power_to_db(S, top_db=None)[1, 0], -100.0, atol=1e-5)


# ------------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("preset,deltas,preemph", [("REFERENCE", False, 0.0), ("BENCH", False, 0.0),
                                                   ("BENCH", True, 0.97)])
def test_full_chain_against_an_independent_float64_recomputation(preset, deltas, preemph):
    """torch.stft (float64) -> |.|^2 -> triangle filters evaluated from their definition -> 10 log10 ->
    whole-utterance clip -> scipy DCT -> savgol deltas.  The oracle follows librosa's float32 storage, so
    the two differ by float32 rounding only (|c| up to ~600)."""
    import torch
    p = dict(getattr(mo, preset), deltas=deltas, preemph=preemph)
    y = _signal(p["sr"], p["sr"], seed=11)
    got = mo.mfcc(y, **p)
    yy = y.astype(np.float64)
    if preemph:
        yy = np.r_[yy[0], yy[1:] - np.float64(np.float32(preemph)) * yy[:-1]]
    w_t = torch.from_numpy(scipy.signal.get_window("hamming", p["win_length"], fftbins=True))
    Z = torch.stft(torch.from_numpy(yy), n_fft=p["n_fft"], hop_length=p["hop_length"], win_length=p["win_length"],
                   window=w_t, center=True, pad_mode="constant", return_complex=True).numpy()
    P = np.abs(Z) ** 2
    sr, n_fft, n_mels = p["sr"], p["n_fft"], p["n_mels"]
    freqs = np.arange(1 + n_fft // 2) * sr / n_fft
    edges = mo.mel_to_hz(np.linspace(mo.hz_to_mel(0.0), mo.hz_to_mel(sr / 2), n_mels + 2))
    M = np.stack([np.maximum(0, np.minimum((freqs - edges[i]) / (edges[i + 1] - edges[i]),
                                           (edges[i + 2] - freqs) / (edges[i + 2] - edges[i + 1])))
                  * 2.0 / (edges[i + 2] - edges[i]) for i in range(n_mels)])
    db = 10.0 * np.log10(np.maximum(1e-10, M @ P))
    db = np.maximum(db, db.max() - p["top_db"])
    c = scipy.fft.dct(db, axis=0, type=2, norm="ortho")[:p["n_mfcc"]]
    if deltas:
        c = np.concatenate([c] + [scipy.signal.savgol_filter(c, 9, deriv=o, polyorder=o, axis=-1, mode="interp")
                                  for o in (1, 2)], axis=0)
    assert got.shape == c.shape and got.dtype == np.float32
    assert got.shape[0] == p["n_mfcc"] * (3 if deltas else 1) and got.shape[1] == 1 + len(y) // p["hop_length"]
    np.testing.assert_allclose(got, c, atol=2e-3)


def test_silence_becomes_identical_frames_after_the_clip():
    """The reference's data shows identical leading frames (forward_backward_results.txt:14-19): digital
    silence hits amin, and the whole-utterance top_db clip lifts it to max-80 dB in every mel band."""
    y = _signal(16000, 16000, seed=5).copy()
    y[:3200] = 0
    c = mo.mfcc(y, **mo.BENCH)
    assert np.all(c[:, 0:1] == c[:, :8])          # frames whose windows lie in the zeroed part
    assert not np.array_equal(c[:, 0], c[:, 50])


@pytest.mark.parametrize("preset", ["REFERENCE", "BENCH"])
def test_chain_against_the_librosa_compatible_routines_of_transformers(preset):
    """`transformers.audio_utils` carries its own re-implementation of librosa's Slaney filterbank, framed STFT and
    power_to_db (the Whisper feature extractor is validated against librosa with them).  It shares no code with this
    restatement: filterbank, log-mel spectrogram and the 13 cepstra (DCT by scipy) must agree."""
    au = pytest.importorskip("transformers.audio_utils")
    cfg = getattr(mo, preset)
    sr, n_fft, win, hop, n_mels = cfg["sr"], cfg["n_fft"], cfg["win_length"], cfg["hop_length"], cfg["n_mels"]
    fb = au.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0,
                            max_frequency=sr / 2, sampling_rate=sr, norm="slaney", mel_scale="slaney")
    mine = mo.mel_filterbank(sr, n_fft, n_mels, dtype=np.float64)
    np.testing.assert_allclose(fb.T, mine, rtol=1e-6, atol=1e-9)
    y = _signal(sr, sr, seed=5)
    window = au.window_function(win, "hamming", periodic=True, frame_length=n_fft, center=True)
    np.testing.assert_allclose(window, mo.padded_window(win, n_fft), rtol=1e-12, atol=1e-15)
    logmel = au.spectrogram(y.astype(np.float64), window, frame_length=n_fft, hop_length=hop, fft_length=n_fft, power=2.0,
                            center=True, pad_mode="constant", mel_filters=fb, mel_floor=1e-10, log_mel="dB",
                            reference=1.0, min_value=1e-10, db_range=80.0, dtype=np.float64)
    assert logmel.shape == (n_mels, 1 + len(y) // hop)
    want = scipy.fft.dct(logmel, type=2, norm="ortho", axis=0)[: cfg["n_mfcc"]]
    got = mo.mfcc(y, **cfg)
    assert got.shape == want.shape
    # float32 chain vs float64 chain on coefficients of magnitude up to ~600
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()   # measured 4.1e-5 / 2.8e-5
