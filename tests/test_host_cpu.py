"""CPU-only checks of the host logic: imports without a GPU, loud failure without a device, WAV
ingest, M-step formulas, tile layout, multi-process statistics reduction over gloo."""
import os
import struct
import subprocess
import sys
import wave

import numpy as np
import pytest

from oracle import hmmlearn_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_modules_import_without_gpu_and_fail_loudly():
    import torch
    from sapr_amd import custom_hmm, decoder, hmmlearn_hmm, mfcc_extract  # noqa: F401
    from sapr_amd._lib import SaprHipError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = custom_hmm.HMM(8, 13)           # no feature_set → no kernel call
    assert h.total_states == 10 and h.pi[0] == 1.0
    with pytest.raises(SaprHipError):   # no CPU fallback anywhere
        custom_hmm.HMM(8, 13, [np.zeros((13, 20), np.float32)])
    m = hmmlearn_hmm.GaussianHMM(n_components=10, n_iter=2, init_params="")
    m.means_, m.covars_ = np.zeros((10, 13)), np.ones((10, 13))
    m.transmat_, m.startprob_ = np.eye(10), np.r_[1.0, np.zeros(9)]
    with pytest.raises(SaprHipError):
        m.decode(np.zeros((5, 13), np.float32))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sapr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_wav_reader_and_feature_store(tmp_path):
    from sapr_amd import mfcc_extract as me
    sr = 22050
    t = np.linspace(0, 1.0, sr)
    y = np.sin(2 * np.pi * 440 * t)
    p16 = tmp_path / "a_heed.wav"
    with wave.open(str(p16), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(sr)
        st = np.stack([y, y], axis=1)
        w.writeframes((st * 32767).astype("<i2").tobytes())
    x, got_sr = me.read_wav(str(p16))
    assert got_sr == sr and x.dtype == np.float32 and x.shape == (sr,)
    np.testing.assert_allclose(x, y, atol=1e-4)
    pf = tmp_path / "b.wav"   # IEEE float mono
    data = y.astype("<f4").tobytes()
    with open(pf, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " +
                struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32) + b"data" + struct.pack("<I", len(data)) + data)
    x2, sr2 = me.load_audio(str(pf), sr=16000)
    assert sr2 == 16000 and np.allclose(x2, y, atol=1e-6)
    # host half of the GPU resampler (taps + trimming) reproduces scipy.signal.resample_poly
    from scipy.signal import resample_poly, upfirdn
    taps, n_pre = me.resample_design(441, 320)
    full = upfirdn(taps.astype(np.float64), y, 441, 320)
    n_out = -(-len(y) * 441 // 320)
    np.testing.assert_allclose(full[n_pre:n_pre + n_out], resample_poly(y, 441, 320), atol=1e-6)
    with pytest.raises(ValueError):
        me.read_wav(__file__)
    # .npy store helpers (mfcc_extract.py:55-89)
    d = tmp_path / "feature_set"
    d.mkdir()
    a, b = np.random.rand(13, 40).astype(np.float32), np.random.rand(13, 50).astype(np.float32)
    np.save(d / "x1_heed.npy", a)
    np.save(d / "x2_hood.npy", b)
    (d / "junk.txt").write_text("x")
    assert len(me.load_mfccs(str(d))) == 2
    np.testing.assert_array_equal(me.load_mfccs_by_word(str(d), "hood")[0], b)
    assert me.load_mfccs_by_word(str(d), "had") == []
    np.testing.assert_array_equal(me.load_mfcc(str(d / "x1_heed.npy")), a)
    with pytest.raises(Exception):
        me.load_mfcc(str(d / "missing.npy"))


def test_m_step_matches_oracle_formulas():
    from sapr_amd.hmmlearn_hmm import ConvergenceMonitor, m_step
    rng = np.random.default_rng(0)
    S, D = 10, 13
    st = {"start": np.r_[7.0, np.zeros(S - 1)], "trans": rng.uniform(0, 5, (S, S)), "post": rng.uniform(1, 9, S),
          "obs": rng.normal(0, 5, (S, D)), "obs**2": rng.uniform(50, 90, (S, D))}
    A = np.triu(np.tril(np.ones((S, S)), 1)) / 2.0
    sp = np.r_[1.0, np.zeros(S - 1)]
    got = m_step(st, sp, A)
    ref = ho.m_step({**st, "obs2": st["obs**2"]}, sp, A)
    for g, r in zip(got, ref):
        np.testing.assert_allclose(g, r, rtol=1e-15)
    assert np.all(got[1][A == 0] == 0)
    mon = ConvergenceMonitor(1e-2, 15)
    for v in (-10.0, -5.0, -4.995):
        mon.report(v)
    assert mon.converged and list(mon.history) == [-10.0, -5.0, -4.995]


def test_tile_layout():
    import torch
    from sapr_amd.trellis import TileLayout, is_bidiagonal, split_stats, stats_width
    lengths = np.r_[np.full(300, 50), np.full(10, 70), np.full(5, 20)]
    utt_model = np.r_[np.zeros(300), np.ones(10), np.full(5, 2)].astype(int)
    lay = TileLayout.build(lengths, utt_model, 3, torch.device("cpu"))
    assert lay.n_tiles == 4 and list(lay.tile_model.numpy()) == [0, 0, 1, 2]
    assert list(lay.model_tile_off.numpy()) == [0, 2, 3, 4]
    su = lay.slot_utt.numpy()
    assert sorted(su[su >= 0]) == list(range(315)) and (su[:300] < 300).all() and su[512] >= 300
    assert stats_width(10, 13) == 2 + 10 + 100 + 10 + 260
    row = np.arange(stats_width(3, 2), dtype=float)
    s = split_stats(row, 3, 2)
    assert s["nobs"] == 0 and s["logprob"] == 1 and s["trans"].shape == (3, 3) and s["obs**2"][-1, -1] == row[-1]
    assert is_bidiagonal(np.eye(4) * 0.5 + np.eye(4, k=1) * 0.5) and not is_bidiagonal(np.ones((3, 3)))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sapr_amd import dist as sd
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
assert sd.is_distributed() and sd.world() == (rank, world)
lo, hi = sd.shard_range(11)
stats = torch.zeros(3, 5, dtype=torch.float64)
for u in range(lo, hi):            # every rank adds the statistics of ITS utterances
    stats += float(u + 1)
sd.allreduce_sum_(stats)
arr = sd.allreduce_sum_numpy(np.array([hi - lo, 1.0]))
assert torch.all(stats == 66.0), stats      # 1 + 2 + ... + 11
assert arr[0] == 11 and arr[1] == world
dist.destroy_process_group()
print("ok", rank)
'''


def test_suffstat_allreduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 1000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o
    from sapr_amd.dist import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


WORKER_EM = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from oracle import hmmlearn_oracle as ho
from sapr_amd import dist as sd
from sapr_amd.hmmlearn_hmm import m_step
from sapr_amd.trellis import split_stats, stats_width
from tests._synth import VOCAB, synth_feature_set
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
by_word, flat = synth_feature_set(VOCAB[:2], 7, D=13, seed=3)
sp, A, mu, cv = ho.flat_start(flat, 8)
mu, cv = mu.astype(np.float64), cv.astype(np.float64)
S, D = 10, 13
def pack(st, lp):
    return np.concatenate([[st["nobs"], lp], st["start"], st["trans"].ravel(), st["post"], st["obs"].ravel(), st["obs2"].ravel()])
stats = np.zeros((2, stats_width(S, D)))
for w, word in enumerate(VOCAB[:2]):                 # this rank's shard of every word's utterances
    lo, hi = sd.shard_range(len(by_word[word]))
    st, lp = ho.new_stats(S, D), 0.0
    for f in by_word[word][lo:hi]:
        lp += ho.accumulate(st, np.ascontiguousarray(f.T), sp, A, mu, cv)
    stats[w] = pack(st, lp)
t = torch.from_numpy(stats)
sd.allreduce_sum_(t)                                  # the one collective of an EM iteration
out = []
for w, word in enumerate(VOCAB[:2]):
    got = m_step(split_stats(t.numpy()[w], S, D), sp, A, means=mu, covars=cv)
    ref_st, _ = ho.new_stats(S, D), None
    for f in by_word[word]:
        ho.accumulate(ref_st, np.ascontiguousarray(f.T), sp, A, mu, cv)
    ref = ho.m_step(ref_st, sp, A)
    for g, r in zip(got, ref):
        np.testing.assert_allclose(g, r, rtol=1e-10, atol=1e-12)
dist.destroy_process_group()
print("ok", rank)
'''


def test_sharded_em_iteration_two_ranks_gloo(tmp_path):
    """N>1 path on CPU: each rank accumulates the statistics of its shard (the oracle stands in for the
    E-step kernel), one all-reduce, identical M-step everywhere == the single-process M-step."""
    script = tmp_path / "worker_em.py"
    script.write_text(WORKER_EM)
    port = str(30500 + os.getpid() % 1000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_compat_shims_expose_the_reference_module_names(tmp_path):
    """with sapr_amd/compat first on the path, the reference's bare imports
    (`from custom_hmm import HMM`, `from decoder import Decoder`, ...) resolve to the drop-in."""
    code = (
        "import sys; sys.path.insert(0, sys.argv[1])\n"
        "from custom_hmm import HMM\n"
        "from hmmlearn_hmm import HMMLearnModel\n"
        "from mfcc_extract import extract_mfcc, extract_mfccs, load_mfcc, load_mfccs, load_mfccs_by_word\n"
        "from decoder import Decoder\n"
        "import sapr_amd.custom_hmm, sapr_amd.decoder\n"
        "assert HMM is sapr_amd.custom_hmm.HMM and Decoder is sapr_amd.decoder.Decoder\n"
        "h = HMM(8, 13); assert h.total_states == 10\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "sapr_amd", "compat")], cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_feature_store_roundtrip_and_word_selection(tmp_path):
    """store.FeatureStore: packed [frames, D] + offsets reproduce the per-file (D,T) arrays and the
    reference's ``_<word>.npy`` selection rule (mfcc_extract.py:82-89)."""
    from sapr_amd.store import FeatureStore
    from tests._synth import synth_feature_set
    by_word, _ = synth_feature_set(words=["heed", "hid", "had"], n_per_word=3)
    fdir = tmp_path / "feature_set"
    fdir.mkdir()
    for w, lst in by_word.items():
        for k, f in enumerate(lst):
            np.save(fdir / f"sp{k:02d}_{w}.npy", f)
    (fdir / "notes.txt").write_text("ignored")
    st = FeatureStore.pack_directory(str(fdir), str(tmp_path / "store"))
    assert len(st) == 9 and st.frames.shape[1] == 13 and st.offsets[-1] == st.frames.shape[0]
    for i, name in enumerate(st.names):
        assert np.array_equal(st.utterance(i), np.load(fdir / (name + ".npy")))
    idx = st.indices_for_word("hid")
    assert [st.names[i] for i in idx] == ["sp00_hid", "sp01_hid", "sp02_hid"]
    (X, ln), = st.training_data(["hid"])
    assert np.array_equal(X, np.concatenate([f.T for f in by_word["hid"]])) and list(ln) == [f.shape[1] for f in by_word["hid"]]
    # shards tile the store without overlap
    parts = [st.shard(r, 4) for r in range(4)]
    assert sum(len(p) for p in parts) == 9
    assert np.array_equal(np.concatenate([p.frames for p in parts]), st.frames)
    assert [n for p in parts for n in p.names] == st.names
    with pytest.raises(ValueError):
        FeatureStore.write(str(tmp_path / "bad"), [np.zeros((13, 4)), np.zeros((12, 4))], ["a_x", "b_x"])


def test_eval_metrics_follow_sklearn_conventions():
    """sapr_amd.eval (eval.py:16-38): label extraction order, sklearn's confusion-matrix convention (sorted
    distinct labels actually present) and accuracy, checked against scikit-learn itself."""
    from sklearn.metrics import accuracy_score, confusion_matrix as sk_cm
    from sapr_amd import eval as ev
    vocab = ["heed", "hid", "head", "had"]
    rng = np.random.default_rng(0)
    results = {w: [{"true_word": w, "predicted_word": vocab[int(rng.integers(0, 3))], "correct": False}
                   for _ in range(5)] for w in vocab[:3]}                    # "had" never occurs
    for w, lst in results.items():
        for r in lst:
            r["correct"] = r["predicted_word"] == w
    t, p = ev.extract_labels(results)
    assert t == [w for w in vocab[:3] for _ in range(5)] and len(p) == 15
    cm, acc = ev.calculate_metrics(t, p, vocab)
    m = {w: i for i, w in enumerate(vocab)}
    ti, pi = [m[x] for x in t], [m[x] for x in p]
    np.testing.assert_array_equal(cm, sk_cm(ti, pi))
    assert cm.shape == (3, 3) and acc == accuracy_score(ti, pi)
    with pytest.raises(KeyError):
        ev.calculate_metrics(["nope"], ["heed"], vocab)


def test_mp3_folder_is_refused_loudly_not_silently_emptied(tmp_path):
    """The reference's dataset is mp3 (mfcc_extract.py:36); this build has no mp3 decoder, so the folder is
    refused before any work instead of returning "Completed processing 0 files"."""
    from sapr_amd import mfcc_extract as me
    src = tmp_path / "dev_set"
    src.mkdir()
    (src / "sp01_heed.mp3").write_bytes(b"ID3\x03\x00")
    with pytest.raises(me.AudioDecodeUnavailable, match="mp3"):
        me.extract_mfccs(str(src), str(tmp_path / "feature_set"))
    assert not (tmp_path / "feature_set").exists()


def test_m_step_startprob_zero_sum_guard():
    """hmmlearn.utils.normalize divides a zero sum by 1 (zeros stay zeros, no NaN)."""
    from sapr_amd.hmmlearn_hmm import m_step
    S, D = 4, 3
    stats = {"start": np.zeros(S), "trans": np.ones((S, S)), "post": np.ones(S), "obs": np.zeros((S, D)),
             "obs**2": np.ones((S, D)), "nobs": 1}
    sp0 = np.r_[1.0, np.zeros(S - 1)]
    out = m_step(stats, sp0, np.full((S, S), 0.25), startprob_prior=1.0, means=np.zeros((S, D)), covars=np.ones((S, D)))
    assert np.all(out[0] == 0.0) and not np.any(np.isnan(out[0]))


def test_torch_ops_register_without_a_gpu_and_have_no_cpu_implementation():
    import torch
    import sapr_amd.torch_ops  # noqa: F401
    for name in ("pcm16_to_f32", "mfcc_batch", "viterbi_decode_best", "hmm_estep", "custom_estep", "custom_decode"):
        assert hasattr(torch.ops.sapr, name)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.sapr.pcm16_to_f32(torch.zeros(4, dtype=torch.int16))


def test_numpy_float32_axis0_sum_is_a_sequential_chain_on_the_golden_build():
    """The order sapr_colsum_f32 reproduces (HMMLearnModel's flat start): on this container's CPU numpy adds the rows
    of a float32 (N, D) array one after another.  Not every CPU / numpy dispatch does (the GPU box's EPYC does not):
    there the check is skipped, not failed — the kernel's contract is the explicit chain."""
    rng = np.random.default_rng(0)
    X = (rng.normal(0, 20, (50_000, 13)) - np.r_[300, np.zeros(12)]).astype(np.float32)
    if not np.array_equal(np.sum(X, axis=0), np.cumsum(X, axis=0, dtype=np.float32)[-1]):
        pytest.skip("this CPU's numpy reduces float32 axis-0 sums in another order")
    n = X.shape[0]
    m = np.cumsum(X, axis=0, dtype=np.float32)[-1] / np.float32(n)
    np.testing.assert_array_equal(np.mean(X, axis=0), m)
    d = X - m
    np.testing.assert_array_equal(np.var(X, axis=0), np.cumsum(d * d, axis=0, dtype=np.float32)[-1] / np.float32(n))


def test_float64_features_that_do_not_fit_float32_are_refused_not_narrowed():
    """GaussianHMM.decode / score / fit compute on float32 features (the reference's MFCCs are float32,
    mfcc_extract.py:15-24); hmmlearn would use a float64 X at full width, so a lossy narrowing is an error."""
    from sapr_amd.hmmlearn_hmm import _features_f32
    x32 = np.linspace(-300, 50, 26, dtype=np.float32).reshape(2, 13)
    assert _features_f32(x32) is not None and _features_f32(x32).dtype == np.float32
    np.testing.assert_array_equal(_features_f32(x32.astype(np.float64)), x32)     # float32 values in a float64 array
    np.testing.assert_array_equal(_features_f32(np.arange(26).reshape(2, 13)), np.arange(26, dtype=np.float32).reshape(2, 13))
    with pytest.raises(ValueError, match="round-trip through float32"):
        _features_f32(x32.astype(np.float64) + 1e-9)
    assert np.isnan(_features_f32(np.array([[np.nan, 1.0]]))[0, 0])


def test_bench_without_a_launcher_starts_n_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus N` with no RANK in the environment must not die (round 3 did): it starts
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same argv>` as a CHILD before anything touches
    torch or the GPU and exits with the child's code.  The child command is captured here, not run."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("sapr_bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_vectorised_m_step_is_the_per_model_m_step_bit_for_bit():
    """hmmlearn_hmm.m_step_batch (one numpy pass over the whole vocabulary's statistics, what fit_models runs) against W
    calls of m_step on split_stats' dictionaries: padded state counts and feature widths, a state that was never
    visited, a zero-sum startprob."""
    from sapr_amd.hmmlearn_hmm import m_step, m_step_batch
    from sapr_amd.trellis import split_stats, stats_width
    for S, D, Sm, Dm in ((10, 13, 10, 13), (18, 39, 14, 26), (10, 13, 7, 5)):
        W = 5
        rng = np.random.default_rng(S * D)
        rows = np.abs(rng.standard_normal((W, stats_width(S, D)))) * 5 + 0.1
        rows[2, 2 + S + S * S + 3] = 0.0
        sp = np.zeros((W, Sm))
        sp[:, 0] = 1
        sp[4] = 0
        A = np.tile(np.eye(Sm) * 0.8 + np.eye(Sm, k=1) * 0.2, (W, 1, 1))
        A[:, -1, -1] = 1
        mu, cv = rng.standard_normal((W, Sm, Dm)), np.abs(rng.standard_normal((W, Sm, Dm))) + 1
        one = [m_step(split_stats(rows[w], S, D, Sm, Dm), sp[w], A[w], means=mu[w], covars=cv[w]) for w in range(W)]
        many = m_step_batch(rows, S, D, sp, A, mu, cv, S_model=Sm, D_model=Dm)
        for i in range(4):
            np.testing.assert_array_equal(np.stack([o[i] for o in one]), many[i])
        np.testing.assert_array_equal(many[4], rows[:, 1])
