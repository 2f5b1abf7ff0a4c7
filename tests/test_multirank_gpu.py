"""N>1 training on the real kernels: two processes share the box's GPU, each runs the E-step on its
utterance shard, statistics are summed with torch.distributed (gloo here: RCCL refuses two ranks on
one device; the collective call sites are backend-agnostic), and every rank must end with the
parameters a single process computes from the whole list."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import io, os, sys, contextlib
sys.path.insert(0, sys.argv[1])
mode, rank, world, port, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
import numpy as np, torch
if world > 1:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
from sapr_amd import dist as sd
from tests._synth import VOCAB, synth_feature_set
words = VOCAB[:3]
by_word, flat = synth_feature_set(words, 11, D=13, seed=9)
res = {}
if mode == "custom":
    from sapr_amd.custom_hmm import HMM
    lo, hi = sd.shard_range(len(flat), rank, world)
    with contextlib.redirect_stdout(io.StringIO()):
        h = HMM(8, 13, feature_set=flat[lo:hi], model_name="heed")       # flat start: sharded sums
        feats = by_word["heed"]
        lo, hi = sd.shard_range(len(feats), rank, world)
        hist = h.baum_welch(feats[lo:hi], max_iter=3)
    res = dict(hist=np.asarray(hist), A=h.A, mean=h.B["mean"], cov=h.B["covariance"], gmean=h.global_mean)
else:
    from oracle import hmmlearn_oracle as ho
    from sapr_amd.hmmlearn_hmm import GaussianHMM, fit_models
    sp, A, mu, cv = ho.flat_start(flat, 8)
    models, data = [], []
    for w in words:
        m = GaussianHMM(n_components=10, covariance_type="diag", n_iter=4, params="stmc", implementation="log",
                        min_covar=0.01, init_params="")
        m.means_, m.covars_, m.transmat_, m.startprob_ = mu.copy(), cv.copy(), A.copy(), sp.copy()
        models.append(m)
        lo, hi = sd.shard_range(len(by_word[w]), rank, world)
        shard = by_word[w][lo:hi]
        data.append((np.concatenate([f.T for f in shard], axis=0), [f.shape[1] for f in shard]))
    fit_models(models, data)
    for k, m in enumerate(models):
        res[f"hist{k}"] = np.asarray(list(m.monitor_.history))
        res[f"A{k}"], res[f"mu{k}"], res[f"cv{k}"] = m.transmat_, m.means_, m._covars_
np.savez(out, **res)
if world > 1:
    dist.destroy_process_group()
print("ok", rank)
'''


def _run(tmp_path, mode, world, tag):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(31500 + os.getpid() % 1000 + (7 if mode == "custom" else 0))
    outs = [str(tmp_path / f"{tag}_{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode, str(r), str(world), port, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, logs):
        assert p.returncode == 0 and "ok" in o, o[-3000:]
    return [dict(np.load(o)) for o in outs]


@pytest.mark.parametrize("mode", ["hmmlearn", "custom"])
def test_two_rank_training_equals_single_process(tmp_path, mode):
    single = _run(tmp_path, mode, 1, "single")[0]
    ranks = _run(tmp_path, mode, 2, "pair")
    for k in single:
        # both ranks hold the same model after every all-reduce
        np.testing.assert_array_equal(ranks[0][k], ranks[1][k], err_msg=k)
        # and it is the single-process model up to the re-association of the sums over utterances
        np.testing.assert_allclose(ranks[0][k], single[k], rtol=1e-8, atol=1e-9, err_msg=k)


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's N>1 control flow (model broadcast, barrier-bracketed timing, MAX over ranks, rank-0 JSON
    line) with two ranks sharing this box's GPU over gloo; the driver's runs use RCCL, one rank per GPU."""
    import json
    env = dict(os.environ, SAPR_BENCH_BACKEND="gloo")
    port = str(32500 + os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--utts", "4000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["utterances_per_gpu"] == 4000 and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_em_mode_two_ranks_rehearsal():
    """bench.py --mode em (BASELINE configs[3]: E-step shard + ONE all-reduce of the statistics + M-step) with
    two ranks over gloo on this box's GPU: one JSON line, the all-reduce time split out, log-likelihood rising."""
    import json
    env = dict(os.environ, SAPR_BENCH_BACKEND="gloo")
    port = str(33500 + os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--utts", "3000", "--mode", "em"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["loglik_monotone"]
    assert set(d["phase_ms"]) == {"estep_incl_model_upload", "allreduce", "mstep_incl_stats_d2h"}
    assert d["config"]["allreduce_doubles"] == 10 * (2 + 10 + 100 + 10 + 2 * 10 * 13)


def test_bench_stream_mode_strong_scaling_two_ranks_rehearsal():
    """bench.py --mode stream (BASELINE configs[4]: a host-resident corpus cut into chunks, a contiguous run of chunks
    per rank, no data-path collective): two ranks over gloo on this box's GPU take 3 + 2 of 5 chunks; one JSON line,
    `scaling: strong`, PCIe-inclusive value with the kernels-only rate beside it; and the one-rank run of the same
    corpus for comparison of the bookkeeping (same frames per step)."""
    import json
    lines = {}
    for world in (2, 1):
        env = dict(os.environ, SAPR_BENCH_BACKEND="gloo")
        port = str(35500 + os.getpid() % 1000 + world)
        tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--utts", "2000",
                "--total-utts", "10000", "--mode", "stream"]
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", port] if world == 2 else [sys.executable]) + tail
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        js = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1, out.stdout[-2000:]
        lines[world] = json.loads(js[0])
    d = lines[2]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["pcie_inclusive"] is True
    assert d["config"]["chunks"] == 5 and d["config"]["chunks_per_rank"] == [3, 2]
    assert d["config"]["total_utterances"] == 10000 and d["value"] > 0
    assert d["frames_per_s_kernels_only"] >= d["value"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert lines[1]["config"]["chunks_per_rank"] == [5] and lines[1]["config"]["total_utterances"] == 10000
    # same corpus, same frames per step whatever the world size: value x ms_per_step is the corpus size
    for v in lines.values():
        assert v["value"] * v["ms_per_step"] * 1e-3 == pytest.approx(10000 * 101, rel=1e-6)


@pytest.mark.parametrize("mode, extra", [("pipeline", ["--utts", "4000"]), ("em", ["--utts", "3000"]),
                                         ("stream", ["--utts", "2000", "--total-utts", "6000"])])
def test_bench_starts_its_own_ranks_without_a_launcher(mode, extra):
    """The driver's command shape is `python bench.py --gpus N ...` with NO launcher in front: bench.py then starts
    the N ranks itself as a child `python -m torch.distributed.run` (the parent never touches the GPU) and forwards the
    one JSON line and the exit code.  Two ranks over gloo on this box's one GPU, all three modes."""
    import json
    env = dict(os.environ, SAPR_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode,
           *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["scaling"] == ("strong" if mode == "stream" else "weak")


NCCL_WORKER = r'''
import contextlib, io, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from sapr_amd import dist as sd
assert sd.is_distributed() and dist.get_backend() == "nccl"
from tests._synth import VOCAB, synth_feature_set
from oracle import hmmlearn_oracle as ho
from sapr_amd.custom_hmm import HMM
from sapr_amd.hmmlearn_hmm import GaussianHMM, fit_models
words = VOCAB[:2]
by_word, flat = synth_feature_set(words, 9, D=13, seed=4)
# the three collective call sites on device tensors over RCCL: stats all-reduce (fit_models), the numpy
# staging path (flat start) and the custom two-pass update_B sums
t = torch.arange(8, dtype=torch.float64, device="cuda")
assert torch.equal(sd.allreduce_sum_(t.clone()), t)
a = np.arange(5.0)
assert np.array_equal(sd.allreduce_sum_numpy(a), a)
sp, A, mu, cv = ho.flat_start(flat, 8)
models, data = [], []
for w in words:
    m = GaussianHMM(n_components=10, covariance_type="diag", n_iter=3, params="stmc", implementation="log",
                    min_covar=0.01, init_params="")
    m.means_, m.covars_, m.transmat_, m.startprob_ = mu.copy(), cv.copy(), A.copy(), sp.copy()
    models.append(m)
    data.append((np.concatenate([f.T for f in by_word[w]], axis=0), [f.shape[1] for f in by_word[w]]))
fit_models(models, data)
with contextlib.redirect_stdout(io.StringIO()):
    h = HMM(8, 13, feature_set=flat, model_name="heed")
    hist = h.baum_welch(by_word["heed"], max_iter=2)
np.savez(sys.argv[3], hist=np.asarray(hist), gmean=h.global_mean, A=h.A,
         h0=np.asarray(list(models[0].monitor_.history)), mu0=models[0].means_)
dist.destroy_process_group()
print("ok nccl")
'''


def test_rccl_code_path_at_world_size_one(tmp_path):
    """backend "nccl" IS RCCL on ROCm.  A one-rank process group makes every collective call site of the
    product run through RCCL on device tensors (sapr_amd/dist.py's nccl staging branch included); the result
    must equal the non-distributed run.  Multi-rank RCCL needs one GPU per rank and is the driver's to run."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    out = str(tmp_path / "nccl.npz")
    port = str(34500 + os.getpid() % 1000)
    p = subprocess.run([sys.executable, str(script), ROOT, port, out], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok nccl" in p.stdout, (p.stdout + p.stderr)[-3000:]
    got = dict(np.load(out))
    # the same two trainings without a process group, in this process
    import contextlib
    import io
    from oracle import hmmlearn_oracle as ho
    from sapr_amd.custom_hmm import HMM
    from sapr_amd.hmmlearn_hmm import GaussianHMM, fit_models
    from tests._synth import VOCAB, synth_feature_set
    words = VOCAB[:2]
    by_word, flat = synth_feature_set(words, 9, D=13, seed=4)
    sp, A, mu, cv = ho.flat_start(flat, 8)
    models, data = [], []
    for w in words:
        m = GaussianHMM(n_components=10, covariance_type="diag", n_iter=3, params="stmc", implementation="log",
                        min_covar=0.01, init_params="")
        m.means_, m.covars_, m.transmat_, m.startprob_ = mu.copy(), cv.copy(), A.copy(), sp.copy()
        models.append(m)
        data.append((np.concatenate([f.T for f in by_word[w]], axis=0), [f.shape[1] for f in by_word[w]]))
    fit_models(models, data)
    with contextlib.redirect_stdout(io.StringIO()):
        h = HMM(8, 13, feature_set=flat, model_name="heed")
        hist = h.baum_welch(by_word["heed"], max_iter=2)
    np.testing.assert_array_equal(got["hist"], np.asarray(hist))
    np.testing.assert_array_equal(got["gmean"], h.global_mean)
    np.testing.assert_array_equal(got["A"], h.A)
    np.testing.assert_array_equal(got["h0"], np.asarray(list(models[0].monitor_.history)))
    np.testing.assert_array_equal(got["mu0"], models[0].means_)
