"""The multi-GPU host path on the hardware a test box has (one MI355X): the default (HIP) sampler behind
`sharded_sample_y0`, shards emulated with the rank / world overrides, and the RCCL branches (all_gather of the samples,
all_reduce of the metric sums) under a one-rank nccl process group.  The world-size-2 logic is covered on CPU with gloo
(tests/test_distributed_gloo.py); a measured multi-GPU run is the driver's (bench.py --gpus N)."""
import os

import pytest
import torch
import torch.distributed as dist

import card_amd
from card_amd import distributed as cd
from card_amd import metrics
from tests.helpers import load_golden
from tests.test_gpu_parity import model_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cuda(t):
    return t.to(DEV)


def _problem(rows=23, n_z=40):
    g = load_golden("reg_boston_T50")
    gen = torch.Generator().manual_seed(3)
    idx = torch.randint(0, g["x"].shape[0], (rows,), generator=gen)
    x = g["x"][idx] + 0.05 * torch.randn(rows, g["x"].shape[1], generator=gen)
    y0 = g["y0hat"][idx] + 0.05 * torch.randn(rows, 1, generator=gen)
    return g, model_of(g), cuda(x), cuda(y0), n_z


def test_emulated_shards_on_the_gpu_equal_one_call():
    """Two (and three) shards sampled one after the other with the default HIP sampler == one call over all rows, bit for
    bit: row blocks + elem_offset make the Philox streams independent of the sharding."""
    g, m, x, y0, n_z = _problem()
    al, om = cuda(g["alphas"]), cuda(g["omabs"])
    whole = cd.sharded_sample_y0(m, x, y0, y0, n_z, g["T"], al, om, seed=77)        # no process group: one shard
    assert whole.shape == (x.shape[0], n_z, 1)
    direct = card_amd.p_sample_loop(m, x, y0, y0, g["T"], al, om, True, seed=77, n_z=n_z)
    assert torch.equal(whole.reshape(-1, 1), direct)
    for world in (2, 3):
        parts = [cd.sharded_sample_y0(m, x, y0, y0, n_z, g["T"], al, om, seed=77, gather=False, rank=r, world=world)
                 for r in range(world)]
        assert [p[1] for p in parts] == [cd.row_shard(x.shape[0], r, world) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts], 0), whole), world


@pytest.fixture()
def one_rank_nccl():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29600 + os.getpid() % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_rccl_branches_under_a_one_rank_group(one_rank_nccl):
    """gather=True goes through dist.all_gather (padded shards, trimmed), the metric sums through dist.all_reduce, both on
    RCCL with device tensors (all_reduce_sums issues the collective whenever a group exists, also with one rank -- counted
    below); with one rank they must return what the local computation gives."""
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append((t.device.type, t.dtype, tuple(t.shape)))
        return real(t, *a, **k)

    dist.all_reduce = counting
    g, m, x, y0, n_z = _problem(rows=11, n_z=64)
    al, om = cuda(g["alphas"]), cuda(g["omabs"])
    local, (lo, hi) = cd.sharded_sample_y0(m, x, y0, y0, n_z, g["T"], al, om, seed=5, gather=False)
    assert (lo, hi) == (0, 11)
    full = cd.sharded_sample_y0(m, x, y0, y0, n_z, g["T"], al, om, seed=5, gather=True)
    assert torch.equal(full, local)
    y_true = y0 + 0.2
    try:
        fin, sums = cd.sharded_regression_metrics(local, y_true, tau=1.0)
    finally:
        dist.all_reduce = real
    assert calls == [("cuda", torch.float64, (1, 16))], calls          # the 16-number vector did go through ncclAllReduce
    want = metrics.regression_metric_sums(local, y_true, tau=1.0)
    assert torch.equal(sums, want) and sums.shape == (1, 16) and float(sums[0, 0]) == 11
    one = metrics.finalize(want)
    for k in ("rmse", "picp", "nll", "qice"):
        assert float(fin[k][0]) == float(one[k][0])


def _bench_child(extra, nproc=None, timeout=420):
    """bench.py as a fresh child process (optionally under torch.distributed.run); returns its one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(29700 + os.getpid() % 200)]
    cmd += [os.path.join(root, "bench.py")] + extra
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])          # rank 0 only, exactly one JSON line
    return json.loads(lines[0])


def test_two_rank_bench_line_on_one_gpu_through_gloo():
    """The N > 1 path of bench.py executed for real on the one GPU a test box has: two ranks under torch.distributed.run,
    both on cuda:0 (--share-device), process group on gloo (--backend gloo: RCCL needs one GPU per rank).  Exercises what
    a driver-launched N = 2 run does outside the kernels -- RANK / WORLD_SIZE from the environment, per-rank seeds and
    `elem_offset = rank * N * y_dim`, dist.barrier on both sides of the timed region, the evaluation sums from the HIP
    metrics kernel through all_reduce, the max-over-ranks time, rank-0-only JSON.  The two ranks share the GPU's time, so
    the whole-job value must come out near the one-rank value (not near twice it): 1 049 rows x 1000 z = one full
    launch of the persistent kernel per rank and step."""
    common = ["--workload", "toy", "--rows", "1049", "--nz", "1000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = _bench_child(["--gpus", "1"] + common)
    two = _bench_child(["--gpus", "2", "--backend", "gloo", "--share-device"] + common, nproc=2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak"
    assert two["config"]["backend"] == "gloo" and two["config"]["ranks_share_one_gpu"] is True
    ratio = two["value"] / one["value"]
    print(f"1 rank {one['value']:.3e}, 2 ranks sharing the GPU {two['value']:.3e} sample-steps/s (ratio {ratio:.3f})")
    assert 0.80 <= ratio <= 1.15, ratio


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the one-GPU test box rehearses this path over gloo above)")
def test_two_rank_bench_line_over_rccl():
    """On a node with at least two GPUs: the driver's own N = 2 launch (torch.distributed.run, one rank per GPU, backend nccl =
    RCCL over xGMI).  Weak scaling with no data-path collective: the whole-job value must come out near twice the one-rank
    value."""
    common = ["--workload", "toy", "--rows", "1049", "--nz", "1000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = _bench_child(["--gpus", "1"] + common)
    two = _bench_child(["--gpus", "2"] + common, nproc=2)
    assert two["n_gpus"] == 2 and two["config"]["backend"] == "nccl" and two["config"]["ranks_share_one_gpu"] is False
    ratio = two["value"] / one["value"]
    print(f"1 rank {one['value']:.3e}, 2 ranks on 2 GPUs {two['value']:.3e} sample-steps/s (ratio {ratio:.3f})")
    assert 1.7 <= ratio <= 2.2, ratio
