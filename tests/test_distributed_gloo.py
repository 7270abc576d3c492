"""World-size-2 gloo test (CPU) of the multi-GPU host path: row sharding, RNG element offsets, all_gather of
the samples and all_reduce of the evaluation sums.  The sampler itself is stood in for by the oracle driven
by its restatement of the in-kernel Philox generator (same (seed, element, draw) keying as the HIP kernel),
so "2 ranks == 1 rank" is checked for exactly the quantity that makes sharding rank-count independent."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from card_amd import distributed as cd
from oracle import card_oracle as orc
from card_amd import metrics
from tests.helpers import cpu_metric_sums, load_golden

N_Z, ROWS = 8, 7


def _oracle_sampler(weights):
    def fn(model, x, y0, mu, n_steps, alphas, omabs, seed, elem_offset):
        noise = orc.philox_noise(seed, n_steps, x.shape[0], y0.shape[1], elem_offset=elem_offset)
        return orc.p_sample_loop(weights, x, y0, mu, n_steps, alphas, omabs, noise, only_last_sample=True)
    return fn


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load_golden("reg_boston_T50")
    x, y0 = g["x"][:ROWS], g["y0hat"][:ROWS]
    full = cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=77,
                                sample_fn=_oracle_sampler(g["weights"]))
    local, (lo, hi) = cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=77,
                                           gather=False, sample_fn=_oracle_sampler(g["weights"]))
    y_true = y0 + 0.3
    fin, sums = cd.sharded_regression_metrics(local, y_true[lo:hi], sums_fn=cpu_metric_sums)
    q.put((rank, full.numpy(), sums.numpy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_shard_is_a_partition():
    for n in (0, 1, 7, 8, 51, 515345):
        for world in (1, 2, 3, 8):
            blocks = [cd.row_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_tile_rows_matches_reference_layout():
    t = torch.arange(6.0).reshape(3, 2)
    ref = t.repeat(4, 1, 1).transpose(0, 1).flatten(0, 1)       # regression/card_regression.py:1027-1035
    assert torch.equal(cd.tile_rows(t, 4), ref)


def test_two_ranks_equal_one_rank():
    g = load_golden("reg_boston_T50")
    x, y0 = g["x"][:ROWS], g["y0hat"][:ROWS]
    single = cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=77,
                                  sample_fn=_oracle_sampler(g["weights"]))
    assert single.shape == (ROWS, N_Z, 1)
    single_sums = cpu_metric_sums(single, y0 + 0.3)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][3] == (0, 4) and res[1][3] == (4, 7)
    for rank, full, sums, _ in res:
        assert np.array_equal(full, single.numpy())           # samples do not depend on the rank count
        np.testing.assert_allclose(sums, single_sums.numpy(), rtol=1e-12)
    m = metrics.finalize(torch.from_numpy(res[0][2]))
    assert res[0][2][0, 0] == ROWS and 0 <= float(m["picp"][0]) <= 1 and 0 <= float(m["qice"][0]) <= 1
    one = metrics.finalize(single_sums)
    for k in ("rmse", "picp", "nll", "qice"):
        assert abs(float(m[k][0]) - float(one[k][0])) < 1e-9, k


def test_emulated_shards_concatenate_to_the_whole():
    """rank / world overrides: one process samples shard r of w without a process group; the blocks tile the whole."""
    g = load_golden("reg_boston_T50")
    x, y0 = g["x"][:ROWS], g["y0hat"][:ROWS]
    fn = _oracle_sampler(g["weights"])
    whole = cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=5, sample_fn=fn)
    parts = [cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=5, sample_fn=fn,
                                  gather=False, rank=r, world=3) for r in range(3)]
    assert [p[1] for p in parts] == [(0, 3), (3, 5), (5, 7)]
    assert torch.equal(torch.cat([p[0] for p in parts], 0), whole)
    with pytest.raises(ValueError):
        cd.sharded_sample_y0(None, x, y0, y0, N_Z, g["T"], g["alphas"], g["omabs"], seed=5, sample_fn=fn, rank=0, world=3)
