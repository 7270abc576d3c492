"""Row-sharded sampling over the GPUs of one node (one process per GPU, torch.distributed; nccl = RCCL).

The sampling path has no data-path collective: every (row, z) chain is independent, so each rank samples a
contiguous block of TEST ROWS (all z-samples of a row stay on one GPU, like the reference's per-row statistics
at regression/card_regression.py:763-825 need) and RNG streams are keyed by the GLOBAL element index
(`elem_offset`), which makes the samples independent of the number of ranks.  RCCL carries only
the evaluation sums (all_reduce of a short vector) or, on request, an all_gather of the y_0 samples.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def row_shard(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced block [lo, hi) of test rows for `rank` (first n_rows % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def tile_rows(t: torch.Tensor, n_z: int) -> torch.Tensor:
    """(rows, d) -> (rows * n_z, d) in the reference's row-major (row, z) order
    (regression/card_regression.py:1027-1035: repeat(n_z,1,1).transpose(0,1).flatten(0,1))."""
    return t.repeat_interleave(n_z, dim=0).contiguous()


def _default_sample_fn(model, x, y0, mu, n_steps, alphas, omabs, seed, elem_offset):
    from .diffusion_utils import p_sample_loop
    return p_sample_loop(model, x, y0, mu, n_steps, alphas, omabs, True, seed=seed, elem_offset=elem_offset)


def _default_sample_rows_fn(model, x_rows, y0_rows, mu_rows, n_z, n_steps, alphas, omabs, seed, elem_offset):
    """Un-tiled rows in, (rows * n_z, y_dim) out: the kernel does the (row, z) expansion (card_run_args.n_z)."""
    from .diffusion_utils import p_sample_loop
    return p_sample_loop(model, x_rows, y0_rows, mu_rows, n_steps, alphas, omabs, True, seed=seed,
                         elem_offset=elem_offset, n_z=n_z)


def sharded_sample_y0(model, x_rows, y_0_hat_rows, y_T_mean_rows, n_z: int, n_steps: int, alphas, omabs, *,
                      seed: int, group=None, gather: bool = True,
                      sample_fn: Optional[Callable] = None, rank: Optional[int] = None, world: Optional[int] = None):
    """Sample y_0 for every (row, z) with rows sharded over the ranks of `group`.

    x_rows (R, x_dim), y_0_hat_rows / y_T_mean_rows (R, y_dim) are the FULL test set on every rank (they are
    small); each rank tiles and samples only its block.  Returns (y0_local (r_local, n_z, y_dim), (lo, hi)) and,
    when gather=True, instead the full (R, n_z, y_dim) tensor assembled on every rank with all_gather.
    `sample_fn` lets tests substitute the sampler (signature of _default_sample_fn); `rank` / `world` override the
    process group's (to sample one shard of a larger job from a single process; implies no collective)."""
    emulated = rank is not None or world is not None
    if emulated:
        if rank is None or world is None:
            raise ValueError("pass rank and world together")
        if gather:
            raise ValueError("an emulated shard (rank/world override) has no group to gather over: use gather=False")
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    R, y_dim = y_0_hat_rows.shape
    lo, hi = row_shard(R, rank, world)
    if hi > lo:
        if sample_fn is None:        # GPU path: rows are expanded to chains inside the kernel
            out = _default_sample_rows_fn(model, x_rows[lo:hi], y_0_hat_rows[lo:hi], y_T_mean_rows[lo:hi], n_z,
                                          n_steps, alphas, omabs, seed, lo * n_z * y_dim)
        else:
            x = tile_rows(x_rows[lo:hi], n_z)
            y0 = tile_rows(y_0_hat_rows[lo:hi], n_z)
            mu = tile_rows(y_T_mean_rows[lo:hi], n_z)
            out = sample_fn(model, x, y0, mu, n_steps, alphas, omabs, seed, lo * n_z * y_dim)
        local = out.reshape(hi - lo, n_z, y_dim)
    else:
        local = y_0_hat_rows.new_zeros((0, n_z, y_dim))
    if not gather:
        return local, (lo, hi)
    if not dist.is_initialized():
        return local
    # all_gather needs equal shapes: pad every shard to the largest block, then trim
    r_max = -(-R // world)
    pad = local.new_zeros((r_max, n_z, y_dim))
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: row_shard(R, r, world)[1] - row_shard(R, r, world)[0]] for r in range(world)], 0)


def sharded_regression_metrics(y0_local: torch.Tensor, y_true_local: torch.Tensor, *, n_bins: int = 10, group=None,
                               sums_fn: Optional[Callable] = None, **kw):
    """RMSE / PICP / NLL / QICE over the whole (sharded) test set: this rank's additive sums from the HIP metrics kernel
    (card_amd.metrics.regression_metric_sums: [n, sum SE, PICP count, sum NLL, QICE bin counts])
    followed by ONE all-reduce of that short vector -- the only collective of the evaluation.
    y0_local: (rows_local, n_z, 1) samples of this rank's rows; y_true_local: (rows_local, 1).  `sums_fn` lets CPU
    tests substitute the kernel.  Returns (metrics dict, all-reduced sums)."""
    from . import metrics
    fn = metrics.regression_metric_sums if sums_fn is None else sums_fn
    sums = all_reduce_sums(fn(y0_local, y_true_local, n_bins=n_bins, **kw), group=group)
    return metrics.finalize(sums, n_bins), sums


def all_reduce_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """Sum `sums` over the ranks of `group`, in place.  Whenever a process group is initialised the collective is ISSUED,
    also with one rank (so the one-rank nccl test and a one-rank `torch.distributed.run bench.py` line do send the
    float64 (1, 16) device vector through RCCL's all-reduce); without a process group it is the identity.  Under the
    gloo backend (CPU rehearsals; `bench.py --backend gloo --share-device`) a device vector is summed through the host."""
    if not dist.is_initialized():
        return sums
    if sums.is_cuda and dist.get_backend(group) == "gloo":
        host = sums.cpu()
        dist.all_reduce(host, group=group)
        sums.copy_(host)
    else:
        dist.all_reduce(sums, group=group)
    return sums
