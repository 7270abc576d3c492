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
                      sample_fn: Optional[Callable] = None):
    """Sample y_0 for every (row, z) with rows sharded over the ranks of `group`.

    x_rows (R, x_dim), y_0_hat_rows / y_T_mean_rows (R, y_dim) are the FULL test set on every rank (they are
    small); each rank tiles and samples only its block.  Returns (y0_local (r_local, n_z, y_dim), (lo, hi)) and,
    when gather=True, instead the full (R, n_z, y_dim) tensor assembled on every rank with all_gather.
    `sample_fn` lets tests substitute the sampler (signature of _default_sample_fn)."""
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
    if not gather or world == 1:
        return (local, (lo, hi)) if not gather else local
    # all_gather needs equal shapes: pad every shard to the largest block, then trim
    r_max = -(-R // world)
    pad = local.new_zeros((r_max, n_z, y_dim))
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: row_shard(R, r, world)[1] - row_shard(R, r, world)[0]] for r in range(world)], 0)


def regression_metric_sums(y0_local: torch.Tensor, y_true_local: torch.Tensor, n_bins: int = 10,
                           picp_range=(2.5, 97.5)) -> torch.Tensor:
    """Per-rank additive evaluation sums for y_dim = 1, the quantities the reference reduces over the test set:
    [n_rows, sum squared error of the predictive mean (card_regression.py:776-783, trimmed range 0-100),
     PICP in-range count (:818-821), QICE bin counts (n_bins + 2 entries, :788-796)].
    y0_local: (rows, n_z, 1) samples, y_true_local: (rows, 1).  Percentiles use numpy's linear interpolation."""
    rows = y0_local.shape[0]
    out = torch.zeros(3 + n_bins + 2, dtype=torch.float64, device=y0_local.device)
    if rows == 0:
        return out
    s = y0_local.reshape(rows, -1).double()
    yt = y_true_local.reshape(rows, 1).double()
    out[0] = rows
    out[1] = ((s.mean(1, keepdim=True) - yt) ** 2).sum()
    q = torch.tensor(picp_range, dtype=torch.float64, device=s.device) / 100
    ci = torch.quantile(s, q, dim=1)                                   # linear interpolation, like np.percentile
    out[2] = ((yt[:, 0] >= ci[0]) & (yt[:, 0] <= ci[1])).sum()
    qs = torch.arange(n_bins + 1, dtype=torch.float64, device=s.device) / n_bins
    quant = torch.quantile(s, qs, dim=1)                               # (n_bins+1, rows)
    member = ((yt.T - quant) > 0).sum(0)                               # 0 .. n_bins+1
    out[3:] = torch.bincount(member, minlength=n_bins + 2).double()
    return out


def finalize_regression_metrics(sums: torch.Tensor, n_bins: int = 10):
    """RMSE, PICP, QICE from (all-reduced) sums; QICE folds the two outer bins like card_regression.py:803-810."""
    n = float(sums[0])
    counts = sums[3:].clone()
    counts[1] += counts[0]
    counts[-2] += counts[-1]
    ratio = counts[1:-1] / n
    return {"rmse": float((sums[1] / n).sqrt()), "picp": float(sums[2] / n),
            "qice": float((1.0 / n_bins - ratio).abs().mean()), "n": int(n)}


def all_reduce_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, group=group)
    return sums
