"""On-device regression test metrics (RMSE / QICE / PICP / NLL pieces) for generated samples.

Host-side mirror of the metric helpers inside `Diffusion.test` (regression/card_regression.py:763-893): the
per-row statistics come from `card_regression_row_metrics` (one workgroup sorts a row's n_z samples in LDS);
the reductions over rows are tiny torch ops and, across GPUs, one all-reduce (card_amd.distributed)."""
import ctypes as C

import torch

from . import _lib
from ._lib import CardError
from .sampler import _dev_f32, _ptr, _stream, _require_cuda


def regression_row_metrics(samples, y_true, *, n_bins=10, picp_range=(2.5, 97.5), trimmed_mean_range=(0.0, 100.0),
                           y_scale=1.0, y_shift=0.0, tau=None):
    """samples: (rows, n_z) / (rows, n_z, 1) or (n_states, rows, n_z[, 1]) generated y (normalised, chain order
    (row, z)); y_true: (rows,) / (rows, 1) normalised targets.  Returns dict of (n_states, rows) tensors:
    se, nll (float32), qice_bin, picp_in (int32)."""
    dev = samples.device
    _require_cuda(dev)
    s = _dev_f32(samples, dev)
    if s.dim() >= 3 and s.shape[-1] == 1:
        s = s.squeeze(-1)
    if s.dim() == 2:
        s = s.unsqueeze(0)
    if s.dim() != 3:
        raise CardError(f"samples: expected (rows, n_z) or (n_states, rows, n_z), got {tuple(samples.shape)}")
    n_states, rows, n_z = s.shape
    yt = _dev_f32(y_true, dev).reshape(-1)
    if yt.numel() != rows:
        raise CardError(f"y_true has {yt.numel()} entries, samples have {rows} rows")
    out = dict(se=torch.empty(n_states, rows, device=dev), nll=torch.empty(n_states, rows, device=dev),
               qice_bin=torch.empty(n_states, rows, device=dev, dtype=torch.int32),
               picp_in=torch.empty(n_states, rows, device=dev, dtype=torch.int32))
    a = _lib.MetricArgs()
    a.struct_bytes = C.sizeof(_lib.MetricArgs)
    a.n_z, a.rows, a.n_states, a.n_bins = n_z, rows, n_states, int(n_bins)
    a.samples, a.y_true = _ptr(s), _ptr(yt)
    a.y_scale, a.y_shift = float(y_scale), float(y_shift)
    a.trim_lo, a.trim_hi = float(trimmed_mean_range[0]), float(trimmed_mean_range[1])
    a.picp_lo, a.picp_hi = float(picp_range[0]), float(picp_range[1])
    a.tau = float(tau) if tau is not None else -1.0
    a.se, a.nll, a.qice_bin, a.picp_in = (_ptr(out[k]) for k in ("se", "nll", "qice_bin", "picp_in"))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().card_regression_row_metrics(C.byref(a), _stream(dev)), "card_regression_row_metrics")
    return out


def regression_metric_sums(samples, y_true, *, n_bins=10, **kw):
    """Additive sums per state, layout [n_rows, sum SE, PICP count, sum NLL, QICE counts (n_bins + 2)] as float64
    (n_states, 4 + n_bins + 2): all-reduce these over ranks, then `finalize`."""
    m = regression_row_metrics(samples, y_true, n_bins=n_bins, **kw)
    n_states, rows = m["se"].shape
    out = torch.zeros(n_states, 4 + n_bins + 2, dtype=torch.float64, device=m["se"].device)
    out[:, 0] = rows
    out[:, 1] = m["se"].double().sum(1)
    out[:, 2] = m["picp_in"].double().sum(1)
    out[:, 3] = m["nll"].double().sum(1)
    onehot = torch.nn.functional.one_hot(m["qice_bin"].long(), n_bins + 2)
    out[:, 4:] = onehot.sum(1).double()
    return out


def finalize(sums, n_bins=10):
    """RMSE, PICP, NLL, QICE per state from (all-reduced) sums (regression/card_regression.py:803-810, :1109-1124)."""
    n = sums[:, 0]
    counts = sums[:, 4:].clone()
    counts[:, 1] += counts[:, 0]
    counts[:, -2] += counts[:, -1]
    ratio = counts[:, 1:-1] / n[:, None]
    return {"rmse": (sums[:, 1] / n).sqrt(), "picp": sums[:, 2] / n, "nll": sums[:, 3] / n,
            "qice": (1.0 / n_bins - ratio).abs().mean(1)}
