"""The guidance network f_phi on the device (card_fphi_*): y_0_hat = f_phi(x), evaluated inside the sampling call.

Reference: regression/model.py:55-82 DeterministicFeedForwardNeuralNetwork and the OLS guidance of
Diffusion.compute_guiding_prediction (regression/card_regression.py:111-124).  The reference evaluates f_phi once per
test minibatch (:1021) and then tiles x, y_0_hat and y_T_mean n_z times (:1027-1035); with `Sampler.run(..., fphi=g,
n_z=n)` both happen inside card_sampler_run: one kernel over the un-tiled rows, then the sampling kernels expand rows to
chains themselves.
"""
import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import CardError
from .sampler import _content_hash, _dev_f32, _fingerprint, _ptr, _require_cuda, _stream


class GuidanceNet:
    """A packed f_phi on one GPU.  Immutable after construction."""

    def __init__(self, weights, biases, *, negative_slope=0.01, bn_scale=None, bn_shift=None, device):
        device = torch.device(device)
        _require_cuda(device)
        n = len(weights)
        if not 1 <= n <= _lib.FPHI_MAX_LAYERS:
            raise CardError(f"f_phi with {n} Linear layers; the kernel takes 1..{_lib.FPHI_MAX_LAYERS}")
        d = _lib.FphiDesc()
        d.struct_bytes = C.sizeof(_lib.FphiDesc)
        d.n_layers = n
        d.negative_slope = float(negative_slope)
        keep = []

        def dev(t):
            t = _dev_f32(t, device)
            keep.append(t)
            return t.data_ptr()

        d.dims[0] = int(weights[0].shape[1])
        for l, W in enumerate(weights):
            if W.dim() != 2 or int(W.shape[1]) != d.dims[l]:
                raise CardError(f"f_phi layer {l}: weight {tuple(W.shape)} does not take {d.dims[l]} inputs")
            d.dims[l + 1] = int(W.shape[0])
            d.W[l] = dev(W)
            if biases[l] is not None:
                d.b[l] = dev(biases[l])
            if bn_scale is not None and bn_scale[l] is not None:
                d.bn_scale[l], d.bn_shift[l] = dev(bn_scale[l]), dev(bn_shift[l])
        self.device, self.dim_in, self.dim_out = device, int(d.dims[0]), int(d.dims[n])
        h = C.c_void_p()
        L = _lib.lib()
        with torch.cuda.device(device):
            _lib.check(L.card_fphi_create(C.byref(d), _stream(device), C.byref(h)), "card_fphi_create")
        self._h = h
        weakref.finalize(self, L.card_fphi_destroy, h)

    @classmethod
    def from_module(cls, net, device=None):
        """From a DeterministicFeedForwardNeuralNetwork (its `.network` Sequential: Linear, [BatchNorm1d], LeakyReLU,
        Dropout per hidden layer, then Linear -- regression/model.py:70-79) or any Sequential of those layer types."""
        d = describe_guidance(net)
        device = device if device is not None else d["weights"][0].device
        sc = [None if bn is None else _bn_fold(bn)[0] for bn in d["batchnorm"]]
        sh = [None if bn is None else _bn_fold(bn)[1] for bn in d["batchnorm"]]
        return cls(d["weights"], d["biases"], negative_slope=d["negative_slope"], bn_scale=sc, bn_shift=sh, device=device)

    @classmethod
    def from_ols(cls, beta, device=None):
        """OLS guidance (regression/card_regression.py:76-92, :115-117): beta is (1 + x_dim, y_dim), y = cat(1, x).mm(beta)."""
        device = device if device is not None else beta.device
        return cls([beta[1:].t().contiguous()], [beta[0].contiguous()], device=device)

    def forward(self, x, out=None):
        """(rows, dim_out) = f_phi(x): card_fphi_forward (compute_guiding_prediction, regression/card_regression.py:111-124)."""
        x = _dev_f32(x, self.device)
        if x.dim() != 2 or x.shape[1] != self.dim_in:
            raise CardError(f"f_phi: x must be (rows, {self.dim_in}), got {tuple(x.shape)}")
        rows = int(x.shape[0])
        if out is None:
            out = torch.empty(rows, self.dim_out, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().card_fphi_forward(self._h, _ptr(x), rows, _ptr(out), _stream(self.device)),
                       "card_fphi_forward")
        return out

    __call__ = forward


def _bn_fold(m):
    """eval-mode BatchNorm1d as (scale, shift) per feature"""
    s = (m.weight if m.affine else torch.ones_like(m.running_var)) / torch.sqrt(m.running_var + m.eps)
    return s, (m.bias if m.affine else torch.zeros_like(m.running_mean)) - m.running_mean * s


def describe_guidance(net):
    """Recognise f_phi (device-free, like card_amd.sampler.describe): walk the `.network` Sequential of a
    DeterministicFeedForwardNeuralNetwork (regression/model.py:70-79) -- or any Sequential of Linear / BatchNorm1d /
    LeakyReLU / ReLU / Dropout / Identity -- and return {"dims": [dim_in, hidden..., dim_out], "weights": [Linear.weight
    ...], "biases": [...], "batchnorm": [the BatchNorm1d module that follows layer l, or None], "negative_slope": float}
    with the module's own Parameter objects.  Raises CardError for train-mode BatchNorm / Dropout and foreign layers."""
    import torch.nn as nn
    net = getattr(net, "module", net) if isinstance(net, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else net
    seq = getattr(net, "network", net)
    Ws, bs, bns, slope = [], [], [], None
    for m in seq.children():
        if isinstance(m, nn.Linear):
            Ws.append(m.weight)
            bs.append(m.bias)
            bns.append(None)
        elif isinstance(m, nn.BatchNorm1d):
            if m.training or not m.track_running_stats:
                raise CardError("f_phi BatchNorm1d must be in eval mode with running statistics (the reference calls "
                                "cond_pred_model.eval() before testing, regression/card_regression.py:951)")
            if not Ws or bns[-1] is not None:
                raise CardError("f_phi: BatchNorm1d must follow a Linear layer")
            bns[-1] = m
        elif isinstance(m, nn.LeakyReLU):
            if slope is not None and slope != m.negative_slope:
                raise CardError("f_phi: LeakyReLU layers with different slopes")
            slope = m.negative_slope
        elif isinstance(m, nn.ReLU):
            if slope not in (None, 0.0):
                raise CardError("f_phi: mixed activation slopes")
            slope = 0.0
        elif isinstance(m, nn.Dropout):
            if m.training and m.p > 0:
                raise CardError("f_phi Dropout must be in eval mode (identity); call .eval() on the guidance model")
        elif isinstance(m, nn.Identity):
            pass
        else:
            raise CardError(f"f_phi: layer type {type(m).__name__} is not part of regression/model.py:70-79")
    if not Ws:
        raise CardError("f_phi: no Linear layer found")
    if bns[-1] is not None:
        raise CardError("f_phi: BatchNorm after the output layer is not part of regression/model.py:70-79")
    dims = [int(Ws[0].shape[1])]
    for l, W in enumerate(Ws):
        if W.dim() != 2 or int(W.shape[1]) != dims[-1]:
            raise CardError(f"f_phi layer {l}: weight {tuple(W.shape)} does not take {dims[-1]} inputs")
        dims.append(int(W.shape[0]))
    return {"dims": dims, "weights": Ws, "biases": bs, "batchnorm": bns, "negative_slope": 0.01 if slope is None else float(slope)}


_gcache = weakref.WeakKeyDictionary()


def guidance_for(f_phi):
    """The cached GuidanceNet of an `nn.Module` (DeterministicFeedForwardNeuralNetwork), or of an OLS coefficient tensor
    (1 + x_dim, y_dim); a GuidanceNet passes through.  Re-packed when the weights change (identity + content check, like
    sampler_for)."""
    if isinstance(f_phi, GuidanceNet):
        return f_phi
    if torch.is_tensor(f_phi):
        return GuidanceNet.from_ols(f_phi)                 # a (1 + x_dim) x y_dim matrix: packing it is one tiny copy
    fp = (_fingerprint(f_phi), bool(getattr(f_phi, "training", False)))
    hit = _gcache.get(f_phi)
    h = None
    if hit is not None and hit[0] == fp:
        h = _content_hash(f_phi)
        if h == hit[2]:
            return hit[1]
    g = GuidanceNet.from_module(f_phi)
    _gcache[f_phi] = (fp, g, _content_hash(f_phi) if h is None else h)
    return g
