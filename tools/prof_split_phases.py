#!/usr/bin/env python3
"""Cycles per interval slot of card_sample_split_kernel (profile build: make EXTRA=-DCARD_SPLIT_PROFILE OUT=..., run with
CARD_AMD_LIB): workgroup 0's eight waves (0-3 front: layer 2; 4-7 back: update + layer 1, layer 3 + lin4), work / barrier
wait of the three slots of an interval, summed over a call."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule
dev, T, x_dim = "cuda:0", 1000, 13
rows, n_z = int(os.environ.get("ROWS", 245)), 100
s = Sampler.from_state_dict(guided_state_dict(x_dim, T), device=dev)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
g = torch.Generator().manual_seed(0)
x, y0 = torch.randn(rows, x_dim, generator=g).to(dev), torch.randn(rows, 1, generator=g).to(dev)
L = _lib.lib()
buf = (ctypes.c_longlong * 64)()
s.run(x, y0, y0, coef, T, seed=1, n_z=n_z, arith=_lib.ARITH_SPLIT_BF16); torch.cuda.synchronize()
L.card_debug_split_prof(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); s.run(x, y0, y0, coef, T, seed=2, n_z=n_z, arith=_lib.ARITH_SPLIT_BF16); e1.record(); torch.cuda.synchronize()
L.card_debug_split_prof(buf, 0)
print(json.dumps({"chains": rows * n_z, "ms": e0.elapsed_time(e1), "intervals": T + 1,
                  "cycles_per_interval_[work,wait]x3": {f"wave{w}": [round(buf[w * 8 + i] / (T + 1)) for i in range(6)] for w in range(8)}}))
