#!/usr/bin/env python3
"""Cycles per phase of the F = 128 classification kernel (profile build of the library: make ... EXTRA=-DCARD_CLS_PROFILE
OUT=card_amd/csrc/_prof/libcard_prof.so, run with CARD_AMD_LIB pointing at it): workgroup 0's four waves, summed over a
call.  Phases of one timestep of the workgroup's tile(s): layer 1 (+ noise) | barrier | layer 2 | barrier | layer 3 + lin4
| barrier | update | barrier."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import cls_state_dict, make_schedule

rows, n_z, C, T = int(os.environ.get("ROWS", 250)), int(os.environ.get("NZ", 100)), int(os.environ.get("C", 10)), 1000
dev = "cuda:0"
s = Sampler.from_cls_state_dict(cls_state_dict(128, C, T, 3), device=dev)
g = torch.Generator().manual_seed(1)
xf = torch.randn(rows, 128, generator=g).to(dev)
y0 = torch.softmax(torch.randn(rows, C, generator=g) * 2, 1).to(dev)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
L = _lib.lib()
buf = (ctypes.c_longlong * 64)()
s.run(xf, y0, y0, coef, T, seed=1, n_z=n_z); torch.cuda.synchronize()
L.card_debug_cls_prof(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); s.run(xf, y0, y0, coef, T, seed=2, n_z=n_z); e1.record(); torch.cuda.synchronize()
L.card_debug_cls_prof(buf, 0)
names = ["layer1+noise", "wait B1", "layer2", "wait B2", "layer3+lin4", "wait B3", "update", "wait B4"]
tot = [sum(buf[w * 8 + i] for w in range(4)) / 4 for i in range(8)]
print(json.dumps({"rows": rows, "n_z": n_z, "C": C, "ms": e0.elapsed_time(e1),
                  "per_wave": [[buf[w * 8 + i] for i in range(8)] for w in range(4)],
                  "share": {n: round(t / sum(tot), 4) for n, t in zip(names, tot)}}))
