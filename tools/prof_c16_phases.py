#!/usr/bin/env python3
"""Cycles per phase of a timestep of the 16-chain cooperative regression kernel (profile build of the library:
make -C card_amd/csrc EXTRA=-DCARD_COOP16_PROFILE OUT=$PWD/card_amd/csrc/_prof/libcard_prof.so OBJDIR=$PWD/card_amd/csrc/_prof/obj,
run with CARD_AMD_LIB pointing at it): workgroup 0's first tile, its four waves, averaged over a call.
Phases: layer 1 (MFMA, epilogue, publish) | barrier 1 | layer 2 | barrier 2 | layer 3 + lin4 | barrier 3 | update."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

dev, T, x_dim = "cuda:0", 1000, 13
s = Sampler.from_state_dict(guided_state_dict(x_dim, T), device=dev)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
L = _lib.lib()
names = ["layer1", "wait B1", "layer2", "wait B2", "layer3+lin4", "wait B3", "update"]
for N in [int(v) for v in sys.argv[1:]] or [1024, 8000]:
    g = torch.Generator().manual_seed(0)
    x, y0 = torch.randn(N, x_dim, generator=g).to(dev), torch.randn(N, 1, generator=g).to(dev)
    s.run(x, y0, y0, coef, T, seed=1); torch.cuda.synchronize()
    L.card_debug_c16_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.run(x, y0, y0, coef, T, seed=2); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    L.card_debug_c16_prof(buf, 0)
    steps = max(1, buf[7])
    per = [[round(buf[w * 8 + i] / steps, 1) for i in range(7)] for w in range(4)]
    avg = [round(sum(p[i] for p in per) / 4, 1) for i in range(7)]
    print(json.dumps({"chains": N, "kernel": s.run(x, y0, y0, coef, T, seed=1, plan=True)["kernel"], "ms": round(e0.elapsed_time(e1), 3),
                      "timesteps_counted": steps, "cycles_per_timestep": dict(zip(names, avg)), "sum": round(sum(avg), 1),
                      "per_wave": per}))
