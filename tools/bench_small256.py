#!/usr/bin/env python3
"""Small-batch latency of one p_sample_loop call at hidden 256 (T = 1000, x_dim 13): the default dispatch (a tile split over
2 / 4 waves below num_cu * 32 / num_cu * 16 chains) against one wave per tile (CARD_AMD_STREAM=nosplit)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

dev, T, x_dim, H = "cuda:0", 1000, int(os.environ.get("XDIM", 13)), 256
sd = guided_state_dict(x_dim, T, hidden=H)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
samplers = {}
ALT = os.environ.get("ALT_MODE", "nosplit")      # the form switch of the second column (CARD_AMD_STREAM)
for mode in ("", ALT):
    os.environ["CARD_AMD_STREAM"] = mode
    samplers[mode] = Sampler.from_state_dict(sd, device=dev)
flops = 2 * H * (x_dim + 2 + 2 * H + 1)
g = torch.Generator().manual_seed(0)
for N in [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096, 6144, 8192, 12288, 16384]:
    x, y0 = torch.randn(N, x_dim, generator=g).to(dev), torch.randn(N, 1, generator=g).to(dev)
    row = {"chains": N, "hidden": H}
    for mode, s in samplers.items():
        s.run(x, y0, y0, coef, T, seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3):
            s.run(x, y0, y0, coef, T, seed=2 + i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        row["default_ms" if mode == "" else ("one_wave_per_tile_ms" if ALT == "nosplit" else ALT + "_ms")] = round(ms, 3)
    row["kernel"] = samplers[""].run(x, y0, y0, coef, T, seed=1, plan=True)["kernel"]
    row["frac_fp32_matrix_peak"] = round(N * T * flops / (row["default_ms"] * 1e-3) / 157.3e12, 4)
    print(json.dumps(row), flush=True)
