import sys, torch
sys.path.insert(0, '.')
import card_amd
from tests.helpers import load_golden, norm_rel_err
from tests.test_gpu_parity import model_of, cuda
g = load_golden("reg_boston_T50")
m, T = model_of(g), g["T"]
for t in range(T - 1, 0, -1):
    out = card_amd.p_sample(m, cuda(g["x"]), cuda(g["traj"][T - t]), cuda(g["y0hat"]), cuda(g["yTmean"]), t,
                            cuda(g["alphas"]), cuda(g["omabs"]), noise=cuda(g["noise"][T - t])).cpu()
    e = norm_rel_err(out, g["traj"][T - t + 1])
    print(t, f"{e:.3e}", out.flatten()[:3].tolist(), g["traj"][T-t+1].flatten()[:3].tolist())
