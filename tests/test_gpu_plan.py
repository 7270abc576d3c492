"""card_sampler_plan (include/card_hip.h): a dry run of the dispatch code of card_sampler_run.  What it reports for the
BASELINE.json shapes -- the kernel that carries the flops, its launches, executed flops per sample-step -- and that a
plan launches nothing and writes nothing."""
import pytest
import torch

from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import cls_state_dict, guided_state_dict, make_schedule

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reg(x_dim, rows, hidden=128, T=1000):
    s = Sampler.from_state_dict(guided_state_dict(x_dim, T, hidden=hidden), device=DEV)
    g = torch.Generator().manual_seed(1)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    return s, torch.randn(rows, x_dim, generator=g).to(DEV), torch.randn(rows, 1, generator=g).to(DEV), make_coef_table(alphas, omabs).to(DEV)


def test_plan_names_the_kernels_of_the_baseline_shapes():
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert cu == 256, "launch counts below are those of a 256-CU MI355X"
    # configs[1] toy: 10 000 rows x 1000 z = 1e7 chains -> 10 launches of the persistent kernel, hoisted tile in LDS
    s, x, y0, coef = _reg(1, 10_000)
    p = s.run(x, y0, y0, coef, 1000, n_z=1000, plan=True)
    assert p["kernel"] == "card_sample_guided_kernel<2>" and p["launches"] == 10 and p["launches_all_kernels"] == 10
    assert p["flops_per_sample_step"] == 2 * 128 * (3 + 256 + 1) == 66560
    assert abs(p["flops_executed_per_sample_step"] - (2 * 128 * 256 + 2 * 128 * 2 + 2 * 128 + 2 * 128 * 8 / 1000)) < 1e-6
    # pre-tiled rows (n_z unknown to the kernel): the scratch-slot form
    xt = x[:100].repeat_interleave(1000, 0)
    assert s.run(xt, y0[:100].repeat_interleave(1000, 0), y0[:100].repeat_interleave(1000, 0), coef, 1000, plan=True)["kernel"] == "card_sample_guided_kernel<1>"
    # configs[0] Boston: 51 rows x 100 z = 319 sixteen-chain tiles -> two time slices of the 16-chain cooperative kernel
    s, x, y0, coef = _reg(13, 51)
    p = s.run(x, y0, y0, coef, 1000, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_coop16_kernel" and p["launches"] == 2
    # the reference's Year minibatch: 64 x 200 = 400 thirty-two-chain tiles -> time-sliced paired form
    s, x, y0, coef = _reg(90, 64)
    p = s.run(x, y0, y0, coef, 1000, n_z=200, plan=True)
    assert p["kernel"] == "card_sample_coop2_kernel" and 2 <= p["launches"] <= 8
    assert s.run(x, y0, y0, coef, 1000, n_z=200, arith=_lib.ARITH_SPLIT_BF16, plan=True)["kernel"] == "card_sample_split_kernel"
    # hidden 256: the streamed kernel cuts a call into whole rounds of its forms (card_stream.hip stream_plan): 40 000 chains =
    # one round of the eight-wave form + one of the shared-tile form; 16 400 = a round of the four-wave form + one tile;
    # 10 000 = two shared-tile workgroups per CU, then one; 4 000: a single launch
    s, x, y0, coef = _reg(13, 400, hidden=256, T=20)
    p = s.run(x, y0, y0, coef, 20, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_stream_kernel" and p["launches"] == 2
    p = s.run(x[:164], y0[:164], y0[:164], coef, 20, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_stream_kernel (4 waves)" and p["launches"] == 2
    p = s.run(x[:100], y0[:100], y0[:100], coef, 20, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_stream_kernel (4 waves, tile over 4 waves)" and p["launches"] == 2
    p = s.run(x[:40], y0[:40], y0[:40], coef, 20, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_stream_kernel (4 waves, tile over 4 waves)" and p["launches"] == 1


def test_plan_of_the_classification_heads_and_no_side_effects():
    T = 1000
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    coef = make_coef_table(alphas, omabs).to(DEV)
    g = torch.Generator().manual_seed(2)
    # MNIST head, one reference test batch: 782 tiles, time-sliced over two workgroups per CU, compact form
    s = Sampler.from_cls_state_dict(cls_state_dict(128, 10, T, 3), device=DEV)
    xf = torch.randn(250, 128, generator=g).to(DEV)
    y0 = torch.softmax(torch.randn(250, 10, generator=g), 1).to(DEV)
    p = s.run(xf, y0, y0, coef, T, n_z=100, plan=True)
    assert p["kernel"] == "card_sample_cls_coop_kernel<PAIR compact>" and p["launches"] >= 4
    # layer 1: 6 real k-steps of 2 classes, lin4 on the 16-class four-block MFMA (round 2's bench priced 32 columns here)
    assert abs(p["flops_executed_per_sample_step"] - (2 * 128 * (12 + 256 + 16) + 2 * 128 * 16 / T)) < 1e-6
    one = s.run(xf, y0, y0, coef, T, t_hi=7, t_lo=7, y_init=y0.repeat_interleave(100, 0), n_z=100, plan=True)
    assert one["kernel"] == "card_sample_cls_coop_kernel<PAIR compact, rounds>" and one["launches"] == 1
    # CIFAR-10 head: GEMM back end, two F x F launches per timestep carry the flops, four launches per timestep in all
    T2 = 5
    s = Sampler.from_cls_state_dict(cls_state_dict(1024, 10, T2, 3), device=DEV)
    a2, _, o2 = make_schedule("linear", T2, 1e-4, 0.3)
    c2 = make_coef_table(a2, o2).to(DEV)
    xf = torch.randn(64, 1024, generator=g).to(DEV)
    y0 = torch.softmax(torch.randn(64, 10, generator=g), 1).to(DEV)
    p = s.run(xf, y0, y0, c2, T2, n_z=4, plan=True)
    assert p["kernel"] == "wide_gemm_kernel<EPI_SOFTPLUS>" and p["launches"] == 2 * T2
    assert p["launches_all_kernels"] == 4 * T2 + 3          # + state packing, y_T initialisation, the final copy
    assert p["flops_executed_per_sample_step"] == 2 * 1024 * (32 + 2 * 1024 + 10)
    # a plan touches nothing: the output tensor of a planned call is never written
    torch.cuda.synchronize()
    a = _lib.RunArgs()
    canary = torch.full((256, 10), 7.0, device=DEV)
    import ctypes as C
    a.struct_bytes = C.sizeof(_lib.RunArgs)
    a.T, a.t_hi, a.t_lo, a.N, a.n_z = T2, T2 - 1, 0, 256, 4
    a.x, a.y_0_hat, a.y_T_mean, a.coef, a.out = xf.data_ptr(), y0.data_ptr(), y0.data_ptr(), c2.data_ptr(), canary.data_ptr()
    ws, a.workspace, a.workspace_bytes = s._workspace(256)
    pl = _lib.Plan()
    pl.struct_bytes = C.sizeof(_lib.Plan)
    _lib.check(_lib.lib().card_sampler_plan(s._h, C.byref(a), C.byref(pl)), "plan")
    torch.cuda.synchronize()
    assert bool((canary == 7.0).all()) and pl.launches == 4 * T2 + 3
    pl.struct_bytes = 8
    assert _lib.lib().card_sampler_plan(s._h, C.byref(a), C.byref(pl)) == -1
