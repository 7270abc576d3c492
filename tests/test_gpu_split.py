"""GPU parity of CARD_ARITH_SPLIT_BF16 (card_split.hip): the regression sampler with its two hidden layers on the bf16
matrix pipe at fp32 accuracy (every fp32 operand cut exactly into three bf16 parts, six partial products per k-step).

Same bar as the FP32-MFMA kernels (BASELINE.json north_star): every stored y_t of the reference's
trajectories within 1e-4 norm-relative under injected noise; the in-kernel generator must produce the same normals as
every other kernel (bit-equal y_T, outputs within rounding of the FP32-MFMA path on the same seed)."""
import pytest
import torch

from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule
from oracle import card_oracle as orc
from tests.helpers import golden_names, load_golden, norm_rel_err, elem_errs, fmt_errs

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"
SPLIT, F32 = _lib.ARITH_SPLIT_BF16, _lib.ARITH_F32_MFMA


def cuda(t):
    return t.to(DEV)


def sampler_of(g):
    return Sampler.from_state_dict(g["sd"], device=DEV, cat_x=g["cat_x"], cat_y_pred=g["cat_y_pred"])


@pytest.mark.parametrize("name", golden_names("reg"))
def test_free_running_loop_matches_reference(name):
    g = load_golden(name)
    s, T = sampler_of(g), g["T"]
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    traj = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), out_mode=_lib.OUT_ALL,
                 arith=SPLIT).cpu()
    assert torch.equal(traj[0], g["noise"][0] + g["yTmean"])
    errs = [elem_errs(traj[k], g["traj"][i]) for i, k in enumerate(g["traj_idx"])]
    ref32 = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), out_mode=_lib.OUT_ALL,
                  arith=F32).cpu()
    e32 = [elem_errs(ref32[k], g["traj"][i]) for i, k in enumerate(g["traj_idx"])]
    print(f"{name}: split {fmt_errs(tuple(max(e[j] for e in errs) for j in range(3)))} | "
          f"fp32 MFMA {fmt_errs(tuple(max(e[j] for e in e32) for j in range(3)))}")
    assert max(e[0] for e in errs) <= TOL
    # last-only and selected states give the same numbers
    last = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), arith=SPLIT).cpu()
    assert torch.equal(last, traj[T])
    sel = [0, T // 3, T - 1, T]
    got = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), out_mode=_lib.OUT_SELECT,
                states=sel, arith=SPLIT).cpu()
    assert torch.equal(got, traj[sel])


@pytest.mark.parametrize("name", ["reg_boston_T50", "reg_toy_T50", "reg_year_T50"])
def test_segments_and_single_steps(name):
    """p_sample-style calls (y_init, t_hi = t_lo = t) chained over the loop reproduce the one-call trajectory bit for bit."""
    g = load_golden(name)
    s, T = sampler_of(g), g["T"]
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    args = (cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T)
    traj = s.run(*args, noise=cuda(g["noise"]), out_mode=_lib.OUT_ALL, arith=SPLIT)
    y = traj[0]
    for t in range(T - 1, -1, -1):
        nz = cuda(g["noise"][T - t:T - t + 1]) if t >= 1 else None
        y = s.run(*args, t_hi=t, t_lo=t, y_init=y, noise=nz, arith=SPLIT)
        assert torch.equal(y, traj[T - t]), t
    # two segments through the in-kernel generator == one call
    one = s.run(*args, seed=11, arith=SPLIT)
    cut = T // 2 + 1
    mid = s.run(*args, t_lo=cut, seed=11, arith=SPLIT)
    two = s.run(*args, t_hi=cut - 1, y_init=mid, seed=11, arith=SPLIT)
    assert torch.equal(one, two)


@pytest.mark.parametrize("rows,n_z,x_dim", [(7, 100, 13), (300, 37, 1), (1200, 64, 90), (51, 1000, 13)])
def test_rng_mode_and_row_tiling(rows, n_z, x_dim):
    """In-kernel Philox: y_T bit-equal to the FP32-MFMA kernels' (same normals), every later state within rounding of
    theirs on the same seed; in-kernel row tiling (n_z) bit-equal to the pre-tiled call; ragged last tile / group."""
    T = 50
    sd = guided_state_dict(x_dim, T, hidden=128)
    s = Sampler.from_state_dict(sd, device=DEV)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.4)
    coef = cuda(make_coef_table(alphas, omabs))
    g = torch.Generator().manual_seed(5)
    x, y0 = cuda(torch.randn(rows, x_dim, generator=g)), cuda(torch.randn(rows, 1, generator=g))
    sel = [0, 1, T // 2, T]
    a = s.run(x, y0, y0, coef, T, seed=99, elem_offset=12345, n_z=n_z, out_mode=_lib.OUT_SELECT, states=sel, arith=SPLIT)
    b = s.run(x, y0, y0, coef, T, seed=99, elem_offset=12345, n_z=n_z, out_mode=_lib.OUT_SELECT, states=sel, arith=F32)
    assert torch.equal(a[0], b[0])
    assert torch.isfinite(a).all()
    for k in range(1, len(sel)):
        assert norm_rel_err(a[k].cpu(), b[k].cpu()) <= 2e-5, (k, norm_rel_err(a[k].cpu(), b[k].cpu()))
    xt, yt = x.repeat_interleave(n_z, 0), y0.repeat_interleave(n_z, 0)
    c = s.run(xt, yt, yt, coef, T, seed=99, elem_offset=12345, out_mode=_lib.OUT_SELECT, states=sel, arith=SPLIT)
    assert torch.equal(a, c)


def test_many_groups_per_workgroup_and_launch_boundary():
    """More chains than one launch covers (256 CUs x 96 groups x 3 tiles x 32 chains = 2 359 296): rows on either side of
    the launch boundary and of a group boundary equal the same rows sampled in a small call (elem_offset keeps the RNG
    streams) -- bit for bit, the arithmetic of a chain does not depend on its position."""
    T, n_z, x_dim = 8, 1000, 1
    sd = guided_state_dict(x_dim, T, hidden=128)
    s = Sampler.from_state_dict(sd, device=DEV)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.4)
    coef = cuda(make_coef_table(alphas, omabs))
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    per_launch = cu * 96 * 3 * 32
    rows = per_launch // n_z + 40
    g = torch.Generator().manual_seed(6)
    x, y0 = cuda(torch.randn(rows, x_dim, generator=g)), cuda(torch.randn(rows, 1, generator=g))
    big = s.run(x, y0, y0, coef, T, seed=3, n_z=n_z, arith=SPLIT).reshape(rows, n_z)
    assert torch.isfinite(big).all()
    for r0 in (0, per_launch // n_z - 1, rows - 3):
        small = s.run(x[r0:r0 + 3], y0[r0:r0 + 3], y0[r0:r0 + 3], coef, T, seed=3, n_z=n_z, elem_offset=r0 * n_z,
                      arith=SPLIT).reshape(3, n_z)
        assert torch.equal(big[r0:r0 + 3], small), r0


def test_unsupported_samplers_say_so():
    from card_amd.synth import cls_state_dict
    s = Sampler.from_cls_state_dict(cls_state_dict(128, 10, 20, 3), device=DEV)
    x, y0 = cuda(torch.randn(4, 128)), cuda(torch.softmax(torch.randn(4, 10), 1))
    alphas, _, omabs = make_schedule("linear", 20, 1e-4, 0.4)
    with pytest.raises(_lib.CardError, match="SPLIT_BF16"):
        s.run(x, y0, y0, cuda(make_coef_table(alphas, omabs)), 20, seed=1, arith=SPLIT)
