"""Error behaviour of the C ABI (include/card_hip.h): every entry point validates its arguments BEFORE it launches or
touches device memory, returns a negative card_status and leaves a message for card_last_error(); a rejected call writes
nothing, and the library keeps working afterwards.  (The reference raises torch RuntimeErrors from inside the model for
the same mistakes -- shape mismatches in regression/model.py:40-51, index errors in `extract`
regression/diffusion_utils.py:34-38; the shim turns a negative status into CardError, a RuntimeError.)"""
import ctypes as C

import pytest
import torch

from card_amd import _lib
from card_amd.sampler import CardError, Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, N, X_DIM = 20, 64, 5
E_INVALID, E_UNSUPPORTED = -1, -2


@pytest.fixture(scope="module")
def env():
    s = Sampler.from_state_dict(guided_state_dict(X_DIM, T), device=DEV)
    g = torch.Generator().manual_seed(3)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    t = dict(x=torch.randn(N, X_DIM, generator=g).to(DEV), y0=torch.randn(N, 1, generator=g).to(DEV),
             coef=torch.cat([make_coef_table(alphas, omabs).flatten(), torch.zeros(8)]).to(DEV),   # + slack for the misaligned case
             out=torch.full((N, 1), 7.0, device=DEV), slot=torch.zeros(T + 1, dtype=torch.int32, device=DEV))
    return s, t


def _args(s, t):
    a = _lib.RunArgs()
    a.struct_bytes = C.sizeof(_lib.RunArgs)
    a.T, a.t_hi, a.t_lo, a.N, a.n_z = T, T - 1, 0, N, 1
    a.x, a.y_0_hat, a.y_T_mean, a.coef, a.out = t["x"].data_ptr(), t["y0"].data_ptr(), t["y0"].data_ptr(), t["coef"].data_ptr(), t["out"].data_ptr()
    a.out_mode = _lib.OUT_LAST
    return a


def _msg():
    return _lib.lib().card_last_error().decode()


BAD_RUNS = [
    ("struct_bytes", lambda a, t: setattr(a, "struct_bytes", 8), E_INVALID, "struct_bytes"),
    ("negative N", lambda a, t: setattr(a, "N", -1), E_INVALID, "N < 0"),
    ("T = 0", lambda a, t: setattr(a, "T", 0), E_INVALID, "outside"),
    # (the gamma tables hold T + 1 rows like the reference's nn.Embedding(T + 1, H), regression/model.py:11)
    ("T beyond the gamma table", lambda a, t: (setattr(a, "T", T + 2), setattr(a, "t_hi", T + 1)), E_INVALID, "n_embed"),
    ("t_hi = T", lambda a, t: setattr(a, "t_hi", T), E_INVALID, "t_lo <= t_hi < T"),
    ("t_lo > t_hi", lambda a, t: setattr(a, "t_lo", T), E_INVALID, "t_lo <= t_hi < T"),
    ("loop start below T-1 without y_init", lambda a, t: setattr(a, "t_hi", T - 2), E_INVALID, "y_init"),
    ("out_mode", lambda a, t: setattr(a, "out_mode", 7), E_INVALID, "out_mode"),
    ("n_z does not divide N", lambda a, t: setattr(a, "n_z", 5), E_INVALID, "multiple"),
    ("null y_0_hat", lambda a, t: setattr(a, "y_0_hat", None), E_INVALID, "null"),
    ("null out", lambda a, t: setattr(a, "out", None), E_INVALID, "null"),
    ("misaligned coef", lambda a, t: setattr(a, "coef", t["coef"].data_ptr() + 4), E_INVALID, "16-byte"),
    ("null x", lambda a, t: setattr(a, "x", None), E_INVALID, "x is NULL"),
    ("SELECT without a slot table", lambda a, t: setattr(a, "out_mode", _lib.OUT_SELECT), E_INVALID, "state_slot"),
    ("unknown arith", lambda a, t: setattr(a, "arith", 5), E_INVALID, "arith"),
]


@pytest.mark.parametrize("name,mutate,code,needle", BAD_RUNS, ids=[b[0] for b in BAD_RUNS])
def test_rejected_run_writes_nothing(env, name, mutate, code, needle):
    s, t = env
    L = _lib.lib()
    t["out"].fill_(7.0)
    for fn in (L.card_sampler_run, L.card_sampler_run_static):
        a = _args(s, t)
        mutate(a, t)
        assert fn(s._h, C.byref(a), None) == code, name
        assert needle in _msg(), _msg()
    pl = _lib.Plan()
    pl.struct_bytes = C.sizeof(_lib.Plan)
    a = _args(s, t)
    mutate(a, t)
    assert L.card_sampler_plan(s._h, C.byref(a), C.byref(pl)) == code      # a plan validates like a run
    torch.cuda.synchronize()
    assert bool((t["out"] == 7.0).all())
    # ... and the library still runs the valid call
    a = _args(s, t)
    assert L.card_sampler_run(s._h, C.byref(a), None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(t["out"]).all() and not bool((t["out"] == 7.0).all())


def test_null_handles_and_other_entry_points(env):
    s, t = env
    L = _lib.lib()
    a = _args(s, t)
    assert L.card_sampler_run(None, C.byref(a), None) == E_INVALID and "null" in _msg()
    assert L.card_sampler_run(s._h, None, None) == E_INVALID
    n = C.c_uint64(0)
    assert L.card_sampler_workspace_bytes(s._h, -1, C.byref(n)) == E_INVALID
    y, out = t["y0"].data_ptr(), t["out"].data_ptr()
    assert L.card_eps_theta(s._h, t["x"].data_ptr(), y, y, N, T + 1, out, None, 0, None) == E_INVALID and "gamma table" in _msg()
    assert L.card_eps_theta(s._h, t["x"].data_ptr(), y, y, -1, 0, out, None, 0, None) == E_INVALID
    assert L.card_eps_theta(s._h, None, y, y, N, 0, out, None, 0, None) == E_INVALID and "x is NULL" in _msg()
    assert L.card_philox_normal(0, 0, -1, 0, 1, out, None) == E_INVALID
    assert L.card_philox_normal(0, 0, 4, 0, 1, None, None) == E_INVALID
    assert L.card_make_coef_table(None, None, 0, None) == E_INVALID


def test_create_rejects_what_the_kernels_are_not_built_for():
    L = _lib.lib()
    d = _lib.ModelDesc()
    h = C.c_void_p()
    assert L.card_sampler_create(None, None, C.byref(h)) == E_INVALID
    d.struct_bytes = 12
    assert L.card_sampler_create(C.byref(d), None, C.byref(h)) == E_INVALID and "struct_bytes" in _msg()
    d.struct_bytes = C.sizeof(_lib.ModelDesc)
    d.variant = 99
    assert L.card_sampler_create(C.byref(d), None, C.byref(h)) == E_INVALID and "variant" in _msg()
    d.variant = _lib.VARIANT_GUIDED
    d.n_embed, d.x_dim, d.y_dim, d.hidden = 10, 3, 1, 128
    assert L.card_sampler_create(C.byref(d), None, C.byref(h)) == E_INVALID and "null weight" in _msg()
    assert not h.value
    # the split-bf16 arithmetic exists for hidden 128 only: refused per call on a hidden-256 sampler, with the default still fine
    s = Sampler.from_state_dict(guided_state_dict(X_DIM, T, hidden=256), device=DEV)
    g = torch.Generator().manual_seed(1)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    coef = make_coef_table(alphas, omabs).to(DEV)
    x, y0 = torch.randn(N, X_DIM, generator=g).to(DEV), torch.randn(N, 1, generator=g).to(DEV)
    with pytest.raises(CardError, match=r"\(-2\).*SPLIT_BF16"):
        s.run(x, y0, y0, coef, T, arith=_lib.ARITH_SPLIT_BF16)
    assert torch.isfinite(s.run(x, y0, y0, coef, T, seed=1)).all()
    # the shim's own checks come first and name the argument
    with pytest.raises(CardError, match="expected shape"):
        s.run(x[:, :2], y0, y0, coef, T)
    with pytest.raises(CardError, match="rows"):
        s.run(x[:10], y0, y0, coef, T)
    with pytest.raises(CardError, match="twice"):
        s.run(x, y0, y0, coef, T, out_mode=_lib.OUT_SELECT, states=[1, 1])
