"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/card_hip.h declares,
and its host-only entry point (coefficient table) matches the oracle.  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from card_amd import _lib
from card_amd.sampler import make_coef_table
from oracle import card_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "card_hip.h")).read()
    return re.findall(r"CARD_API\s+[\w\s\*]+?\b(card_\w+)\s*\(", text)


def test_library_loads_and_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 9 and set(names) == set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert _lib.lib().card_version() == 300


def test_struct_sizes_match_header_layout(tmp_path):
    """sizeof / offsetof of the ABI structs as gcc sees include/card_hip.h == the ctypes mirror in _lib.py."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "card_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(card_model_desc), sizeof(card_run_args),'
        ' offsetof(card_model_desc, W), offsetof(card_run_args, N), offsetof(card_run_args, out_mode),'
        ' offsetof(card_run_args, workspace_bytes), offsetof(card_run_args, fphi_out), sizeof(card_fphi_desc),'
        ' offsetof(card_fphi_desc, negative_slope), offsetof(card_fphi_desc, W), offsetof(card_fphi_desc, bn_shift),'
        ' sizeof(card_metric_args), sizeof(card_cls_metric_args), offsetof(card_cls_metric_args, rows),'
        ' offsetof(card_cls_metric_args, scale_T), offsetof(card_cls_metric_args, sample_probs),'
        ' sizeof(card_plan), offsetof(card_plan, kernel), offsetof(card_plan, flops_executed_per_sample_step));'
        ' return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    M, R, F, CM = _lib.ModelDesc, _lib.RunArgs, _lib.FphiDesc, _lib.ClsMetricArgs
    assert got == [ctypes.sizeof(M), ctypes.sizeof(R), M.W.offset, R.N.offset, R.out_mode.offset,
                   R.workspace_bytes.offset, R.fphi_out.offset, ctypes.sizeof(F), F.negative_slope.offset, F.W.offset,
                   F.bn_shift.offset, ctypes.sizeof(_lib.MetricArgs), ctypes.sizeof(CM), CM.rows.offset, CM.scale_T.offset,
                   CM.sample_probs.offset, ctypes.sizeof(_lib.Plan), _lib.Plan.kernel.offset,
                   _lib.Plan.flops_executed_per_sample_step.offset]


def test_error_reporting_without_gpu():
    L = _lib.lib()
    assert L.card_make_coef_table(None, None, 0, None) == -1
    assert b"card_make_coef_table" in L.card_last_error()
    with pytest.raises(_lib.CardError):
        _lib.check(-1, "x")


@pytest.mark.parametrize("sched", [("linear", 1000, 1e-4, 0.02), ("linear", 50, 1e-4, 0.4), ("cosine", 100, 1e-4, 0.02),
                                   ("quad", 100, 1e-4, 0.02), ("sigmoid", 100, 1e-4, 0.02), ("const", 20, 1e-4, 0.02)])
def test_coef_table_matches_ieee_restatement_and_oracle(sched):
    alphas, _, omabs = orc.make_schedule(*sched)
    got = make_coef_table(alphas, omabs).numpy()
    # (a) bit-exact against an IEEE fp32 numpy restatement of regression/diffusion_utils.py:71-80, :88
    f = np.float32
    a, s = alphas.numpy(), omabs.numpy()
    sab = np.sqrt(f(1) - s * s)
    want = np.zeros_like(got)
    want[:, 0], want[:, 1], want[:, 2], want[:, 7] = f(1) / sab, f(1) - sab, s, sab
    s2, sm, sabm, at = s[1:] * s[1:], s[:-1], sab[:-1], a[1:]
    want[1:, 3] = (f(1) - at) * sabm / s2
    want[1:, 4] = (sm * sm) * np.sqrt(at) / s2
    want[1:, 5] = f(1) + (sab[1:] - f(1)) * (np.sqrt(at) + sabm) / s2
    want[1:, 6] = np.sqrt((sm * sm) / s2 * (f(1) - at))
    assert np.array_equal(got, want)
    # (b) against the oracle's torch table: torch's vectorised CPU sqrt is not correctly rounded (about
    # 0.6 % of inputs are 1 ulp off), so the two agree to an ulp of the sqrt; gamma_2 = 1 + (...) amplifies
    # that by its cancellation, hence the absolute bound there
    ref = orc.make_coef_table(alphas, omabs).numpy()
    for c in (0, 1, 2, 3, 4, 6, 7):
        np.testing.assert_allclose(got[:, c], ref[:, c], rtol=4e-7, atol=2e-7)
    np.testing.assert_allclose(got[:, 5], ref[:, 5], rtol=0, atol=1e-6)


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import card_amd
    m = torch.nn.Linear(2, 2)
    with pytest.raises(_lib.CardError):
        card_amd.p_sample_loop(m.eval(), torch.zeros(1, 1), torch.zeros(1, 1), torch.zeros(1, 1), 10,
                               torch.ones(10), torch.ones(10))


def test_missing_library_fails_loudly(tmp_path):
    """No built library -> CardError on first use, never a CPU / eager fallback."""
    import subprocess, sys, os
    code = ("import torch, card_amd\n"
            "from card_amd._lib import CardError\n"
            "try:\n"
            "    card_amd.sampler.make_coef_table(torch.ones(4), torch.full((4,), 0.5))\n"
            "except CardError as e:\n"
            "    print('CardError:', e)\n"
            "else:\n"
            "    raise SystemExit('no error raised')\n")
    env = dict(os.environ, CARD_AMD_LIB=str(tmp_path / "nope.so"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "CardError" in out.stdout, out.stdout + out.stderr
