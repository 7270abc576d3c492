"""ctypes binding of libcard_hip.so (include/card_hip.h).  Fails loudly if the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CARD_AMD_LIB") or os.path.join(_HERE, "libcard_hip.so")  # override: kernel A/B builds

CARD_OK = 0
VARIANT_GUIDED, VARIANT_BN_HADAMARD, VARIANT_LINEAR_HADAMARD = 0, 1, 2
OUT_LAST, OUT_ALL, OUT_SELECT = 0, 1, 2
ARITH_F32_MFMA, ARITH_SPLIT_BF16 = 0, 1
COEF_COLS = 8

EXPORTS = (
    "card_version", "card_last_error", "card_make_coef_table", "card_sampler_create", "card_sampler_destroy",
    "card_sampler_run", "card_sampler_workspace_bytes", "card_eps_theta", "card_q_sample", "card_philox_normal",
    "card_regression_row_metrics", "card_classification_row_metrics", "card_sampler_run_static", "card_content_hash",
    "card_fphi_create", "card_fphi_destroy", "card_fphi_forward", "card_sampler_plan",
)
FPHI_MAX_LAYERS, FPHI_MAX_WIDTH = 8, 1024
STATIC_SCRATCH_BYTES = 32 << 20

_fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("variant", C.c_int32), ("x_dim", C.c_int32), ("y_dim", C.c_int32),
        ("hidden", C.c_int32), ("n_embed", C.c_int32), ("cat_x", C.c_int32), ("cat_y_pred", C.c_int32),
        ("W", _fp * 3), ("b", _fp * 3), ("gamma", _fp * 3), ("W4", _fp), ("b4", _fp),
        ("bn_scale", _fp * 3), ("bn_shift", _fp * 3),
    ]


class RunArgs(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("T", C.c_int32), ("t_hi", C.c_int32), ("t_lo", C.c_int32),
        ("N", C.c_int64), ("x", _fp), ("y_0_hat", _fp), ("y_T_mean", _fp), ("y_init", _fp), ("coef", _fp),
        ("noise", _fp), ("seed", C.c_uint64), ("elem_offset", C.c_uint64), ("out", _fp),
        ("out_mode", C.c_int32), ("n_z", C.c_int32), ("state_slot", _fp), ("workspace", _fp), ("workspace_bytes", C.c_uint64),
        ("fphi", _fp), ("fphi_out", _fp), ("arith", C.c_int32), ("reserved0", C.c_int32),
    ]


class Plan(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("backend", C.c_int32), ("launches", C.c_int32), ("dominant_launches", C.c_int32),
        ("kernel", C.c_char * 96), ("flops_per_sample_step", C.c_double), ("flops_executed_per_sample_step", C.c_double),
    ]


class FphiDesc(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("n_layers", C.c_int32), ("dims", C.c_int32 * (FPHI_MAX_LAYERS + 1)),
        ("negative_slope", C.c_float), ("W", _fp * FPHI_MAX_LAYERS), ("b", _fp * FPHI_MAX_LAYERS),
        ("bn_scale", _fp * FPHI_MAX_LAYERS), ("bn_shift", _fp * FPHI_MAX_LAYERS),
    ]


class MetricArgs(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("n_z", C.c_int32), ("rows", C.c_int64), ("n_states", C.c_int32),
        ("n_bins", C.c_int32), ("samples", _fp), ("y_true", _fp), ("y_scale", C.c_float), ("y_shift", C.c_float),
        ("trim_lo", C.c_float), ("trim_hi", C.c_float), ("picp_lo", C.c_float), ("picp_hi", C.c_float),
        ("tau", C.c_float), ("se", _fp), ("nll", _fp), ("qice_bin", _fp), ("picp_in", _fp),
    ]


class ClsMetricArgs(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("n_z", C.c_int32), ("n_classes", C.c_int32), ("n_states", C.c_int32),
        ("rows", C.c_int64), ("samples", _fp), ("labels", _fp), ("scale_T", C.c_float), ("ci_lo", C.c_float),
        ("ci_hi", C.c_float), ("ttest_alpha", C.c_float), ("label_probs", _fp), ("mean_probs", _fp), ("ci", _fp),
        ("majority_vote", _fp), ("instance_accuracy", _fp), ("ttest_pvalue", _fp), ("ttest_reject", _fp),
        ("sample_probs", _fp),
    ]


class CardError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CardError(
            f"{LIB_PATH} is missing: build it with `make -C card_amd/csrc` (or __graft_entry__.build()). "
            "card_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.card_version.restype = C.c_int
    L.card_last_error.restype = C.c_char_p
    L.card_make_coef_table.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]
    L.card_sampler_create.argtypes = [C.POINTER(ModelDesc), C.c_void_p, C.POINTER(C.c_void_p)]
    L.card_sampler_destroy.argtypes = [C.c_void_p]
    L.card_sampler_destroy.restype = None
    L.card_sampler_run.argtypes = [C.c_void_p, C.POINTER(RunArgs), C.c_void_p]
    L.card_sampler_run_static.argtypes = [C.c_void_p, C.POINTER(RunArgs), C.c_void_p]
    L.card_sampler_plan.argtypes = [C.c_void_p, C.POINTER(RunArgs), C.POINTER(Plan)]
    L.card_content_hash.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_void_p]
    L.card_sampler_workspace_bytes.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64)]
    L.card_eps_theta.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_int64, C.c_int32, _fp, _fp, C.c_uint64, C.c_void_p]
    L.card_q_sample.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_uint64,
                                C.c_uint64, _fp, C.c_void_p]
    L.card_regression_row_metrics.argtypes = [C.POINTER(MetricArgs), C.c_void_p]
    L.card_classification_row_metrics.argtypes = [C.POINTER(ClsMetricArgs), C.c_void_p]
    L.card_fphi_create.argtypes = [C.POINTER(FphiDesc), C.c_void_p, C.POINTER(C.c_void_p)]
    L.card_fphi_destroy.argtypes = [C.c_void_p]
    L.card_fphi_destroy.restype = None
    L.card_fphi_forward.argtypes = [C.c_void_p, _fp, C.c_int64, _fp, C.c_void_p]
    L.card_philox_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_void_p]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = a symbol the header declares is not exported
    _lib = L
    return L


def check(status, what):
    if status != CARD_OK:
        msg = lib().card_last_error().decode("utf-8", "replace")
        raise CardError(f"{what} failed ({status}): {msg}")
