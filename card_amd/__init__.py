"""card_amd - MI355X-native reverse-diffusion sampler for CARD (XzwHan/CARD).

Drop-in for the reference's `diffusion_utils` hot path (p_sample_loop / p_sample / p_sample_t_1to0 /
q_sample), backed by libcard_hip.so (hand-written gfx950 HIP kernels behind a C ABI, include/card_hip.h).
There is no CPU or eager-PyTorch fallback: without the built library, or without a GPU, calls raise.
"""
from .diffusion_utils import (  # noqa: F401
    extract,
    kld,
    make_beta_schedule,
    p_sample,
    p_sample_loop,
    p_sample_loop_from_x,
    p_sample_loop_repeated,
    p_sample_loop_with_eval,
    p_sample_loop_with_cls_metrics,
    p_sample_loop_with_reg_metrics,
    p_sample_loop_with_y,
    p_sample_t_1to0,
    q_sample,
    y_0_reparam,
)
from .guidance import GuidanceNet, describe_guidance, guidance_for  # noqa: F401
from .sampler import ModelDescription, Sampler, describe, invalidate, sampler_for, unwrap  # noqa: F401

__version__ = "0.3.0"
