/*
 * card_hip.h - C ABI of libcard_hip.so: CARD's reverse-diffusion sampling path on MI355X (gfx950).
 *
 * The reference (XzwHan/CARD) is pure Python; it has no FFI of its own.  The "plugin" boundary for
 * this path is the set of module-level functions in its tree-local `diffusion_utils` module
 * (star-imported by the runners: regression/card_regression.py:16, classification/card_classification.py:19).
 * Each entry point below names the reference function it stands in for; `card_amd/diffusion_utils.py`
 * is the Python binding a maintainer drops in.
 *
 * Conventions
 *   - every pointer called "device" is a HIP device pointer on the current device; fp32, row-major, contiguous
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     never synchronises the device
 *   - return value: 0 = ok, negative = card_status; card_last_error() gives a thread-local message
 *   - a created sampler's weights are immutable; it also owns 16 KB of scratch per resident wave (32 MB) that
 *     every launch reuses, so calls on ONE sampler must not overlap in time (same stream, or caller-ordered
 *     streams).  Use one sampler per concurrent stream.
 *   - timestep index t runs T-1 ... 0 exactly like the reference's `reversed(range(1, n_steps))` + t = 0;
 *     "state index" s = T - t (s = 0 is y_T, s = T is y_0), i.e. position in the reference's y_p_seq list
 */
#ifndef CARD_HIP_H
#define CARD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARD_HIP_VERSION 300 /* major*10000 + minor*100 + patch */
#define CARD_API __attribute__((visibility("default")))

typedef enum card_status {
    CARD_OK = 0,
    CARD_E_INVALID = -1,     /* bad argument (null pointer, negative size, t out of range ...) */
    CARD_E_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
    CARD_E_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString) */
    CARD_E_NOMEM = -4
} card_status;

typedef enum card_variant {
    /* regression/model.py:21-51 ConditionalGuidedModel: cat(y_t, y_0_hat, x) -> 3 x (Linear, *gamma[t], softplus) -> Linear.
       hidden = 128, y_dim = 1 (every reference config): persistent kernel, weights resident in LDS, no workspace;
       hidden = 256, y_dim = 1, d_in <= 96: persistent kernel with weights streamed through LDS, no workspace;
       any other hidden that is a multiple of 128, any y_dim: streaming-GEMM back end (needs a workspace). */
    CARD_VARIANT_GUIDED = 0,
    /* classification/model.py:22-94 ConditionalModel after the hoisted encoder: eval-BatchNorm after each
       ConditionalLinear and a Hadamard product with x_feat = norm(encoder_x(x)) after layer 1.
       hidden = 128, y_dim <= 32 (MNIST / gaussmix configs): cooperative persistent kernel, no workspace;
       any other hidden that is a multiple of 128: streaming-GEMM back end (needs a workspace). */
    CARD_VARIANT_BN_HADAMARD = 1,
    /* classification/imagenet/card/builder.py:69-92 ConditionalModel (ImageNet head): no BatchNorm, no softplus, two
       conditional layers: lin2(gamma2[t]) o ( linx(z) * (gamma1[t] * lin1(cat(y_t, y_0_hat))) ) then lin3.
       Descriptor: W/b/gamma[0] = lin1, [1] = lin2, [2] unused (NULL); W4/b4 = lin3; x = linx(z) (N, hidden), which
       is step-invariant and therefore an input here. */
    CARD_VARIANT_LINEAR_HADAMARD = 2
} card_variant;

typedef enum card_out_mode {
    CARD_OUT_LAST = 0,  /* out is (N, y_dim): the last state the call produces (y_0 when t_lo == 0) */
    CARD_OUT_ALL = 1,   /* out is (n_states, N, y_dim): every state the call produces, in order */
    CARD_OUT_SELECT = 2 /* out is (n_slots, N, y_dim): state s goes to slot state_slot[s], skipped when < 0 */
} card_out_mode;

/* Arithmetic of the two hidden-width matrix products of a timestep (card_run_args.arith).  Everything else -- layer 1,
 * gamma / softplus, lin4, the posterior update, the noise -- is fp32 on the vector ALU in both.
 * NO COUNTERPART IN THE REFERENCE (its addmm is plain fp32). */
typedef enum card_arith {
    /* v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulation.  The default, and what every published number of this
       library is measured with unless it says otherwise. */
    CARD_ARITH_F32_MFMA = 0,
    /* Every fp32 operand is cut EXACTLY into three bf16 numbers (8 + 8 + 8 significant bits) and the product is formed
       from the six largest of the nine partial products on v_mfma_f32_32x32x16_bf16 (exact bf16 x bf16 products, fp32
       accumulation); the dropped terms are below 2^-24 |a||b|.  Measured error of a K = 128 dot product: below the
       FP32 MFMA chain's (tools/ubench/splitbf16.hip).  GUIDED samplers at hidden 128, y_dim 1 only (card_split.hip);
       CARD_E_UNSUPPORTED elsewhere.  Not bit-identical to CARD_ARITH_F32_MFMA (a different, equally accurate
       rounding); same RNG streams.  The environment variable CARD_AMD_ARITH=split (read when the sampler is created)
       selects it for calls that leave arith at 0. */
    CARD_ARITH_SPLIT_BF16 = 1
} card_arith;

/* Columns of one row of the (T, 8) coefficient table (card_make_coef_table). */
enum {
    CARD_COEF_INV_SQRT_AB = 0,
    CARD_COEF_ONE_M_SQRT_AB = 1,
    CARD_COEF_SOMAB = 2,
    CARD_COEF_G0 = 3,
    CARD_COEF_G1 = 4,
    CARD_COEF_G2 = 5,
    CARD_COEF_SIGMA = 6,
    CARD_COEF_SQRT_AB = 7,
    CARD_COEF_COLS = 8
};

/* Weights of the eps_theta network, exactly as the reference's state_dict holds them
 * (lin{1,2,3}.lin.weight (H,in), .lin.bias (H,), lin{1,2,3}.embed.weight (T+1,H),
 * lin4.weight (y_dim,H), lin4.bias (y_dim,)).  All device pointers; they are copied/packed at create time. */
typedef struct card_model_desc {
    int32_t struct_bytes; /* sizeof(card_model_desc), for forward compatibility */
    int32_t variant;      /* card_variant */
    int32_t x_dim;        /* GUIDED: columns of x (used when cat_x);  BN_HADAMARD: must equal hidden */
    int32_t y_dim;        /* 1 for regression, number of classes for classification */
    int32_t hidden;       /* H (128 for every reference regression model, regression/model.py:32-35); a multiple of 128 */
    int32_t n_embed;      /* rows of the gamma tables (= timesteps + 1, regression/model.py:24) */
    int32_t cat_x;        /* regression/model.py:25, GUIDED only */
    int32_t cat_y_pred;   /* regression/model.py:26; BN_HADAMARD: `guidance` (classification/model.py:31) */
    const float* W[3];    /* lin{1,2,3}.lin.weight */
    const float* b[3];    /* lin{1,2,3}.lin.bias */
    const float* gamma[3];/* lin{1,2,3}.embed.weight */
    const float* W4;      /* lin4.weight */
    const float* b4;      /* lin4.bias */
    const float* bn_scale[3]; /* BN_HADAMARD: weight / sqrt(running_var + eps) of unetnorm{1,2,3}; else NULL */
    const float* bn_shift[3]; /* BN_HADAMARD: bias - running_mean * scale; else NULL */
} card_model_desc;

typedef struct card_sampler card_sampler;

/* The guidance network f_phi that produces y_0_hat (and y_T_mean) from x, exactly as the reference's state_dict holds it.
 * Replaces: DeterministicFeedForwardNeuralNetwork.forward (regression/model.py:55-82; eval mode: Dropout is the identity,
 * BatchNorm1d is folded to a per-feature scale / shift by the caller) behind Diffusion.compute_guiding_prediction
 * (regression/card_regression.py:111-124, method "NN"); method "OLS" (`cat(1, x).mm(beta)`, :115-117) is the one-layer
 * case W = beta[1:].T, b = beta[0].  All device pointers; copied at create time. */
#define CARD_FPHI_MAX_LAYERS 8
#define CARD_FPHI_MAX_WIDTH 1024
typedef struct card_fphi_desc {
    int32_t struct_bytes;                        /* sizeof(card_fphi_desc) */
    int32_t n_layers;                            /* nn.Linear layers, output layer included: len(hid_layers) + 1 */
    int32_t dims[CARD_FPHI_MAX_LAYERS + 1];      /* dims[0] = dim_in, dims[l + 1] = out features of layer l, dims[n_layers] = dim_out */
    float negative_slope;                        /* LeakyReLU slope (regression/model.py:65, :76) */
    const float* W[CARD_FPHI_MAX_LAYERS];        /* (dims[l + 1], dims[l]) nn.Linear weight */
    const float* b[CARD_FPHI_MAX_LAYERS];        /* (dims[l + 1]) or NULL */
    const float* bn_scale[CARD_FPHI_MAX_LAYERS]; /* use_batchnorm (model.py:74-75), hidden layers: weight / sqrt(running_var + eps); else NULL */
    const float* bn_shift[CARD_FPHI_MAX_LAYERS]; /* bias - running_mean * scale; else NULL */
} card_fphi_desc;

typedef struct card_fphi card_fphi;
CARD_API int card_fphi_create(const card_fphi_desc* desc, void* stream, card_fphi** out);
CARD_API void card_fphi_destroy(card_fphi* f);
/* out (rows, dim_out) = f_phi(x (rows, dim_in)).  Replaces: compute_guiding_prediction (regression/card_regression.py:111-124). */
CARD_API int card_fphi_forward(const card_fphi* f, const float* x, int64_t rows, float* out, void* stream);

/* Library version (CARD_HIP_VERSION of the build). */
CARD_API int card_version(void);

/* Message of the last failing call on this thread ("" if none). */
CARD_API const char* card_last_error(void);

/* Host-side, no GPU: the per-timestep scalars p_sample derives from t
 * (regression/diffusion_utils.py:71-80, :88; p_sample_t_1to0 :97-98), same fp32 operation order.
 * alphas, one_minus_alphas_bar_sqrt: host arrays of T floats.  out: host array of T*8 floats. */
CARD_API int card_make_coef_table(const float* alphas, const float* one_minus_alphas_bar_sqrt, int32_t T, float* out);

/* Pack the weights of one eps_theta network into the kernel's LDS image and per-timestep gamma table.
 * Replaces: the `model` argument of p_sample / p_sample_loop (regression/diffusion_utils.py:57, :107). */
CARD_API int card_sampler_create(const card_model_desc* desc, void* stream, card_sampler** out);
CARD_API void card_sampler_destroy(card_sampler* s);

/* Arguments of one sampling call. */
typedef struct card_run_args {
    int32_t struct_bytes;   /* sizeof(card_run_args) */
    int32_t T;              /* n_steps of the reference loop; rows of coef */
    int32_t t_hi;           /* first timestep index processed (T-1 for a full loop) */
    int32_t t_lo;           /* last timestep index processed (0 for a full loop; 0 runs p_sample_t_1to0) */
    int64_t N;              /* rows = test rows x z-samples, flattened (row, z) */
    const float* x;         /* device (N, x_dim): GUIDED x (may be NULL when !cat_x); BN_HADAMARD x_feat (N, H) */
    const float* y_0_hat;   /* device (N, y_dim) */
    const float* y_T_mean;  /* device (N, y_dim) */
    const float* y_init;    /* device (N, y_dim) state at timestep t_hi+... : y_t fed to the first step.
                               NULL = start of a loop: y_T = noise draw 0 + y_T_mean (diffusion_utils.py:109-110);
                               requires t_hi == T-1 */
    const float* coef;      /* device (T, 8) from card_make_coef_table, 16-byte aligned */
    const float* noise;     /* device (n_draws, N, y_dim) injected N(0,1) draws in consumption order, or NULL
                               for the in-kernel Philox4x32-10 generator.  n_draws = (y_init ? 0 : 1) +
                               number of processed timesteps with t >= 1 */
    uint64_t seed;          /* Philox key (noise == NULL) */
    uint64_t elem_offset;   /* global index of element (0,0) of this call: keeps RNG streams independent of
                               how rows are sharded over GPUs / calls */
    float* out;             /* device, see card_out_mode */
    int32_t out_mode;
    int32_t n_z;            /* > 1: x, y_0_hat and y_T_mean hold N / n_z ROWS and chain i reads row i / n_z, i.e. the
                               reference's repeat-tiling of each test row into n_z chains
                               (regression/card_regression.py:1027-1035) happens inside the kernel.  0 or 1: they
                               hold N rows.  y_init, noise and out are always per chain. */
    const int32_t* state_slot; /* device (T+1) ints, CARD_OUT_SELECT only */
    void* workspace;        /* device scratch, 256-byte aligned: activations between the per-step GEMMs of the GEMM back
                               end (card_sampler_workspace_bytes reports a sufficient size; a smaller one makes the run
                               walk the batch in chunks of as many chains as fit, at least n_z).  NULL for samplers on
                               the persistent kernels (card_sampler_workspace_bytes reports 0). */
    uint64_t workspace_bytes;
    const card_fphi* fphi;  /* not NULL (GUIDED only): the call first evaluates y_0_hat = f_phi(x) over the N / max(n_z, 1) un-tiled
                               rows of x (x is then (rows, dim_in of f_phi) and must not be NULL) into fphi_out and samples with
                               it -- regression/card_regression.py:1021-1035 (compute_guiding_prediction, then the n_z tiling)
                               inside the call.  y_0_hat is ignored; y_T_mean == NULL means y_T_mean = y_0_hat (:1031), pass
                               zeros for config.diffusion.noise_prior (:1032-1033). */
    float* fphi_out;        /* device (rows, y_dim), required with fphi: receives y_0_hat = f_phi(x) (the caller's metrics need
                               it too, and so the call allocates nothing) */
    int32_t arith;          /* card_arith: how the hidden-layer products are formed (0 = the default, FP32 MFMA) */
    int32_t reserved0;      /* 0 */
} card_run_args;

/* Run timesteps t_hi ... t_lo of the reverse process for N independent chains (persistent kernels: one launch per
 * ~1M / ~262k chains for all timesteps; GEMM back end: four launches per timestep and chunk, five above 32 classes).
 * Replaces: p_sample_loop (regression/diffusion_utils.py:107-119, classification/diffusion_utils.py:119-144)
 * with y_init == NULL, t_hi = T-1, t_lo = 0;  p_sample (:57-90) with t_hi = t_lo = t >= 1;
 * p_sample_t_1to0 (:94-104) with t_hi = t_lo = 0. */
CARD_API int card_sampler_run(const card_sampler* s, const card_run_args* args, void* stream);

/* What a card_sampler_run call with these arguments WOULD launch, without launching or touching the device: a dry run of
 * the same dispatch code (batch-size thresholds, time slices, chunking), so it cannot drift from it.  NO COUNTERPART IN
 * THE REFERENCE; it exists so that a benchmark or a profiler script can name the kernel that carries the flops, count its
 * launches and price the flops it issues without re-implementing the library's dispatch.  args as for card_sampler_run
 * (validated the same way; pointers are not dereferenced). */
typedef struct card_plan {
    int32_t struct_bytes;            /* in: sizeof(card_plan) */
    int32_t backend;                 /* 0 hidden-128 regression kernels, 1 GEMM back end, 2 streamed hidden-256 kernel, 3 cooperative classification kernel */
    int32_t launches;                /* kernel launches of the call, all kinds */
    int32_t dominant_launches;       /* of them, launches that carry the matrix work (of `kernel` and, in a call cut into rounds, of its other forms) */
    char kernel[96];                 /* the kernel that carries the matrix work, with its form; a call that is cut into whole
                                        rounds of several forms names the form of its first, largest segment */
    double flops_per_sample_step;          /* ALGORITHMIC: 2 H (d_in + 2 H + y_dim), SURVEY.md section 8d */
    double flops_executed_per_sample_step; /* what the kernels issue: hoisted layer 1 once per tile and call, K / class padding priced in
                                              (CARD_ARITH_SPLIT_BF16: the bf16 flops of its six partial products) */
} card_plan;
CARD_API int card_sampler_plan(const card_sampler* s, const card_run_args* args, card_plan* out);

/* card_sampler_run for callers that cannot allocate device memory or pass a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same arguments and results as
 * card_sampler_run (args->workspace / workspace_bytes are ignored); every scratch buffer -- the per-wave layer-1 slots of
 * the persistent scratch-slot kernel, the chain states handed between the launches of the time-sliced cooperative kernels,
 * the activations of the GEMM back end -- lives in ONE array of
 * CARD_STATIC_SCRATCH_BYTES that the library holds on each GPU it runs on until the process ends.  The call allocates
 * nothing (no hipMalloc / hipFree, no allocating library routine).  Limits: on the GEMM back end the batch is walked in
 * chunks of as many chains as the array holds (12 483 at hidden 256, 1 004 at hidden 4096); if not even one input row's
 * n_z chains fit, the call fails with CARD_E_UNSUPPORTED.  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (different streams or threads) race -- order them.  The sampler's device must be current.
 * The FIRST call on a device looks the array's address up (hipGetSymbolAddress): a caller under HIP graph capture makes
 * one call on that device before the capture begins. */
#define CARD_STATIC_SCRATCH_BYTES (32u << 20)
CARD_API int card_sampler_run_static(const card_sampler* s, const card_run_args* args, void* stream);

/* 64-bit position-weighted checksum of n_arrays device arrays of 32-bit words (n_words[k] words each), written to the
 * device word *out on `stream`.  No reference counterpart: the Python binding uses it to notice weights that were
 * modified in place behind PyTorch's version counters (the reference's EMA copy, regression/ema.py:21, is
 * `param.data.copy_`) and re-pack the sampler.  arrays / n_words are host arrays; at most 32 arrays. */
CARD_API int card_content_hash(const void* const* arrays, const int64_t* n_words, int32_t n_arrays, uint64_t* out, void* stream);

/* Scratch bytes that suffice for card_sampler_run / card_eps_theta on N chains: 0 for samplers on the persistent
 * kernels; on the GEMM back end the activations of min(N, about 4 GiB worth of) chains -- larger batches run in
 * chunks, so the figure stops growing with N.  No reference counterpart: the reference lets PyTorch allocate its
 * activations; here the caller owns them so that a run never allocates. */
CARD_API int card_sampler_workspace_bytes(const card_sampler* s, int64_t N, uint64_t* bytes);

/* eps_theta forward only: out (N, y_dim) = model(x, y_t, y_0_hat, t).
 * Replaces: ConditionalGuidedModel.forward (regression/model.py:37-51) /
 * ConditionalModel.forward after the encoder (classification/model.py:82-94). */
CARD_API int card_eps_theta(const card_sampler* s, const float* x, const float* y_t, const float* y_0_hat, int64_t N,
                   int32_t t, float* out, void* workspace, uint64_t workspace_bytes, void* stream);

/* Forward process sample, regression/diffusion_utils.py:42-53:
 * out = sqrt_ab[t_i]*y + (1 - sqrt_ab[t_i])*y_0_hat + somab[t_i]*noise, per-row t (device int64 (N,)).
 * noise: device (N, y_dim) or NULL for Philox (seed, elem_offset). */
CARD_API int card_q_sample(const float* y, const float* y_0_hat, const float* alphas_bar_sqrt,
                  const float* one_minus_alphas_bar_sqrt, const int64_t* t, int64_t N, int32_t y_dim, int32_t T,
                  const float* noise, uint64_t seed, uint64_t elem_offset, float* out, void* stream);

/* Fill out (n_draws, N, y_dim) with exactly the normals the sampler draws in RNG mode (draw d0 + i in slab i).
 * No reference counterpart (the reference uses torch.randn_like); used to check RNG mode against injected mode. */
CARD_API int card_philox_normal(uint64_t seed, uint64_t elem_offset, int64_t n_elems, int32_t draw0, int32_t n_draws,
                       float* out, void* stream);

/* Per-row evaluation statistics of generated samples (y_dim = 1), computed on the device.
 * Replaces the numpy side of Diffusion.test: compute_prediction_SE (regression/card_regression.py:763-784),
 * compute_true_coverage_by_gen_QI (:786-811, per-row decile membership), compute_PICP (:813-825) and
 * compute_batch_NLL (:866-893).  Sums over rows (and the all-reduce over GPUs) are left to the caller. */
typedef struct card_metric_args {
    int32_t struct_bytes;
    int32_t n_z;            /* samples per row, <= 8192 */
    int64_t rows;
    int32_t n_states;       /* leading dimension of samples: metrics for several trajectory states at once */
    int32_t n_bins;         /* config.testing.n_bins (10) */
    const float* samples;   /* device (n_states, rows, n_z): chain order (row, z) of the sampler output */
    const float* y_true;    /* device (rows): normalised targets */
    float y_scale, y_shift; /* un-normalise v -> v*scale + shift for SE and NLL (scaler_y.inverse_transform); 1, 0 if none */
    float trim_lo, trim_hi; /* config.testing.trimmed_mean_range in percent: mean of sorted samples [int(lo/100*n_z), int(hi/100*n_z))
                               (regression/card_regression.py:773-776); (50, 50) = the median (:770-771); any other range
                               that keeps no sample is CARD_E_INVALID */
    float picp_lo, picp_hi; /* config.testing.PICP_range in percent (2.5, 97.5) */
    float tau;              /* > 0: global precision (set_NLL_global_precision); <= 0: per-row 1/var of the samples */
    float* se;              /* out (n_states, rows): squared error of the trimmed-mean prediction */
    float* nll;             /* out (n_states, rows) */
    int32_t* qice_bin;      /* out (n_states, rows): decile boundaries below y_true, 0 .. n_bins+1 */
    int32_t* picp_in;       /* out (n_states, rows): 1 if y_true lies inside the PICP interval */
} card_metric_args;

CARD_API int card_regression_row_metrics(const card_metric_args* args, void* stream);

/* Per-image evaluation statistics of generated classification samples (one workgroup per (state, image)).
 * Replaces the host side of compute_and_store_cls_metrics, classification/card_classification.py:540-603: for the n_z
 * generated y vectors of an image -- raw value -(y - 1)^2 (:529-536), softmax(raw / scale_T) (:552-554), the credible
 * interval of the raw values per class (nanquantile at PICP_range, :556-560), per-sample predicted labels and their vote
 * proportions (:562-569), the mean probabilities (:570), the majority vote (:572), the share of samples that predict the
 * true label (:574) and the paired two-sided t-test between the two most voted classes (:576-596, scipy ttest_rel).
 * Ties (equal vote counts) go to the lower class index, for the majority vote and for the runner-up of the t-test
 * (torch.argmax / argsort leave that order unspecified).  Reductions over images stay with the caller. */
typedef struct card_cls_metric_args {
    int32_t struct_bytes;
    int32_t n_z;              /* samples per image (config.testing.n_samples), <= 1024 */
    int32_t n_classes;        /* <= 1024; n_classes * (n_z + pow2ceil(n_z)) * 4 bytes must fit 144 KB of LDS */
    int32_t n_states;         /* leading dimension of samples */
    int64_t rows;             /* images */
    const float* samples;     /* device (n_states, rows, n_z, n_classes): chain order (image, z) of the sampler output */
    const int32_t* labels;    /* device (rows): true class of every image */
    float scale_T;            /* temperature of the softmax (self.tuned_scale_T; 1 when not tuned) */
    float ci_lo, ci_hi;       /* config.testing.PICP_range in percent (2.5, 97.5) */
    float ttest_alpha;        /* config.testing.ttest_alpha (0.05) */
    float* label_probs;       /* out (n_states, rows, n_classes): share of samples whose arg-max is that class */
    float* mean_probs;        /* out (n_states, rows, n_classes): mean softmax probability */
    float* ci;                /* out (n_states, rows, 2, n_classes): quantiles of the raw values */
    int32_t* majority_vote;   /* out (n_states, rows) */
    float* instance_accuracy; /* out (n_states, rows) */
    double* ttest_pvalue;     /* out (n_states, rows): NaN when the differences have no variance (as scipy) */
    int32_t* ttest_reject;    /* out (n_states, rows): pvalue < ttest_alpha */
    float* sample_probs;      /* out, optional (n_states, rows, n_z, n_classes): softmax probabilities of every sample, or NULL */
} card_cls_metric_args;

CARD_API int card_classification_row_metrics(const card_cls_metric_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARD_HIP_H */
