// card_internal.h - the sampler handle and the per-variant back ends behind the C ABI.
#ifndef CARD_INTERNAL_H
#define CARD_INTERNAL_H
#include "card_common.h"

// What card_sampler_plan reports.  A plan is a DRY RUN of the back end's run function: the same code takes the same
// decisions and, where it would launch, only notes the kernel (CARD_LAUNCH below), so a plan cannot drift from the dispatch.
struct CardPlanRec {
    int launches, dominant_launches;
    char kernel[96];
    double executed;        // matrix-pipe + lin4 flops the kernels issue per sample-step
};
inline bool card_plan_note(CardPlanRec* pl, const char* name, bool dominant) {
    if (!pl) return false;
    pl->launches++;
    if (dominant) {
        pl->dominant_launches++;
        if (!pl->kernel[0]) snprintf(pl->kernel, sizeof(pl->kernel), "%s", name);
    }
    return true;
}
// CARD_LAUNCH(r, "kernel name", carries_the_flops, kernel<<<grid, block, lds, stream>>>(args));
#define CARD_LAUNCH(r, name, dominant, ...)                      \
    do {                                                         \
        if (!card_plan_note((r).plan, name, dominant)) {         \
            __VA_ARGS__;                                         \
            HIP_TRY(hipGetLastError());                          \
        }                                                        \
    } while (0)

// Form switches, for tests and A/B runs: ONE environment variable per back end, a comma-separated token list, read ONCE
// when a sampler is created (card_tune_from_env, card_api.hip) and kept in the sampler; a run reads no environment.
//   CARD_AMD_COOP     regression, hidden 128:  off (0) = persistent kernel for every batch size | force (2) = cooperative
//                     kernel for every batch size | no16 = no 16-chain-tile form | no2 = no time-sliced paired form
//   CARD_AMD_STREAM   regression, hidden 256:  off (0) = GEMM back end instead | big = eight-wave ring form only |
//                     nosplit = never a tile shared by four waves | noslice = no time-sliced paired form
//   CARD_AMD_CLSCOOP  classification, F = 128: off (0) = GEMM back end instead | noslice = no round-robin time slicing
//   CARD_AMD_WIDE     GEMM back end:           gemmhead = lin4 as a GEMM launch also up to 32 classes | notail = no 64 x 64 tail tiles
//   CARD_AMD_ARITH    split = CARD_ARITH_SPLIT_BF16 for calls that leave card_run_args.arith at 0
struct CardTune {
    int coop_off, coop_force, coop_no16, coop_no2;
    int stream_off, stream_big, stream_nosplit, stream_noslice;
    int cls_off, cls_noslice;
    int wide_gemmhead, wide_notail;
    int arith_split;
};
void card_tune_from_env(CardTune* t);

// Launch-level arguments shared by both back ends (validated by card_sampler_run / card_eps_theta).
struct CardRun {
    const float* x;         // GUIDED: x (N, x_dim);  BN_HADAMARD: x_feat (N, F)
    const float* y0hat;
    const float* ytmean;
    const float* y_init;
    const float* coef;      // (T, 8)
    const float* noise;
    float* out;
    const int* state_slot;
    void* workspace;
    unsigned long long workspace_bytes;
    long long N;
    unsigned long long seed, elem_offset;
    int T, t_hi, t_lo, out_mode, eps_only;
    int n_z;                // >= 1: chains per input row
    int arith;              // card_arith
    int lib_scratch;        // card_sampler_run_static: `workspace` is the library's per-GPU static array, to be used
                            // for every scratch buffer of the back end (also in place of the sampler's own slots)
    CardPlanRec* plan;      // not NULL: dry run for card_sampler_plan (no launch, no device access)
};

struct card_sampler {
    int variant, x_dim, y_dim, hidden, n_embed, cat_x, cat_y_pred;
    int d_in;
    int device, num_cu;
    CardTune tune; // form switches read from the environment at create time
    int backend;   // 0: persistent kernel (card_guided.hip), 1: streaming GEMMs (card_wide.hip), 2: persistent kernel with streamed weights (card_stream.hip), 3: cooperative classification kernel (card_clscoop.hip)
    // ---- CARD_VARIANT_GUIDED (card_guided.hip)
    int nqs;           // k-step groups of the hoisted static columns of W1 (card_clscoop.hip: of the class columns)
    int k1;            // card_stream.hip: k-steps of layer 1
    int coop_blocks;   // > 0: workgroups per CU of the cooperative small-batch kernel (0: not available); card_stream.hip: of its shared-tile form
    float* wpk;    // packed LDS image
    float* gtab;   // (n_embed, 3, H) gamma * log2e [* bn_scale]
    float* w1s;    // hoist: packed static columns of W1
    float* ws;     // hoist: per-wave scratch slots (num_cu * 8 * 16 KB)
    float* wpk16;  // 16-chain-tile cooperative kernel: W2 | W3 in 16x16x4 fragment order
    float* w1s16;  // ... static columns of W1
    unsigned short* wsplit;   // card_split.hip: bf16 parts of W2 | W3 in A-fragment order (also uses W1 below, raw)
    void* kernel;         // the back end's __global__ function for this model shape (cast back by its owner)
    size_t lds_bytes;
    // ---- GEMM back end (card_wide.hip): BN_HADAMARD, LINEAR_HADAMARD, and GUIDED at hidden != 128 or y_dim != 1
    float* W1;     // (F, d_in)
    float* W2;     // (F, F)
    float* W3;     // (F, F)
    float* W4;     // (C, F)
    float* bias;   // b1 | b2 | b3 (3F) then b4 (C)
    float* shift;  // bn_shift * log2e, (3, F)
};

int guided_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void guided_destroy(card_sampler* s);
int guided_run(const card_sampler* s, const CardRun& r, hipStream_t stream);

// card_split.hip: CARD_ARITH_SPLIT_BF16 for backend 0 (shares wpk's small tables, gtab)
int split_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void split_destroy(card_sampler* s);
int split_run(const card_sampler* s, const CardRun& r, hipStream_t stream);

int wide_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void wide_destroy(card_sampler* s);
int wide_run(const card_sampler* s, const CardRun& r, hipStream_t stream);
unsigned long long wide_workspace_bytes(const card_sampler* s, long long N);

bool stream_supports(const card_model_desc* d, int d_in, const CardTune& tune);
int stream_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void stream_destroy(card_sampler* s);
int stream_run(const card_sampler* s, const CardRun& r, hipStream_t stream);

// card_fphi.hip: the guidance network evaluated at the head of a sampling call
int fphi_in_dim(const card_fphi* f);
int fphi_out_dim(const card_fphi* f);
int fphi_forward(const card_fphi* f, const float* x, long long rows, float* out, hipStream_t stream);

bool clscoop_supports(const card_model_desc* d, const CardTune& tune);
int clscoop_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void clscoop_destroy(card_sampler* s);
int clscoop_run(const card_sampler* s, const CardRun& r, hipStream_t stream);
#endif
