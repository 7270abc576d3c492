// card_internal.h - the sampler handle and the per-variant back ends behind the C ABI.
#ifndef CARD_INTERNAL_H
#define CARD_INTERNAL_H
#include "card_common.h"

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
};

struct card_sampler {
    int variant, x_dim, y_dim, hidden, n_embed, cat_x, cat_y_pred;
    int d_in;
    int device, num_cu;
    int backend;   // 0: persistent kernel (card_guided.hip), 1: streaming GEMMs (card_wide.hip), 2: persistent kernel with streamed weights (card_stream.hip), 3: cooperative classification kernel (card_clscoop.hip)
    // ---- CARD_VARIANT_GUIDED (card_guided.hip)
    int k1, hoist, nqs;
    int coop_force;
    int coop_blocks;   // > 0: workgroups per CU of the cooperative small-batch kernel (0: not available)
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

bool stream_supports(const card_model_desc* d, int d_in);
int stream_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void stream_destroy(card_sampler* s);
int stream_run(const card_sampler* s, const CardRun& r, hipStream_t stream);

// card_fphi.hip: the guidance network evaluated at the head of a sampling call
int fphi_in_dim(const card_fphi* f);
int fphi_out_dim(const card_fphi* f);
int fphi_forward(const card_fphi* f, const float* x, long long rows, float* out, hipStream_t stream);

bool clscoop_supports(const card_model_desc* d);
int clscoop_create(card_sampler* s, const card_model_desc* d, hipStream_t stream);
void clscoop_destroy(card_sampler* s);
int clscoop_run(const card_sampler* s, const CardRun& r, hipStream_t stream);
#endif
