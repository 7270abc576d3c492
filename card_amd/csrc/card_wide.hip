// card_wide.hip - streaming-GEMM back end for CARD_VARIANT_BN_HADAMARD (classification ConditionalModel,
// classification/model.py:79-94 after the hoisted encoder), any feature width F that is a multiple of 128
// (128 toy/MNIST, 2048 FashionMNIST, 4096 CIFAR-10, 6144 CIFAR-100) and any class count C.
//
// F x F fp32 weights (64 MB at F = 4096) do not fit on chip, so one timestep is a short chain of launches
// over all N chains, activations round-tripping through HBM (N x F fp32; ~0.4 GB at the CIFAR-10 test
// shape, < 3 % of a step's time at 6 TB/s):
//     wide_gemm_kernel   : h1 = x_feat * softplus(BN1(gamma1[t] * (S . W1^T + b1))),  S = (y_t | y_0_hat | 0), K = 2C -> 32
//     wide_gemm_kernel   : h2 = softplus(BN2(gamma2[t] * (h1 . W2^T + b2)))        fp32 MFMA, fused epilogue
//     wide_gemm_kernel   : h3 = softplus(BN3(gamma3[t] * (h2 . W3^T + b3)))
//     wide_gemm_kernel   : eps = h3 . W4^T + b4      (W4 zero-padded to 128 rows: same kernel, bias epilogue)
//     wide_update_kernel : posterior update of y_t (regression/diffusion_utils.py:83-89), noise, outputs
// Eval-mode BatchNorm is the per-feature affine (scale, shift); scale and log2e are folded into the per-step
// gamma table at create time, so the epilogue is xs = fma(acc + b, g'[t], shift'); h = ln2 * softplus_l2(xs).
//
// GEMM: out[M x Nout] = A[M x K] . W[Nout x K]^T, both operands K-contiguous.  128 x 128 x 32 block tile, 4
// waves as 2 x 2, each wave 2 x 2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator VGPRs); LDS tiles are
// [128][32 + 4 pad] floats, so one ds_read_b128 at [row][8t + 4*(lane>>5)] feeds 4 consecutive MFMAs
// (k order inside a tile is permuted identically for both operands) and is bank-conflict free; global->LDS is
// register-staged one tile ahead (issue loads, compute, write the other stage, one barrier per tile).
#include "card_internal.h"

using namespace card;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int kStageFloats = (BM + BN) * LDT;
constexpr int kGemmLds = 2 * kStageFloats * (int)sizeof(float);   // 73 728 B -> two blocks per CU

enum { EPI_SOFTPLUS = 0, EPI_BIAS = 1, EPI_SOFTPLUS_MUL = 2, EPI_SCALE = 3, EPI_SCALE_MUL = 4 };

template <int EPI>
__global__ __launch_bounds__(256, 2) void wide_gemm_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ g,
                                                           const float* __restrict__ sh, const float* __restrict__ mul,
                                                           int mul_div, float* __restrict__ out, long long M, int K,
                                                           int ldo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hh = lane >> 5;
    const long long m0 = (long long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    // staging map: 8 threads cover one 128-byte row segment, 32 rows per pass, 4 passes per operand.  Both panels
    // are read through buffer resources (base = this block's first row, scalar k offset): the k loop carries no
    // per-lane address arithmetic, and rows past M read as zero (num_records) instead of being clamped.
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    const long long rows_a = M - m0 < BM ? M - m0 : BM;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * K), 0, (int)(rows_a * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (long long)n0 * K), 0, BN * K * 4, 0x00020000);
    int vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vo[i] = ((sr + 32 * i) * K + sc) * 4;
    f32x4 ra[4], rw[4];
    auto g_load = [&](int kt) {
        const int so = kt * BK * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo[i], so, 0));
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vo[i], so, 0));
        }
    };
    float* const st_base = sm + sr * LDT + sc;
    auto s_store = [&](int st) {
        float* As = st_base + st * kStageFloats;
        float* Ws = As + BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(As + 32 * i * LDT) = ra[i];
            *reinterpret_cast<f32x4*>(Ws + 32 * i * LDT) = rw[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const float* const a_base = sm + (wm * 64 + li) * LDT + 4 * hh;
    const float* const w_base = sm + BM * LDT + (wn * 64 + li) * LDT + 4 * hh;
    // one 32-deep k tile from LDS stage `st` (a literal at both call sites: every LDS offset is an immediate)
    auto compute = [&](int st) {
        const float* As = a_base + st * kStageFloats;
        const float* Ws = w_base + st * kStageFloats;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + 8 * t);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 32 * LDT + 8 * t);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ws + 8 * t);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Ws + 32 * LDT + 8 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
            }
        }
    };
    const int nk = K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {                        // two k tiles per trip: stage 0 then stage 1
        const bool more = kt + 2 < nk;
        g_load(kt + 1);
        compute(0);
        s_store(1);
        __syncthreads();
        if (more) g_load(kt + 2);
        compute(1);
        if (more) s_store(0);
        __syncthreads();
    }
    if (kt < nk) {                                         // odd tile count (layer 1: K = 32): last tile sits in stage 0
        compute(0);
        __syncthreads();
    }
    // Hadamard operand row of each chain row of this block (chain / n_z), one 64-bit division per row instead of
    // one per element: table in LDS (the tiles are dead after the loop's last barrier)
    int* mrow = reinterpret_cast<int*>(sm);
    if (EPI == EPI_SOFTPLUS_MUL || EPI == EPI_SCALE_MUL) {
        if (tid < BM) {
            long long r = m0 + tid;
            if (r > M - 1) r = M - 1;
            mrow[tid] = (int)(r / mul_div);
        }
        __syncthreads();
    }
    // epilogue: C/D map of the 32x32 tile: column = lane & 31 (feature), row = (r&3) + 8(r>>2) + 4(lane>>5) (chain)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn * 64 + b * 32 + li;
        const float bb = bias[col];
        float gg = 0.0f, ss = 0.0f;
        if (EPI != EPI_BIAS) { gg = g[col]; ss = sh[col]; }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const long long row = m0 + lr;
                float v = acc[a][b][r] + bb;
                if (EPI == EPI_SOFTPLUS || EPI == EPI_SOFTPLUS_MUL) v = kLn2 * softplus_l2(fmaf(v, gg, ss));
                if (EPI == EPI_SCALE || EPI == EPI_SCALE_MUL) v = gg * v;      // gamma[t] * lin(x), builder.py:63-65
                if (row < M) {
                    if (EPI == EPI_SOFTPLUS_MUL || EPI == EPI_SCALE_MUL) v *= mul[(long long)mrow[lr] * ldo + col];   // Hadamard with x_feat (model.py:87)
                    out[row * ldo + col] = v;
                }
            }
        }
    }
}

// Layer 1 (classification/model.py:82-87) is the same GEMM with K = Kp = 2C padded to a multiple of 32: its A
// operand is the state matrix S (N x Kp): columns [0, C) hold y_t (rewritten every step by the update kernel),
// columns [C, 2C) hold y_0_hat (guidance), the rest is zero.  This kernel writes the static columns once per run
// and, when y_src is given, the y_t columns too.
__global__ void wide_pack_state_kernel(const float* __restrict__ y_src, const float* __restrict__ yhat,
                                       const float* __restrict__ x, int C, int guidance, int x_cols, int Kp, int n_z,
                                       long long N, float* __restrict__ S) {
    const long long total = N * Kp;
    const int x0 = guidance ? 2 * C : C;               // GUIDED on this back end: cat(y_t, [y_0_hat], [x]), model.py:40-47
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / Kp;
        const int k = (int)(i - n * Kp);
        if (k < C) {
            if (y_src) S[i] = y_src[n * C + k];
        } else if (guidance && k < 2 * C) {
            S[i] = yhat[(n / n_z) * C + (k - C)];
        } else {
            S[i] = (k - x0 < x_cols) ? x[(n / n_z) * x_cols + (k - x0)] : 0.0f;
        }
    }
}

// y_T = draw 0 + y_T_mean (classification/diffusion_utils.py:123-124)
__global__ void wide_init_kernel(const float* __restrict__ mu, const float* __restrict__ noise, unsigned long long seed,
                                 unsigned long long elem_offset, long long n_elems, int C, int Kp, int n_z,
                                 float* __restrict__ S, float* __restrict__ out, const int* __restrict__ state_slot,
                                 long long fixed_slot, long long slot_stride) {
    const long long slot = state_slot ? (long long)state_slot[0] : fixed_slot;
    float* out0 = slot >= 0 ? out + slot * slot_stride : nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
         i += (long long)gridDim.x * blockDim.x) {
        float z;
        if (noise) {
            z = noise[i];
        } else {
            float zz[4];
            philox_normal4(seed, elem_offset + (unsigned long long)i, 0u, zz);
            z = zz[0];
        }
        const long long n = i / C;
        const int c = (int)(i - n * C);
        const float v = z + mu[(n / n_z) * C + c];
        S[n * Kp + c] = v;
        if (out0) out0[i] = v;
    }
}

// posterior update of every (chain, class) element for timestep t (classification/diffusion_utils.py:80-86, :98-100)
__global__ void wide_update_kernel(const float* __restrict__ eps, int ld_eps, const float* __restrict__ mu,
                                   const float* __restrict__ coef_row, const float* __restrict__ noise_d,
                                   unsigned long long seed, unsigned long long elem_offset, int draw, int t, int C,
                                   long long n_elems, int Kp, int n_z, float* __restrict__ S, float* __restrict__ out,
                                   const int* __restrict__ state_slot, int state, long long fixed_slot,
                                   long long slot_stride) {
    const long long slot = state_slot ? (long long)state_slot[state] : fixed_slot;
    float* out_slot = slot >= 0 ? out + slot * slot_stride : nullptr;
    const float c_inv = coef_row[CARD_COEF_INV_SQRT_AB], c_om = coef_row[CARD_COEF_ONE_M_SQRT_AB];
    const float c_s = coef_row[CARD_COEF_SOMAB], g0 = coef_row[CARD_COEF_G0], g1 = coef_row[CARD_COEF_G1];
    const float g2 = coef_row[CARD_COEF_G2], sig = coef_row[CARD_COEF_SIGMA];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / C;
        const int c = (int)(i - n * C);
        const float e = eps[n * ld_eps + c], m = mu[(n / n_z) * C + c], yy = S[n * Kp + c];
        const float y0r = c_inv * ((yy - c_om * m) - e * c_s);
        float v = y0r;
        if (t >= 1) {
            float z;
            if (noise_d) {
                z = noise_d[i];
            } else {
                float zz[4];
                philox_normal4(seed, elem_offset + (unsigned long long)i, (uint32_t)draw >> 2, zz);
                z = pick4(zz, draw & 3);
            }
            v = ((g0 * y0r + g1 * yy) + g2 * m) + sig * z;
        }
        S[n * Kp + c] = v;
        if (out_slot) out_slot[i] = v;
    }
}

__global__ void wide_copy_eps_kernel(const float* __restrict__ eps, int ld_eps, int C, long long n_elems,
                                     float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / C;
        out[i] = eps[n * ld_eps + (int)(i - n * C)];
    }
}

// create-time: gtab[t][l][f] = gamma_l[t][f] * bn_scale_l[f] * log2e ; shift'[l][f] = bn_shift_l[f] * log2e
__global__ void wide_pack_gamma_kernel(const float* g1, const float* g2, const float* g3, const float* s1,
                                       const float* s2, const float* s3, float factor, int n_embed, int F,
                                       float* __restrict__ gtab) {
    const long long total = (long long)n_embed * 3 * F;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(idx % F), l = (int)((idx / F) % 3);
        const long long t = idx / (3LL * F);
        const float* g = l == 0 ? g1 : (l == 1 ? g2 : g3);
        const float* s = l == 0 ? s1 : (l == 1 ? s2 : s3);
        gtab[idx] = g ? (s ? g[t * F + f] * s[f] : g[t * F + f]) * factor : 0.0f;
    }
}

// (rows x cols) -> (rows x ldp), zero padded
__global__ void wide_pad_cols_kernel(const float* __restrict__ in, int rows, int cols, int ldp, float* __restrict__ out) {
    const long long total = (long long)rows * ldp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ldp;
        const int c = (int)(i - r * ldp);
        out[i] = c < cols ? in[r * cols + c] : 0.0f;
    }
}

__global__ void wide_scale_kernel(const float* __restrict__ in, float scale, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i] * scale;
}

inline unsigned grid_for(long long n, int block = 256, long long cap = 8192) {
    long long g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

inline int state_ld(const card_sampler* s) { return (s->d_in + BK - 1) / BK * BK; }

struct WideWs {
    float *hA, *hB, *eps, *S;
    size_t bytes;
};
WideWs carve(void* base, long long N, int F, int Kp, int Cpad) {
    WideWs w;
    size_t off = 0;
    char* p = (char*)base;
    w.hA = (float*)(p + off); off += align256((size_t)N * F * 4);
    w.hB = (float*)(p + off); off += align256((size_t)N * F * 4);
    w.eps = (float*)(p + off); off += align256((size_t)N * Cpad * 4);
    w.S = (float*)(p + off); off += align256((size_t)N * Kp * 4);
    w.bytes = off;
    return w;
}

int gemm(int epi, const float* A, const float* W, const float* bias, const float* g, const float* sh, const float* mul,
         int mul_div, float* out, long long M, int K, int Nout, int ldo, hipStream_t stream) {
    dim3 grid((unsigned)(Nout / BN), (unsigned)((M + BM - 1) / BM));
    if (epi == EPI_SOFTPLUS)
        wide_gemm_kernel<EPI_SOFTPLUS><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo);
    else if (epi == EPI_SOFTPLUS_MUL)
        wide_gemm_kernel<EPI_SOFTPLUS_MUL><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo);
    else if (epi == EPI_SCALE)
        wide_gemm_kernel<EPI_SCALE><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo);
    else if (epi == EPI_SCALE_MUL)
        wide_gemm_kernel<EPI_SCALE_MUL><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo);
    else
        wide_gemm_kernel<EPI_BIAS><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

}  // namespace

// A run works through the batch in chunks of as many chains as the workspace holds (chunks start on a row
// boundary when rows are tiled in-kernel), so the workspace never has to grow with N: the size reported here
// covers min(N, ~4 GiB worth of chains), enough rounds of GEMM tiles per launch to fill the chip.
namespace {
inline size_t per_chain_bytes(const card_sampler* s) {
    const int Cpad = (s->y_dim + BN - 1) / BN * BN;
    return ((size_t)2 * s->hidden + Cpad + state_ld(s)) * 4;
}
inline long long gcd_ll(long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; }
}  // namespace

unsigned long long wide_workspace_bytes(const card_sampler* s, long long N) {
    const int Cpad = (s->y_dim + BN - 1) / BN * BN;
    long long chunk = (long long)((4ULL << 30) / per_chain_bytes(s)) / 16000 * 16000;   // multiple of 128 and of n_z = 100, 1000
    if (chunk < 16000) chunk = 16000;
    return carve(nullptr, N < chunk ? N : chunk, s->hidden, state_ld(s), Cpad).bytes;
}

int wide_create(card_sampler* s, const card_model_desc* d, hipStream_t stream) {
    const int F = d->hidden, C = d->y_dim;
    if (F < 128 || F % 128) return card_fail(CARD_E_UNSUPPORTED, "card_sampler_create: GEMM back end: hidden=%d must be a multiple of 128", F);
    const bool guided = d->variant == CARD_VARIANT_GUIDED;            // regression/model.py:37-51 at widths the persistent kernel lacks
    if (!guided && d->x_dim != F) return card_fail(CARD_E_INVALID, "card_sampler_create: BN_HADAMARD x_dim (%d) must equal hidden (%d): x is x_feat", d->x_dim, F);
    const bool linear = d->variant == CARD_VARIANT_LINEAR_HADAMARD;   // imagenet/card/builder.py:69-92
    for (int l = 0; l < 3 && !linear && !guided; ++l)
        if (!d->bn_scale[l] || !d->bn_shift[l]) return card_fail(CARD_E_INVALID, "card_sampler_create: BN_HADAMARD needs bn_scale/bn_shift (layer %d)", l + 1);
    const int d1 = s->d_in, Cpad = (C + BN - 1) / BN * BN;
    const size_t FF = (size_t)F * F;
    const int Kp = state_ld(s);
    HIP_TRY(hipMalloc(&s->W1, (size_t)F * Kp * 4));
    HIP_TRY(hipMalloc(&s->W2, FF * 4));
    if (!linear) HIP_TRY(hipMalloc(&s->W3, FF * 4));
    HIP_TRY(hipMalloc(&s->W4, (size_t)Cpad * F * 4));
    HIP_TRY(hipMalloc(&s->bias, (size_t)(3 * F + Cpad) * 4));
    HIP_TRY(hipMalloc(&s->shift, (size_t)3 * F * 4));
    HIP_TRY(hipMalloc(&s->gtab, (size_t)d->n_embed * 3 * F * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)wide_gemm_kernel<EPI_SOFTPLUS>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    HIP_TRY(hipFuncSetAttribute((const void*)wide_gemm_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    HIP_TRY(hipFuncSetAttribute((const void*)wide_gemm_kernel<EPI_SOFTPLUS_MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    HIP_TRY(hipFuncSetAttribute((const void*)wide_gemm_kernel<EPI_SCALE>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    HIP_TRY(hipFuncSetAttribute((const void*)wide_gemm_kernel<EPI_SCALE_MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    wide_pad_cols_kernel<<<grid_for((long long)F * Kp), 256, 0, stream>>>(d->W[0], F, d1, Kp, s->W1);
    HIP_TRY(hipMemcpyAsync(s->W2, d->W[1], FF * 4, hipMemcpyDeviceToDevice, stream));
    if (!linear) HIP_TRY(hipMemcpyAsync(s->W3, d->W[2], FF * 4, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemsetAsync(s->W4, 0, (size_t)Cpad * F * 4, stream));
    HIP_TRY(hipMemcpyAsync(s->W4, d->W4, (size_t)C * F * 4, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemsetAsync(s->bias, 0, (size_t)(3 * F + Cpad) * 4, stream));
    HIP_TRY(hipMemsetAsync(s->shift, 0, (size_t)3 * F * 4, stream));
    for (int l = 0; l < (linear ? 2 : 3); ++l) {
        HIP_TRY(hipMemcpyAsync(s->bias + (size_t)l * F, d->b[l], (size_t)F * 4, hipMemcpyDeviceToDevice, stream));
        if (!linear && !guided) wide_scale_kernel<<<grid_for(F), 256, 0, stream>>>(d->bn_shift[l], kLog2e, F, s->shift + (size_t)l * F);
    }
    HIP_TRY(hipMemcpyAsync(s->bias + (size_t)3 * F, d->b4, (size_t)C * 4, hipMemcpyDeviceToDevice, stream));
    // per-step epilogue scale: gamma * bn_scale * log2e (softplus in log2 units) or plain gamma (linear head)
    wide_pack_gamma_kernel<<<grid_for((long long)d->n_embed * 3 * F), 256, 0, stream>>>(
        d->gamma[0], d->gamma[1], linear ? nullptr : d->gamma[2], (linear || guided) ? nullptr : d->bn_scale[0],
        (linear || guided) ? nullptr : d->bn_scale[1], (linear || guided) ? nullptr : d->bn_scale[2], linear ? 1.0f : kLog2e,
        d->n_embed, F, s->gtab);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

void wide_destroy(card_sampler* s) {
    float* bufs[] = {s->W1, s->W2, s->W3, s->W4, s->bias, s->shift, s->gtab};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
}

namespace {
// one chunk of chains [c0, c0 + r.N): r holds chunk-local pointers; `stride` = (chains of the whole run) * C is the
// distance between consecutive draws of `noise` and consecutive slots of `out`
int wide_run_chunk(const card_sampler* s, const CardRun& r, long long stride, hipStream_t stream) {
    const int F = s->hidden, C = s->y_dim, Cpad = (C + BN - 1) / BN * BN;
    const long long N = r.N, ne = N * C;
    const int Kp = state_ld(s);
    const WideWs w = carve(r.workspace, N, F, Kp, Cpad);
    const float* b1 = s->bias;
    const float* b2 = s->bias + F;
    const float* b3 = s->bias + 2 * (size_t)F;
    const float* b4 = s->bias + 3 * (size_t)F;
    const bool guided = s->variant == CARD_VARIANT_GUIDED;

    // one eps_theta forward from the state matrix S (y_t | y_0_hat | [x] | 0) at timestep t -> w.eps (N x Cpad)
    auto forward = [&](int t) -> int {
        const float* g = s->gtab + (size_t)t * 3 * F;
        int rc;
        if (s->variant == CARD_VARIANT_LINEAR_HADAMARD) {
            // lin3( gamma2 * lin2( linx(z) * (gamma1 * lin1(S)) ) ), imagenet/card/builder.py:86-92
            if ((rc = gemm(EPI_SCALE_MUL, w.S, s->W1, b1, g, s->shift, r.x, r.n_z, w.hA, N, Kp, F, F, stream))) return rc;
            if ((rc = gemm(EPI_SCALE, w.hA, s->W2, b2, g + F, s->shift + F, nullptr, 1, w.hB, N, F, F, F, stream))) return rc;
            return gemm(EPI_BIAS, w.hB, s->W4, b4, nullptr, nullptr, nullptr, 1, w.eps, N, F, Cpad, Cpad, stream);
        }
        if ((rc = gemm(guided ? EPI_SOFTPLUS : EPI_SOFTPLUS_MUL, w.S, s->W1, b1, g, s->shift, r.x, r.n_z, w.hA, N, Kp, F, F, stream))) return rc;
        if ((rc = gemm(EPI_SOFTPLUS, w.hA, s->W2, b2, g + F, s->shift + F, nullptr, 1, w.hB, N, F, F, F, stream))) return rc;
        if ((rc = gemm(EPI_SOFTPLUS, w.hB, s->W3, b3, g + 2 * (size_t)F, s->shift + 2 * (size_t)F, nullptr, 1, w.hA, N, F, F, F, stream))) return rc;
        return gemm(EPI_BIAS, w.hA, s->W4, b4, nullptr, nullptr, nullptr, 1, w.eps, N, F, Cpad, Cpad, stream);
    };
    // static columns of S (+ y_t when the caller supplies it)
    wide_pack_state_kernel<<<grid_for(N * Kp), 256, 0, stream>>>(r.y_init, r.y0hat, guided ? r.x : nullptr, C, s->cat_y_pred,
                                                                (guided && s->cat_x) ? s->x_dim : 0, Kp, r.n_z, N, w.S);
    HIP_TRY(hipGetLastError());

    if (r.eps_only) {
        if (int rc = forward(r.t_hi)) return rc;
        wide_copy_eps_kernel<<<grid_for(ne), 256, 0, stream>>>(w.eps, Cpad, C, ne, r.out);
        HIP_TRY(hipGetLastError());
        return CARD_OK;
    }
    const int T = r.T;
    const int base = r.y_init ? (T - r.t_hi) : 0;          // first draw consumed / first state produced
    // output slot of a state: fixed (CARD_OUT_ALL), looked up on the device (CARD_OUT_SELECT) or none
    const int* sel = r.out_mode == CARD_OUT_SELECT ? r.state_slot : nullptr;
    auto fixed = [&](int state) -> long long { return r.out_mode == CARD_OUT_ALL ? state - base : -1; };
    if (!r.y_init) {
        wide_init_kernel<<<grid_for(ne), 256, 0, stream>>>(r.ytmean, r.noise, r.seed, r.elem_offset, ne, C, Kp, r.n_z, w.S,
                                                          r.out, sel, fixed(0), stride);
        HIP_TRY(hipGetLastError());
    }
    for (int t = r.t_hi; t >= r.t_lo; --t) {
        if (int rc = forward(t)) return rc;
        const int draw = T - t;
        const float* nz = (r.noise && t >= 1) ? r.noise + (long long)(draw - base) * stride : nullptr;
        wide_update_kernel<<<grid_for(ne), 256, 0, stream>>>(w.eps, Cpad, r.ytmean, r.coef + (size_t)t * CARD_COEF_COLS, nz,
                                                            r.seed, r.elem_offset, draw, t, C, ne, Kp, r.n_z, w.S, r.out, sel, T - t,
                                                            fixed(T - t), stride);
        HIP_TRY(hipGetLastError());
    }
    if (r.out_mode == CARD_OUT_LAST) {
        wide_copy_eps_kernel<<<grid_for(ne), 256, 0, stream>>>(w.S, Kp, C, ne, r.out);   // y_t columns of S -> (N, C)
        HIP_TRY(hipGetLastError());
    }
    return CARD_OK;
}
}  // namespace

int wide_run(const card_sampler* s, const CardRun& r, hipStream_t stream) {
    const int C = s->y_dim, nz = r.n_z;
    const long long N = r.N;
    if (!r.workspace) return card_fail(CARD_E_INVALID, "card_sampler_run: this model runs on the GEMM back end and needs a workspace (card_sampler_workspace_bytes)");
    if (((uintptr_t)r.workspace & 255) != 0) return card_fail(CARD_E_INVALID, "card_sampler_run: workspace must be 256-byte aligned");
    // chains per chunk: what the workspace holds, cut to whole 128-row GEMM tiles and whole input rows
    const size_t pc = per_chain_bytes(s);
    const int Cpad = (C + BN - 1) / BN * BN;
    const bool fits = carve(nullptr, N, s->hidden, state_ld(s), Cpad).bytes <= r.workspace_bytes;
    const long long cap = r.workspace_bytes > 4 * 256 ? (long long)((r.workspace_bytes - 4 * 256) / pc) : 0;   // 4 buffers, 256-B aligned
    long long chunk = N;
    if (!fits) {
        const long long L = (long long)nz / gcd_ll(nz, BM) * BM;        // lcm(n_z, 128)
        chunk = cap >= L ? cap / L * L : cap / nz * nz;
        if (chunk < 1 && r.lib_scratch)
            return card_fail(CARD_E_UNSUPPORTED, "card_sampler_run_static: the static scratch of %llu bytes holds %lld chains of this model, fewer "
                             "than one input row's n_z=%d: beyond the no-allocation entry point's limit (use card_sampler_run with a workspace)",
                             r.workspace_bytes, cap, nz);
        if (chunk < 1)
            return card_fail(CARD_E_INVALID, "card_sampler_run: workspace of %llu bytes holds %lld chains, fewer than one input row's n_z=%d "
                             "(card_sampler_workspace_bytes reports a sufficient size)", r.workspace_bytes, cap, nz);
    }
    for (long long c0 = 0; c0 < N; c0 += chunk) {
        CardRun c = r;
        c.N = N - c0 < chunk ? N - c0 : chunk;
        const long long row0 = c0 / nz;                                  // chunks start on an input-row boundary
        const int xw = s->variant == CARD_VARIANT_GUIDED ? s->x_dim : s->hidden;
        if (r.x) c.x = r.x + row0 * xw;
        c.y0hat = r.y0hat + row0 * C;
        if (r.ytmean) c.ytmean = r.ytmean + row0 * C;
        if (r.y_init) c.y_init = r.y_init + c0 * C;
        if (r.noise) c.noise = r.noise + c0 * C;
        c.out = r.out + c0 * C;
        c.elem_offset = r.elem_offset + (unsigned long long)c0 * C;
        if (int rc = wide_run_chunk(s, c, N * C, stream)) return rc;
    }
    return CARD_OK;
}
