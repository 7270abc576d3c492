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
//     wide_head_update_kernel : eps = h3 . W4^T + b4 and the posterior update of y_t (diffusion_utils.py:83-89), noise,
//                               outputs, in one VALU launch at memory speed (up to 32 classes; above that the head is a
//                               fourth wide_gemm_kernel launch over W4 zero-padded to 128 rows + wide_update_kernel)
// Eval-mode BatchNorm is the per-feature affine (scale, shift); scale and log2e are folded into the per-step
// gamma table at create time, so the epilogue is xs = fma(acc + b, g'[t], shift'); h = ln2 * softplus_l2(xs).
//
// GEMM: out[M x Nout] = A[M x K] . W[Nout x K]^T, both operands K-contiguous.  128 x 128 x 32 block tile, 4
// waves as 2 x 2, each wave 2 x 2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator VGPRs); LDS tiles are
// [128][32 + 4 pad] floats, so one ds_read_b128 at [row][8t + 4*(lane>>5)] feeds 4 consecutive MFMAs
// (k order inside a tile is permuted identically for both operands) and is bank-conflict free; global->LDS is
// register-staged one tile ahead into a single LDS stage (issue loads, compute, barrier, write, barrier); four workgroups
// per CU cover each other's barriers (see kGemmWgPerCu).
#include "card_internal.h"

using namespace card;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int kStageFloats = (BM + BN) * LDT;
// One LDS stage (36 KB: the next k tile waits in registers while this one is computed, two barriers per tile) and four
// workgroups per CU beat round 1's two stages / two workgroups per CU: four waves per SIMD keep the matrix pipe fed through
// each other's barriers (CIFAR-10 minibatch 12.24 -> 11.73 ms per timestep; three per CU 11.85).  128 VGPRs: the compiler
// spills 14 registers, all of them in the epilogue.  (The two-stage and three-workgroup builds of round 2 are gone.)
constexpr int kGemmWgPerCu = 4;
constexpr int kGemmLds = kStageFloats * (int)sizeof(float);   // 36 864 B

constexpr int kHeadFusedMaxC = 32;     // wide_head_update_kernel
enum { EPI_SOFTPLUS = 0, EPI_BIAS = 1, EPI_SOFTPLUS_MUL = 2, EPI_SCALE = 3, EPI_SCALE_MUL = 4 };

// TBM = TBN = 128: the 128 x 128 block tile (4 waves as 2 x 2, each 2 x 2 MFMA tiles).  TBM = TBN = 64: 64 x 64 (each wave one
// tile), same k loop, for the last, partly filled round of a launch: the reference's test minibatch of 250 images x 100 z
// is 196 row blocks x 32 column blocks = 6.125 rounds of the 1 024 resident workgroups, and 128 big tiles on an otherwise
// empty chip (one per CU: a single wave per SIMD hides neither the barriers nor the load latency) cost more than half a full
// round; as four times as many 64 x 64 tiles they cost about half that.  Every output element is the same fp32 FMA chain
// over k in either shape, so which shape computes a row changes no bits.
template <int EPI, int TBM, int TBN>
__device__ __forceinline__ void wide_gemm_tile(const float* __restrict__ A, const float* __restrict__ W,
                                               const float* __restrict__ bias, const float* __restrict__ g,
                                               const float* __restrict__ sh, const float* __restrict__ mul,
                                               int mul_div, float* __restrict__ out, long long M, int K,
                                               int ldo, long long m_first, int bx, long long by) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TM = TBM / 64, TN = TBN / 64;                // MFMA tiles per wave along the rows / columns
    constexpr int kStage = (TBM + TBN) * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hh = lane >> 5;
    const long long m0 = m_first + by * TBM;
    const int n0 = bx * TBN;
    // staging map: 8 threads cover one 128-byte row segment, 32 rows per pass, 4 passes per 128 rows.  Both panels
    // are read through buffer resources (base = this block's first row, scalar k offset): the k loop carries no
    // per-lane address arithmetic, and rows past M read as zero (num_records) instead of being clamped.
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    const long long rows_a = M - m0 < TBM ? M - m0 : TBM;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * K), 0, (int)(rows_a * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (long long)n0 * K), 0, TBN * K * 4, 0x00020000);
    int vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vo[i] = ((sr + 32 * i) * K + sc) * 4;
    f32x4 ra[2 * TM], rw[2 * TN];
    auto g_load = [&](int kt) {
        const int so = kt * BK * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 2 * TM) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo[i], so, 0));
            if (i < 2 * TN) rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vo[i], so, 0));
        }
    };
    float* const st_base = sm + sr * LDT + sc;
    auto s_store = [&](int st) {
        float* As = st_base + st * kStage;
        float* Ws = As + TBM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 2 * TM) *reinterpret_cast<f32x4*>(As + 32 * i * LDT) = ra[i];
            if (i < 2 * TN) *reinterpret_cast<f32x4*>(Ws + 32 * i * LDT) = rw[i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const float* const a_base = sm + (wm * 32 * TM + li) * LDT + 4 * hh;
    const float* const w_base = sm + TBM * LDT + (wn * 32 * TN + li) * LDT + 4 * hh;
    // one 32-deep k tile from LDS stage `st` (a literal at both call sites: every LDS offset is an immediate)
    auto compute = [&](int st) {
        const float* As = a_base + st * kStage;
        const float* Ws = w_base + st * kStage;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * LDT + 8 * t);
            f32x4 bf[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * LDT + 8 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
            }
        }
    };
    const int nk = K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) g_load(kt + 1);
        compute(0);
        __syncthreads();
        if (more) s_store(0);
        __syncthreads();
    }
    // Hadamard operand row of each chain row of this block (chain / n_z), one 64-bit division per row instead of
    // one per element: table in LDS (the tiles are dead after the loop's last barrier)
    int* mrow = reinterpret_cast<int*>(sm);
    if (EPI == EPI_SOFTPLUS_MUL || EPI == EPI_SCALE_MUL) {
        if (tid < TBM) {
            long long r = m0 + tid;
            if (r > M - 1) r = M - 1;
            mrow[tid] = (int)(r / mul_div);
        }
        __syncthreads();
    }
    // epilogue: C/D map of the 32x32 tile: column = lane & 31 (feature), row = (r&3) + 8(r>>2) + 4(lane>>5) (chain)
    constexpr bool kMul = EPI == EPI_SOFTPLUS_MUL || EPI == EPI_SCALE_MUL;
    // Hadamard operand: with in-kernel row tiling (n_z >= 64: every classification config) the rows of a tile come from at
    // most three input rows, so a lane needs three values of its column, not one global load per output element
    const int xr0 = kMul ? mrow[0] : 0;
    const bool few = kMul && mrow[TBM - 1] - xr0 <= 2;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * 32 * TN + b * 32 + li;
        const float bb = bias[col];
        float gg = 0.0f, ss = 0.0f;
        if (EPI != EPI_BIAS) { gg = g[col]; ss = sh[col]; }
        float mv[3] = {0.0f, 0.0f, 0.0f};
        if (few) {
            const int xl = mrow[TBM - 1];
#pragma unroll
            for (int j = 0; j < 3; ++j) mv[j] = mul[(long long)(xr0 + j <= xl ? xr0 + j : xl) * ldo + col];
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const long long row = m0 + lr;
                float v = acc[a][b][r] + bb;
                if (EPI == EPI_SOFTPLUS || EPI == EPI_SOFTPLUS_MUL) v = kLn2 * softplus_l2(fmaf(v, gg, ss));
                if (EPI == EPI_SCALE || EPI == EPI_SCALE_MUL) v = gg * v;      // gamma[t] * lin(x), builder.py:63-65
                if (row < M) {
                    if (kMul) {                                                 // Hadamard with x_feat (model.py:87)
                        if (few) {
                            const int dxr = mrow[lr] - xr0;
                            v *= dxr == 0 ? mv[0] : (dxr == 1 ? mv[1] : mv[2]);
                        } else {
                            v *= mul[(long long)mrow[lr] * ldo + col];
                        }
                    }
                    out[row * ldo + col] = v;
                }
            }
        }
    }
}

// One launch = n_main 128 x 128 tiles (rows [0, m_main), column block fastest) followed by 64 x 64 tiles over the rows
// [m_main, M): the small tiles are dispatched last, into the workgroup slots the big ones free one by one.
template <int EPI>
__global__ __launch_bounds__(256, kGemmWgPerCu) void wide_gemm_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ g,
                                                           const float* __restrict__ sh, const float* __restrict__ mul,
                                                           int mul_div, float* __restrict__ out, long long M, int K,
                                                           int ldo, int cols, long long n_main, long long m_main) {
    const long long id = blockIdx.x;
    if (id < n_main) {
        wide_gemm_tile<EPI, 128, 128>(A, W, bias, g, sh, mul, mul_div, out, m_main, K, ldo, 0, (int)(id % cols), id / cols);
    } else {
        const long long t = id - n_main;
        wide_gemm_tile<EPI, 64, 64>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, m_main, (int)(t % (2 * cols)), t / (2 * cols));
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

// Head + posterior update in ONE launch for up to 32 classes (every classification config but CIFAR-100; multi-output
// regression): eps = h . W4^T + b4 (classification/model.py:94) followed by the update of wide_update_kernel.  As a GEMM
// launch the head was a single, latency-bound round of N / 128 workgroups that each walked the whole K = F loop for a
// 128-column tile of which C are real (0.31 ms of a 12.6 ms CIFAR-10 timestep, the update 5 us more); the work is a
// read of h (N x F fp32, once) and C dot products per row, so it runs on the VALU at memory speed instead:
//   * a wave owns RW rows at a time; lane l walks features 4l .. 4l+3 (+256 ...), one coalesced 1 KB row segment per
//     load, and keeps RW x CP partial sums (fmaf in k order);
//   * the 64 partial sums of every (row, class) are added by a halving butterfly: at the step with lane distance d the
//     lane whose bit d is clear keeps the lower half of its values and hands the upper half to its partner (and vice
//     versa), so the values per lane halve while the lane distance halves -- V + V/2 + ... shuffles for all V = RW x CP
//     sums instead of 6 V, and value i ends on lane i / (V / 64).  The order of additions is the same for every value and
//     fp32 addition commutes, so a row's eps does not depend on where the row sits in its group, the launch or the chunk;
//   * that lane then updates its (row, class) elements: noise, coefficients and stores as in wide_update_kernel.
struct HeadParams {
    const float* h;          // (N, F) last hidden activations
    const float* W4;         // (>= CP, F), rows >= C zero
    const float* b4;
    const float* mu;         // (rows, C)
    const float* coef_row;
    const float* noise_d;    // this draw's (N, C) block or NULL
    float* S;                // (N, Kp) state matrix, columns [0, C) = y_t
    float* out;
    const int* state_slot;
    float* eps_out;          // eps-only mode: (N, C), nothing else is touched
    long long N, fixed_slot, slot_stride;
    unsigned long long seed, elem_offset;
    int F, C, Kp, n_z, draw, t, state;
};

template <int CP, int RW>
__global__ __launch_bounds__(256) void wide_head_update_kernel(const HeadParams p) {
    constexpr int V = CP * RW, PER = V / 64;
    static_assert(V % 64 == 0 && (V & (V - 1)) == 0, "RW x CP must be a power of two >= 64");
    // W4 is shared by every row: the four waves of a workgroup walk k in lockstep and read it through a double-buffered
    // LDS slab of 256 features x C classes (one global read per workgroup instead of one per wave: with per-wave reads
    // the kernel was bound by the 2.5 W4 loads it issued per row-segment load)
    __shared__ __attribute__((aligned(16))) float slab[2][CP][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long groups = (p.N + RW - 1) / RW, sgroups = (groups + 3) / 4;
    const int F = p.F, C = p.C;
    float c_inv = 0.f, c_om = 0.f, c_s = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, sig = 0.f;
    long long slot = -1;
    if (!p.eps_out) {
        c_inv = p.coef_row[CARD_COEF_INV_SQRT_AB]; c_om = p.coef_row[CARD_COEF_ONE_M_SQRT_AB]; c_s = p.coef_row[CARD_COEF_SOMAB];
        g0 = p.coef_row[CARD_COEF_G0]; g1 = p.coef_row[CARD_COEF_G1]; g2 = p.coef_row[CARD_COEF_G2]; sig = p.coef_row[CARD_COEF_SIGMA];
        slot = p.state_slot ? (long long)p.state_slot[p.state] : p.fixed_slot;
    }
    float* const out_slot = slot >= 0 ? p.out + slot * p.slot_stride : nullptr;
    const int k0 = 4 * lane;
    const int n_ks = (F + 255) / 256;
    constexpr int WR = (CP + 3) / 4;                        // W4 rows a wave stages per slab: classes wv, wv + 4, ...
    auto w_fetch = [&](int ks, f32x4 (&wr)[WR]) {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int c = wv + 4 * i;
            wr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < C && ks * 256 + k0 < F) wr[i] = *reinterpret_cast<const f32x4*>(p.W4 + (long long)c * F + ks * 256 + k0);
        }
    };
    auto w_park = [&](int buf, const f32x4 (&wr)[WR]) {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int c = wv + 4 * i;
            if (c < CP) *reinterpret_cast<f32x4*>(&slab[buf][c][k0]) = wr[i];
        }
    };
    for (long long sg = blockIdx.x; sg < sgroups; sg += gridDim.x) {     // uniform over the workgroup (barriers inside)
        const long long grp = sg * 4 + wv;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.0f;
        const float* hrow[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            long long r = grp * RW + j;
            r = r < p.N ? r : p.N - 1;
            hrow[j] = p.h + r * F + k0;
        }
        // the row segments and the W4 slab of k-step i + 1 are requested before the FMAs of k-step i
        f32x4 hv[RW], hn[RW], wr[WR];
#pragma unroll
        for (int j = 0; j < RW; ++j) hn[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k0 < F) {
#pragma unroll
            for (int j = 0; j < RW; ++j) hn[j] = *reinterpret_cast<const f32x4*>(hrow[j]);
        }
        w_fetch(0, wr);
        __syncthreads();                                   // the previous group's last slab has been read by every wave
        w_park(0, wr);
        __syncthreads();
        for (int ks = 0; ks < n_ks; ++ks) {
            const bool more = ks + 1 < n_ks;
#pragma unroll
            for (int j = 0; j < RW; ++j) hv[j] = hn[j];
            if (more) {
                w_fetch(ks + 1, wr);
                if ((ks + 1) * 256 + k0 < F) {
#pragma unroll
                    for (int j = 0; j < RW; ++j) hn[j] = *reinterpret_cast<const f32x4*>(hrow[j] + (ks + 1) * 256);
                }
            }
            if (ks * 256 + k0 < F) {
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    if (c < C) {                           // uniform: padded classes cost nothing
                        const f32x4 w = *reinterpret_cast<const f32x4*>(&slab[ks & 1][c][k0]);
#pragma unroll
                        for (int j = 0; j < RW; ++j) {
                            float a = acc[j * CP + c];
                            a = fmaf(hv[j][0], w[0], a); a = fmaf(hv[j][1], w[1], a); a = fmaf(hv[j][2], w[2], a); a = fmaf(hv[j][3], w[3], a);
                            acc[j * CP + c] = a;
                        }
                    }
                }
            }
            if (more) w_park((ks + 1) & 1, wr);            // that buffer was last read at k-step ks - 1, before the barrier below of that trip
            __syncthreads();
        }
        // halving butterfly: after the step with distance d a lane holds the values whose index bit (the one this step
        // decides) equals its lane bit d
#pragma unroll
        for (int d = 32, half = V / 2; d >= 1; d >>= 1, half >>= 1) {
            const bool up = (lane & d) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float keep = up ? acc[i + half] : acc[i];
                const float send = up ? acc[i] : acc[i + half];
                acc[i] = keep + __shfl_xor(send, d, 64);
            }
        }
        // this lane's PER values: index i = PER * lane + s -> row i / CP of the group, class i % CP
#pragma unroll
        for (int s = 0; s < PER; ++s) {
            const int i = PER * lane + s, j = i / CP, c = i % CP;
            const long long n = grp * RW + j;
            if (n >= p.N || c >= C) continue;
            const float e = acc[s] + p.b4[c];
            const long long el = n * C + c;
            if (p.eps_out) { p.eps_out[el] = e; continue; }
            const float m = p.mu[(n / p.n_z) * C + c], yy = p.S[n * p.Kp + c];
            const float y0r = c_inv * ((yy - c_om * m) - e * c_s);
            float v = y0r;
            if (p.t >= 1) {
                float z;
                if (p.noise_d) {
                    z = p.noise_d[el];
                } else {
                    float zz[4];
                    philox_normal4(p.seed, p.elem_offset + (unsigned long long)el, (uint32_t)p.draw >> 2, zz);
                    z = pick4(zz, p.draw & 3);
                }
                v = ((g0 * y0r + g1 * yy) + g2 * m) + sig * z;
            }
            p.S[n * p.Kp + c] = v;
            if (out_slot) out_slot[el] = v;
        }
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

// out[M x Nout] = epilogue(A[M x K] . W[Nout x K]^T).  Whole rounds of kGemmWgPerCu x num_cu resident 128 x 128 tiles, then, when the
// last round would fill at most half of the chip, its rows as 64 x 64 tiles (see wide_gemm_tile) -- all in one launch.
int gemm(const card_sampler* s, const CardRun& r, int epi, const float* A, const float* W, const float* bias, const float* g, const float* sh,
         const float* mul, int mul_div, float* out, long long M, int K, int Nout, int ldo, hipStream_t stream) {
    const int num_cu = s->num_cu;
    const long long cols = Nout / BN, rb = (M + BM - 1) / BM, slots = (long long)kGemmWgPerCu * num_cu;
    long long main_rb = rb;
    const bool tail_on = !s->tune.wide_notail;                       // CARD_AMD_WIDE=notail: 128 x 128 tiles only (tests, A/B)
    if (tail_on && K >= 8 * BK && cols <= slots) {
        const long long rb_round = slots / cols;                       // row blocks of one round of resident workgroups
        const long long tail_rb = rb % rb_round;
        if (tail_rb * cols * 2 <= slots) main_rb = rb - tail_rb;       // the last round is at most half full
    }
    const long long m_main = main_rb * BM < M ? main_rb * BM : M;
    const long long n_main = main_rb * cols, n_tail = (M - m_main + 63) / 64 * (2 * cols);
    const dim3 grid((unsigned)(n_main + n_tail));
    const int c = (int)cols;
    const bool big = K == Nout;                      // the F x F layers carry the flops
    if (epi == EPI_SOFTPLUS)
        CARD_LAUNCH(r, "wide_gemm_kernel<EPI_SOFTPLUS>", big, wide_gemm_kernel<EPI_SOFTPLUS><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, c, n_main, m_main));
    else if (epi == EPI_SOFTPLUS_MUL)
        CARD_LAUNCH(r, "wide_gemm_kernel<EPI_SOFTPLUS_MUL>", false, wide_gemm_kernel<EPI_SOFTPLUS_MUL><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, c, n_main, m_main));
    else if (epi == EPI_SCALE)
        CARD_LAUNCH(r, "wide_gemm_kernel<EPI_SCALE>", big, wide_gemm_kernel<EPI_SCALE><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, c, n_main, m_main));
    else if (epi == EPI_SCALE_MUL)
        CARD_LAUNCH(r, "wide_gemm_kernel<EPI_SCALE_MUL>", false, wide_gemm_kernel<EPI_SCALE_MUL><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, c, n_main, m_main));
    else
        CARD_LAUNCH(r, "wide_gemm_kernel<EPI_BIAS>", false, wide_gemm_kernel<EPI_BIAS><<<grid, 256, kGemmLds, stream>>>(A, W, bias, g, sh, mul, mul_div, out, M, K, ldo, c, n_main, m_main));
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
    {
        const void* fns[] = {(const void*)wide_gemm_kernel<EPI_SOFTPLUS>, (const void*)wide_gemm_kernel<EPI_BIAS>,
                             (const void*)wide_gemm_kernel<EPI_SOFTPLUS_MUL>, (const void*)wide_gemm_kernel<EPI_SCALE>,
                             (const void*)wide_gemm_kernel<EPI_SCALE_MUL>};
        for (const void* fn : fns) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    }
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

    // the hidden layers of one eps_theta forward from the state matrix S (y_t | y_0_hat | [x] | 0) at timestep t; *last =
    // the buffer that holds the activations lin4 reads
    auto hidden = [&](int t, const float** last) -> int {
        const float* g = s->gtab + (size_t)t * 3 * F;
        int rc;
        if (s->variant == CARD_VARIANT_LINEAR_HADAMARD) {
            // lin3( gamma2 * lin2( linx(z) * (gamma1 * lin1(S)) ) ), imagenet/card/builder.py:86-92
            if ((rc = gemm(s, r, EPI_SCALE_MUL, w.S, s->W1, b1, g, s->shift, r.x, r.n_z, w.hA, N, Kp, F, F, stream))) return rc;
            *last = w.hB;
            return gemm(s, r, EPI_SCALE, w.hA, s->W2, b2, g + F, s->shift + F, nullptr, 1, w.hB, N, F, F, F, stream);
        }
        if ((rc = gemm(s, r, guided ? EPI_SOFTPLUS : EPI_SOFTPLUS_MUL, w.S, s->W1, b1, g, s->shift, r.x, r.n_z, w.hA, N, Kp, F, F, stream))) return rc;
        if ((rc = gemm(s, r, EPI_SOFTPLUS, w.hA, s->W2, b2, g + F, s->shift + F, nullptr, 1, w.hB, N, F, F, F, stream))) return rc;
        *last = w.hA;
        return gemm(s, r, EPI_SOFTPLUS, w.hB, s->W3, b3, g + 2 * (size_t)F, s->shift + 2 * (size_t)F, nullptr, 1, w.hA, N, F, F, F, stream);
    };
    // lin4 as a GEMM launch into w.eps (N x Cpad): more than kHeadFusedMaxC classes, or CARD_AMD_WIDE=gemmhead (tests, A/B)
    auto head_gemm = [&](const float* last) -> int {
        return gemm(s, r, EPI_BIAS, last, s->W4, b4, nullptr, nullptr, nullptr, 1, w.eps, N, F, Cpad, Cpad, stream);
    };
    const bool fused_head = C <= kHeadFusedMaxC && !s->tune.wide_gemmhead;
    // lin4 + posterior update (or eps only) in one VALU launch
    auto head_fused = [&](const HeadParams& hp) -> int {
        // a wave owns RW = 64 / CP rows at a time; as many waves as give every wave the same number of row groups
        const int RW = C <= 8 ? 8 : (C <= 16 ? 4 : 2);
        const long long sgroups = ((N + RW - 1) / RW + 3) / 4, max_wgs = (long long)s->num_cu * 2;   // resident workgroups (registers)
        const long long rounds = (sgroups + max_wgs - 1) / max_wgs;
        const dim3 grid((unsigned)((sgroups + rounds - 1) / rounds));
        if (C <= 8) CARD_LAUNCH(r, "wide_head_update_kernel<8>", false, wide_head_update_kernel<8, 8><<<grid, 256, 0, stream>>>(hp));
        else if (C <= 16) CARD_LAUNCH(r, "wide_head_update_kernel<16>", false, wide_head_update_kernel<16, 4><<<grid, 256, 0, stream>>>(hp));
        else CARD_LAUNCH(r, "wide_head_update_kernel<32>", false, wide_head_update_kernel<32, 2><<<grid, 256, 0, stream>>>(hp));
        return CARD_OK;
    };
    HeadParams hp;
    memset(&hp, 0, sizeof(hp));
    hp.W4 = s->W4; hp.b4 = b4; hp.mu = r.ytmean; hp.S = w.S; hp.out = r.out; hp.N = N; hp.slot_stride = stride;
    hp.seed = r.seed; hp.elem_offset = r.elem_offset; hp.F = F; hp.C = C; hp.Kp = Kp; hp.n_z = r.n_z;
    // static columns of S (+ y_t when the caller supplies it)
    CARD_LAUNCH(r, "wide_pack_state_kernel", false,
                wide_pack_state_kernel<<<grid_for(N * Kp), 256, 0, stream>>>(r.y_init, r.y0hat, guided ? r.x : nullptr, C, s->cat_y_pred,
                                                                            (guided && s->cat_x) ? s->x_dim : 0, Kp, r.n_z, N, w.S));

    if (r.eps_only) {
        const float* last = nullptr;
        if (int rc = hidden(r.t_hi, &last)) return rc;
        if (fused_head) {
            hp.h = last; hp.eps_out = r.out;
            return head_fused(hp);
        }
        if (int rc = head_gemm(last)) return rc;
        CARD_LAUNCH(r, "wide_copy_eps_kernel", false, wide_copy_eps_kernel<<<grid_for(ne), 256, 0, stream>>>(w.eps, Cpad, C, ne, r.out));
        return CARD_OK;
    }
    const int T = r.T;
    const int base = r.y_init ? (T - r.t_hi) : 0;          // first draw consumed / first state produced
    // output slot of a state: fixed (CARD_OUT_ALL), looked up on the device (CARD_OUT_SELECT) or none
    const int* sel = r.out_mode == CARD_OUT_SELECT ? r.state_slot : nullptr;
    auto fixed = [&](int state) -> long long { return r.out_mode == CARD_OUT_ALL ? state - base : -1; };
    if (!r.y_init) {
        CARD_LAUNCH(r, "wide_init_kernel", false,
                    wide_init_kernel<<<grid_for(ne), 256, 0, stream>>>(r.ytmean, r.noise, r.seed, r.elem_offset, ne, C, Kp, r.n_z, w.S,
                                                                      r.out, sel, fixed(0), stride));
    }
    for (int t = r.t_hi; t >= r.t_lo; --t) {
        const float* last = nullptr;
        if (int rc = hidden(t, &last)) return rc;
        const int draw = T - t;
        const float* nz = (r.noise && t >= 1) ? r.noise + (long long)(draw - base) * stride : nullptr;
        if (fused_head) {
            hp.h = last; hp.coef_row = r.coef + (size_t)t * CARD_COEF_COLS; hp.noise_d = nz; hp.draw = draw; hp.t = t;
            hp.state_slot = sel; hp.state = T - t; hp.fixed_slot = fixed(T - t);
            if (int rc = head_fused(hp)) return rc;
            continue;
        }
        if (int rc = head_gemm(last)) return rc;
        CARD_LAUNCH(r, "wide_update_kernel", false,
                    wide_update_kernel<<<grid_for(ne), 256, 0, stream>>>(w.eps, Cpad, r.ytmean, r.coef + (size_t)t * CARD_COEF_COLS, nz,
                                                                        r.seed, r.elem_offset, draw, t, C, ne, Kp, r.n_z, w.S, r.out, sel, T - t,
                                                                        fixed(T - t), stride));
    }
    if (r.out_mode == CARD_OUT_LAST) {
        CARD_LAUNCH(r, "wide_copy_eps_kernel", false, wide_copy_eps_kernel<<<grid_for(ne), 256, 0, stream>>>(w.S, Kp, C, ne, r.out));   // y_t columns of S -> (N, C)
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
    if (r.plan) {    // layer 1 with K padded to the state matrix's 32-column pitch, the F x F layers, lin4 fused (C columns) or as a GEMM padded to 128
        const bool fused = C <= kHeadFusedMaxC && !s->tune.wide_gemmhead;
        const int hidden_layers = s->variant == CARD_VARIANT_LINEAR_HADAMARD ? 1 : 2;
        r.plan->executed = 2.0 * s->hidden * (state_ld(s) + hidden_layers * (double)s->hidden + (fused ? C : Cpad));
    }
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
