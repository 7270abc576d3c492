// card_guided.hip - the sampler kernels for CARD_VARIANT_GUIDED at hidden 128 (regression ConditionalGuidedModel):
// the persistent kernel described here and, for small batches, the cooperative form further down.
//
// One launch runs all T timesteps of p_sample_loop (reference: regression/diffusion_utils.py:107-119) for up
// to 1 048 576 independent (row, z) chains:
//   * a wave owns a tile of 32 chains for the whole loop; y_t, y_T_mean and the layer-1 state stay on chip
//   * activations are kept TRANSPOSED (feature x chain): a 32x32 fp32 MFMA accumulator of layer l is,
//     register for register, the B operand of layer l+1 (v_mfma_f32_32x32x2_f32 sums over the
//     accumulator's row index), so the MLP chains through registers with no cross-lane traffic
//   * W2, W3, biases and W4 live in LDS for the whole kernel in "fragment order" (one ds_read_b128 per
//     lane yields the A operands of four consecutive MFMAs, conflict-free); per-step gamma rows are staged
//     through a small wave-private LDS slab, prefetched one step ahead
//   * layer 1: the step-invariant part b1 + W1[:,1:].(y_0_hat, x) is computed once per tile and one K=2 MFMA per
//     output tile adds W1[:,0]*y_t each step (any d_in): kept as two columns in LDS when rows are tiled in-kernel
//     with n_z >= 32 (HOIST = 2), else in a per-wave 16 KB scratch slot re-read from L2 (HOIST = 1).  (Round 1's
//     un-hoisted form -- W1 in LDS, layer 1 on MFMA every step, d_in <= 32 -- lost its A/B by 1.5-3.5 % and is gone.)
//   * epilogue per element: xs = acc * (gamma*log2e); softplus/ln2 = xs > 20*log2e ? xs : log2(1 + exp2(xs))
//     (ln2 is folded into the next layer's weights), i.e. F.softplus(beta=1, threshold=20)
//   * measured on gfx950: v_mfma_f32_32x32x2_f32 barely overlaps with VALU work of either wave of a SIMD
//     (tools/ubench), so MFMA and epilogue are issued as large blocks (A fragments pinned one group ahead, the
//     epilogue staged over a tile and run at raised wave priority) and the kernel's ceiling is about
//     MFMA time / (MFMA time + VALU time) = 80 %
//   * noise: injected (validation) or Philox4x32-10 + Box-Muller keyed by (seed, element, draw)
#include "card_guided.h"

using namespace card;

namespace {

// gamma multiply + softplus of one 32-feature x 32-chain accumulator tile, in place
// PRIO: raise the wave's priority for the epilogue (see kPrioValu).  It pays where the waves sharing a SIMD run independently
// (the persistent kernel +2 %, two cooperative workgroups on a CU +3 %) and costs 1 % where they meet at the same barriers
// (the two halves of the paired time-sliced kernels), so those pass false.
template <bool PRIO = true>
__device__ __forceinline__ void epilogue_tile(f32x16& acc, const float* gl, int j) {
    // staged over the tile's 16 registers (all multiplies, all exp2, all +1, all log2, all selects): the transcendental
    // pipe is quarter rate with a long result latency, and a per-element chain mul -> exp -> add -> log -> med3 (what
    // the scheduler emits for the obvious loop) stalls on every link
    float xs[16], e[16];
    if (PRIO) __builtin_amdgcn_s_setprio(kPrioValu);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 g = lds4(gl + 32 * j + 8 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[4 * q + k] = acc[4 * q + k] * g[k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(xs[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = 1.0f + e[r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_logf(e[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_fmed3f(e[r], xs[r], kSpThresh);
    if (PRIO) __builtin_amdgcn_s_setprio(kPrioMfma);
}

__device__ __forceinline__ void load_bias(f32x16& acc, const float* bl, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = lds4(bl + 32 * j + 8 * q);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
}

// One ConditionalLinear + softplus with a 128-wide input held in registers (hin = previous layer's
// accumulators).  wl: packed weights, bl: bias, gl: this step's gamma*log2e (all LDS, pre-offset by lane).
__device__ __forceinline__ void layer_hidden(const float* wl, const float* bl, const float* gl, const f32x16 (&hin)[4],
                                             f32x16 (&hout)[4]) {
    // A fragments run one group (4 MFMAs = 256 cycles) ahead of their use, across output tiles too; the compiler
    // barriers pin that order (left alone, the scheduler sinks every ds_read next to its first MFMA and each group
    // then waits out the full LDS latency)
    f32x4 a = lds4(wl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x16 acc;
        load_bias(acc, bl, j);
#pragma unroll
        for (int i = 0; i < 16; ++i) {                  // i = 4jj + q
            const int g = j * 16 + i;
            f32x4 an = a;
            asm volatile("" ::: "memory");
            if (g + 1 < 64) an = lds4(wl + (g + 1) * 256);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], hin[i >> 2][4 * (i & 3) + e], acc, 0, 0, 0);
            a = an;
        }
        epilogue_tile(acc, gl, j);
        hout[j] = acc;
    }
}

// Layer 1 with the step-invariant part hoisted: the accumulator starts from this wave's scratch slot
// c1 = b1 + W1[:, 1:] . (y_0_hat, x) (computed once per tile) and one K=2 MFMA per output tile adds
// W1[:, 0] * y_t (a10: column 0 on lanes 0-31, zero on lanes 32-63; yb: y_t on lanes 0-31, zero above).
template <int STRIDE>   // floats between consecutive (j, q) fragments: 256 in a global slot, 8 in an LDS row
__device__ __forceinline__ void layer_first_hoisted(const float* slot, const float (&a10)[4], float yb,
                                                    const float* gl, f32x16 (&hout)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(slot + (j * 4 + q) * STRIDE);
            hout[j][4 * q + 0] = c[0]; hout[j][4 * q + 1] = c[1]; hout[j][4 * q + 2] = c[2]; hout[j][4 * q + 3] = c[3];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hout[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10[j], yb, hout[j], 0, 0, 0);
        epilogue_tile(hout[j], gl, j);
    }
}

// ------------------------------------------------------------------------------------------------
// The persistent sampler kernel (CARD_VARIANT_GUIDED, y_dim = 1, H = 128).
//   HOIST = 1    : any d_in; the step-invariant part of layer 1 is computed once per tile into the wave's
//                  scratch slot (16 KB, re-read from L2 every step) and only W1[:,0]*y_t is added per step.
// ------------------------------------------------------------------------------------------------
//   HOIST = 2    : as 1, for in-kernel row tiling with n_z >= 32: the 32 chains of a tile then come from at most
//                  two test rows, so the hoisted tile has only two distinct columns and lives in 1 KB of LDS per
//                  wave -- no scratch slots, no per-step global traffic at all.
template <int HOIST>
__global__ __launch_bounds__(kThreads, 2) void card_sample_guided_kernel(const RunParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(HOIST == 1 || HOIST == 2, "layer 1 is always hoisted");
    constexpr int IMG = kImgFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid * 4; i < IMG; i += kThreads * 4)
        *reinterpret_cast<f32x4*>(&lds[i]) = *reinterpret_cast<const f32x4*>(&p.wpk[i]);
    __syncthreads();

    const int m = lane & 31, hh = lane >> 5;
    float* gst = lds + IMG + wave * kGammaRow;           // wave-private gamma slab
    const float* w2l = lds + kOffW2 + lane * 4;
    const float* w3l = lds + kOffW3 + lane * 4;
    const float* bl = lds + kOffB + 4 * hh;
    const float* w4l = lds + kOffW4 + 4 * hh;
    const float* gl = gst + 4 * hh;
    const float b4 = lds[kOffB4];
    float a10[4];
    float* slot = nullptr;
    float* crow = lds + IMG + kWaves * kGammaRow + wave * (2 * kH) + 4 * hh;   // HOIST == 2: two hoisted columns
#pragma unroll
    for (int j = 0; j < 4; ++j) a10[j] = hh == 0 ? lds[kOffW1 + 32 * j + m] : 0.0f;
    if constexpr (HOIST == 1) slot = p.ws + ((long long)blockIdx.x * kWaves + wave) * kSlotFloats + lane * 4;

    const long long n_rows = p.n_end - p.n_begin;
    const long long n_tiles = (n_rows + kTileM - 1) / kTileM;
    const long long stride = (long long)gridDim.x * kWaves;
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;       // first draw this call consumes
    const int slot_base = p.y_init ? (T - p.t_hi) : 0;   // CARD_OUT_ALL: first state this call produces

    // tile -> (workgroup, wave): consecutive tiles go to different workgroups first, so a batch with fewer tiles than
    // wave slots spreads over all CUs (one wave per SIMD up to 4 x grid tiles) instead of filling a few CUs
    for (long long tile = (long long)wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += stride) {
        const long long row = p.n_begin + tile * kTileM + m;
        const bool valid = row < p.n_end;
        const long long rc = valid ? row : p.n_end - 1;
        const long long rr = p.n_z > 1 ? rc / p.n_z : rc;     // input row of this chain
        const float mu = p.ytmean[rr];
        const float y0h = p.y0hat[rr];
        const unsigned long long elem = p.elem_offset + (unsigned long long)rc;

        // feature k of cat(y_t, y_0_hat, x) (regression/model.py:38-47) for this lane's chain; k >= 1 is static
        auto static_feature = [&](int k) -> float {
            if (p.cat_y_pred && k == 1) return y0h;
            const int xi = k - 1 - (p.cat_y_pred ? 1 : 0);
            return (p.cat_x && xi < p.x_dim) ? p.x[rr * p.x_dim + xi] : 0.0f;
        };
        int sel = 0;                       // HOIST == 2: which of the tile's two test rows this chain belongs to
        {
            // c1 = b1 + W1[:, 1:] . static features, once per tile, on MFMA with A fragments from global (L2)
            f32x16 c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) load_bias(c[j], bl, j);
            for (int q = 0; q < p.nqs; ++q) {
                float bq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[e] = static_feature(1 + 8 * q + 2 * e + hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.w1s + ((long long)(j * p.nqs + q) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[e], c[j], 0, 0, 0);
                }
            }
            if constexpr (HOIST == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {c[j][4 * q + 0], c[j][4 * q + 1], c[j][4 * q + 2], c[j][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(slot + (j * 4 + q) * 256) = v;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                // n_z >= 32: chains of this tile come from test row rowA or rowA + 1; the first chain of each row
                // publishes its column of c (its 64 features per lane half) to the wave's LDS rows
                const long long rowA = (p.n_begin + tile * kTileM) / p.n_z;
                sel = rr != rowA ? 1 : 0;
                const unsigned long long second = __ballot(sel == 1) & 0xFFFFFFFFull;
                const int first_b = second ? __builtin_ctzll(second) : -1;
                __builtin_amdgcn_wave_barrier();
                if (m == 0 || m == first_b) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = {c[j][4 * q + 0], c[j][4 * q + 1], c[j][4 * q + 2], c[j][4 * q + 3]};
                            *reinterpret_cast<f32x4*>(crow + sel * kH + 32 * j + 8 * q) = v;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        float zc[4];
        uint32_t zblock = 0xFFFFFFFFu;
        auto draw = [&](int d) -> float {
            if (p.noise) return p.noise[(long long)(d - draw0) * p.N + rc];
            const uint32_t blk = (uint32_t)d >> 2;
            if (blk != zblock) {
                philox_normal4(p.seed, elem, blk, zc);
                zblock = blk;
            }
            return pick4(zc, d & 3);
        };
        auto store_state = [&](int s, float v) {
            long long slot;
            if (p.out_mode == CARD_OUT_ALL) slot = s - slot_base;
            else if (p.out_mode == CARD_OUT_SELECT) slot = p.state_slot[s];
            else return;
            if (slot >= 0 && valid && hh == 0) p.out[slot * p.N + row] = v;
        };

        float y;
        if (p.y_init) {
            y = p.y_init[rc];
        } else {
            y = draw(0) + mu;                                 // diffusion_utils.py:109-110
            store_state(0, y);
        }

        // gamma row of the first step
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 6; ++i) gst[i * 64 + lane] = p.gtab[(long long)p.t_hi * kGammaRow + i * 64 + lane];
        __builtin_amdgcn_wave_barrier();

        for (int t = p.t_hi; t >= p.t_lo; --t) {
            float gn[6];
            const bool more = t > p.t_lo;
            if (more) {
#pragma unroll
                for (int i = 0; i < 6; ++i) gn[i] = p.gtab[(long long)(t - 1) * kGammaRow + i * 64 + lane];
            }
            const float z = t >= 1 && !p.eps_only ? draw(T - t) : 0.0f;
            CoefRow cf = {};
            if (!p.eps_only) cf = load_coef(p.coef, t);        // (card_eps_theta passes no table)

            f32x16 hA[4], hB[4];
            if constexpr (HOIST == 1) {
                int zero = 0;
                asm volatile("" : "+v"(zero));    // keep the slot reads inside the step loop (no LICM into 64 VGPRs)
                layer_first_hoisted<256>(slot + zero, a10, hh == 0 ? y : 0.0f, gl, hA);
            } else {
                // (the OFFSET is laundered, not the pointer: a pointer that has been through an asm operand is a generic
                // one and the 16 reads below became flat_load_dwordx4 with a vmcnt(0) wait, not ds_read_b128)
                int off = sel * kH;
                asm volatile("" : "+v"(off));
                layer_first_hoisted<8>(crow + off, a10, hh == 0 ? y : 0.0f, gl, hA);
            }
            layer_hidden(w2l, bl + kH, gl + kH, hA, hB);
            layer_hidden(w3l, bl + 2 * kH, gl + 2 * kH, hB, hA);
            // lin4: eps = W4 . h3 + b4 (regression/model.py:51) as four per-tile partial sums (two half-wave chains each)
            // added in tile order -- the association the cooperative kernel below produces with one tile per wave, so
            // both forms give the same bits and results do not depend on which one a batch size selects
            float tp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float part = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = lds4(w4l + 32 * j + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part = fmaf(hA[j][4 * q + e], w[e], part);
                }
                tp[j] = part + __shfl_xor(part, 32, 64);
            }
            const float eps = (((tp[0] + tp[1]) + tp[2]) + tp[3]) + b4;

            if (p.eps_only) {
                if (valid && hh == 0) p.out[row] = eps;
                break;
            }

            y = posterior_update(cf, t, y, mu, eps, z);
            store_state(T - t, y);

            __builtin_amdgcn_wave_barrier();
            if (more) {
#pragma unroll
                for (int i = 0; i < 6; ++i) gst[i * 64 + lane] = gn[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.out_mode == CARD_OUT_LAST && !p.eps_only && valid && hh == 0) p.out[row] = y;
    }
}


// ------------------------------------------------------------------------------------------------
// Small batches: the cooperative form.  A timestep of a tile is a dependent chain of ~520 MFMAs for the wave that
// owns it (~19.5 us with a SIMD to itself), however few tiles there are (regression/card_regression.py:1043 calls
// p_sample_loop once per minibatch: 6 400 - 64 000 chains).  Here the FOUR waves of a workgroup (one per SIMD) share
// one 32-chain tile: wave w owns output features [32w, 32w+32) of every layer, keeps ITS rows of W2 and W3 in
// registers (2 x 64 VGPRs of A fragments: no weight traffic at all inside the loop) and the hoisted layer-1 tile in
// 16 more; after each layer the waves exchange activations through a 17 KB LDS tile (chain-major, 4 x ds_write_b128
// + barrier + 16 x ds_read_b128 per wave), and lin4 is a 4-way partial-sum exchange.  A timestep of a tile is then
// ~130 dependent MFMAs per wave instead of 516: ~5.6 us.
// ------------------------------------------------------------------------------------------------
// Philox draws of a tile, shared: every lane of the four waves used to run the generator for its own chain -- the same
// call on all 8 (wave, lane half) copies of a chain, ~1 500 cycles per wave every 4 timesteps.  Now copy (w, hh) draws
// block 8 grp + 2w + hh of chain m, i.e. one call per lane covers 32 timesteps of the tile, and the normals wait in LDS
// as [block & 7][draw & 3][chain] (conflict-free b32 reads).  Same generator, same (seed, element, block) -> same bits.
constexpr int kZFloats = 8 * 4 * kTileM;
constexpr int kCoopLdsFloats = 2 * kTileM * kExLd + kCoopWaves * kTileM + 2 * kGammaRow + kZFloats + kTabFloats;

__global__ __launch_bounds__(kCoopThreads, 2) void card_sample_coop_kernel(const RunParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const ex = lds;                                         // 2 exchange tiles [32][132]
    float* const red = ex + 2 * kTileM * kExLd;                    // [4][32] lin4 partial sums
    float* const gsl = red + kCoopWaves * kTileM;                  // 2 gamma rows
    float* const zb = gsl + 2 * kGammaRow;                         // the tile's normals for 32 draws
    float* const tab = zb + kZFloats;                              // small tables
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hh = lane >> 5;
    for (int i = tid; i < kTabFloats; i += kCoopThreads) tab[i] = p.wpk[kOffB + i];

    // this wave's A fragments of W2 and W3 (rows 32w .. 32w+31), same fragment order as the LDS image of the kernel above
    float A2[64], A3[64];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.wpk + kOffW2 + ((w * 16 + i) * 64 + lane) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.wpk + kOffW3 + ((w * 16 + i) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { A2[4 * i + e] = a[e]; A3[4 * i + e] = b[e]; }
    }
    __syncthreads();
    const float* const tb = tab + 4 * hh;                                        // bias rows: + layer * kH + 32w + 8q
    const float* const w4l = tab + (kOffW4 - kOffB) + 32 * w + 4 * hh;
    const float b4 = tab[kOffB4 - kOffB];
    const float a10 = hh == 0 ? tab[(kOffW1 - kOffB) + 32 * w + m] : 0.0f;
    float* const exw = ex + m * kExLd + 32 * w + 4 * hh;                         // where this lane publishes (+ 8q)
    const float* const exr = ex + m * kExLd + 4 * hh;                            // where it reads (+ 32jj + 8q)

    const long long n_rows = p.n_end - p.n_begin;
    const long long n_tiles = (n_rows + kTileM - 1) / kTileM;
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;
    int gcur = 0;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {       // same trip count for the 4 waves
        const long long row = p.n_begin + tile * kTileM + m;
        const bool valid = row < p.n_end;
        const long long rc = valid ? row : p.n_end - 1;
        const long long rr = p.n_z > 1 ? rc / p.n_z : rc;
        const float mu = p.ytmean[rr];
        const float y0h = p.y0hat[rr];
        const unsigned long long elem = p.elem_offset + (unsigned long long)rc;
        auto static_feature = [&](int k) -> float {
            if (p.cat_y_pred && k == 1) return y0h;
            const int xi = k - 1 - (p.cat_y_pred ? 1 : 0);
            return (p.cat_x && xi < p.x_dim) ? p.x[rr * p.x_dim + xi] : 0.0f;
        };
        // hoisted layer 1 for this wave's 32 features: c1 = b1 + W1[:, 1:] . (y_0_hat, x), kept in registers
        f32x16 c1;
        load_bias(c1, tb, w);
        for (int q = 0; q < p.nqs; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.w1s + ((long long)(w * p.nqs + q) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], static_feature(1 + 8 * q + 2 * e + hh), c1, 0, 0, 0);
        }
        // zfill(d): make sure zb holds the group of 32 draws that contains draw d (workgroup-uniform; the reads of a draw
        // come after the next barrier, and a refill is at least two barriers behind the last read of the old group)
        int zgrp = -1;
        auto zfill = [&](int d) {
            if (__builtin_expect(p.noise || (d >> 5) == zgrp, 1)) return;   // (the refill, once per 32 timesteps, out of line)
            zgrp = d >> 5;
            float zz[4];
            philox_normal4(p.seed, elem, (uint32_t)(8 * zgrp + 2 * w + hh), zz);
#pragma unroll
            for (int j = 0; j < 4; ++j) zb[((2 * w + hh) * 4 + j) * kTileM + m] = zz[j];
        };
        auto draw_lds = [&](int d) -> float { return zb[(((d >> 2) & 7) * 4 + (d & 3)) * kTileM + m]; };
        // (a select of VALUES: `p.noise ? p.noise[..] : zb[..]` as one load compiles to a select of the two ADDRESSES and a flat_load)
        auto draw = [&](int d) -> float {
            float zl = draw_lds(d);
            asm volatile("" : "+v"(zl));
            float zg = 0.0f;
            if (p.noise) zg = p.noise[(long long)(d - draw0) * p.N + rc];
            return p.noise ? zg : zl;
        };
        // (the common cases of a timestep fall through, see card_sample_coop16_kernel: the rarer ones sit out of line)
        const bool st_wave = w == 0 && p.out_mode != CARD_OUT_LAST;       // wave-uniform: this wave stores states
        auto store_state = [&](int s, float v, int selected) {
            if (__builtin_expect(st_wave, 0)) {
                const long long slot = p.out_mode == CARD_OUT_ALL ? s - slot_base : selected;
                if (slot >= 0 && valid && hh == 0) p.out[slot * p.N + row] = v;
            }
        };
        if (!p.y_init) zfill(0);
        // gamma row of the first step (the previous tile's last reads of either buffer are behind its barriers)
        for (int i = tid; i < kGammaRow; i += kCoopThreads) gsl[gcur * kGammaRow + i] = p.gtab[(long long)p.t_hi * kGammaRow + i];
        __syncthreads();
        float y;
        if (p.y_init) {
            y = p.y_init[rc];
        } else {
            y = draw(0) + mu;                                 // diffusion_utils.py:109-110
            store_state(0, y, p.out_mode == CARD_OUT_SELECT ? p.state_slot[0] : 0);
        }

        // publish this wave's 32 x 32 block of activations / gather all 128 features of the tile's chains
        auto publish = [&](const f32x16& a, int buf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                *reinterpret_cast<f32x4*>(exw + buf * kTileM * kExLd + 8 * q) = v;
            }
        };
        auto hidden = [&](const float (&A)[64], int buf, int layer, const float* gl) -> f32x16 {
            f32x16 acc;
            load_bias(acc, tb + layer * kH, w);
#pragma unroll
            for (int i = 0; i < 16; ++i) {                    // i = 4jj + q: features 32jj + 8q + 4hh .. +3
                const f32x4 h = lds4(exr + buf * kTileM * kExLd + 32 * (i >> 2) + 8 * (i & 3));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * i + e], h[e], acc, 0, 0, 0);
            }
            epilogue_tile(acc, gl + layer * kH, w);
            return acc;
        };

        // (What card_sample_coop16_kernel does at this point -- every load waited for before the loop, the timestep's
        // global loads unconditional at its head -- measured 0.6 % SLOWER here: with two workgroups per CU the waits are
        // covered, and every lane requesting the gamma row costs more than it saves.)
        const int grow = (kGammaRow / 16) * w + (lane < kGammaRow / 16 ? lane : 0);      // this lane's 16 bytes of a gamma row
        for (int t = p.t_hi; t >= p.t_lo; --t) {
            const bool more = t > p.t_lo;
            f32x4 gn = {0.f, 0.f, 0.f, 0.f};
            // (a quarter of the row per wave, lanes 0-23: no wave branches around an empty load or store)
            if (more && lane < kGammaRow / 16) gn = *reinterpret_cast<const f32x4*>(p.gtab + (long long)(t - 1) * kGammaRow + grow * 4);
            const bool noisy = t >= 1 && !p.eps_only;
            CoefRow cf = {};
            if (!p.eps_only) cf = load_coef(p.coef, t);
            if (noisy) zfill(T - t);
            const float* const gl = gsl + gcur * kGammaRow + 4 * hh;

            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, hh == 0 ? y : 0.0f, c1, 0, 0, 0);
            epilogue_tile(acc, gl, w);
            publish(acc, 0);
            __syncthreads();
            const float z = noisy ? draw(T - t) : 0.0f;
            acc = hidden(A2, 0, 1, gl);
            publish(acc, 1);
            __syncthreads();
            acc = hidden(A3, 1, 2, gl);
            // next step's gamma row: written between two barriers of this step, read after the next one
            if (more && lane < kGammaRow / 16) *reinterpret_cast<f32x4*>(gsl + (gcur ^ 1) * kGammaRow + grow * 4) = gn;
            // lin4: this wave's partial over its 32 features, then the four waves' partials in a fixed order
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = lds4(w4l + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(acc[4 * q + e], wv[e], part);
            }
            part += __shfl_xor(part, 32, 64);
            if (hh == 0) red[w * kTileM + m] = part;
            __syncthreads();
            const float eps = (((red[m] + red[kTileM + m]) + red[2 * kTileM + m]) + red[3 * kTileM + m]) + b4;
            gcur ^= more ? 1 : 0;

            if (__builtin_expect(p.eps_only, 0)) {
                if (valid && hh == 0 && w == 0) p.out[row] = eps;
                break;
            }
            y = posterior_update(cf, t, y, mu, eps, z);
            store_state(T - t, y, p.out_mode == CARD_OUT_SELECT ? p.state_slot[T - t] : 0);
        }
        if (p.out_mode == CARD_OUT_LAST && !p.eps_only && valid && hh == 0 && w == 0) p.out[row] = y;
        __syncthreads();                                      // the next tile rewrites the gamma buffer and red[]
    }
}

// ------------------------------------------------------------------------------------------------
// Between one and two tiles per CU: the paired cooperative form, time-sliced.
// With 256 < n_tiles < 512 (12 800 chains = the reference's Year minibatch, regression/configs/uci_year.yml:62,67) the
// kernel above puts two workgroups on some CUs and one on the others; a CU with two tiles needs ~9.1 us per timestep,
// one with a single tile ~5.3 us, so the singles idle for 40 % of the launch.  Here a workgroup is TWO cooperative
// groups ("halves", 4 waves each, one workgroup per CU, so which tiles share a CU is the kernel's choice and not the
// dispatcher's), and the loop is cut into sl_K launches ("slices").  In slice k the tiles [k * n_single, (k+1) * n_single)
// run alone on a CU (half 1 of their workgroup only keeps the barriers company) and advance L timesteps, every other tile
// shares a CU and advances S < L timesteps, with L * 5.3 us = S * 9.1 us: every CU is busy for the whole slice.  Each
// tile is "single" in exactly one slice, so after sl_K slices every chain has done (sl_K - 1) * S + L = all its
// timesteps.  Between slices a chain's state is one float in ybuf (stream order is the only synchronisation); the
// arithmetic per chain and timestep is the kernel's above, so results are bit-identical.  sl_K = 1 (S = L) is the
// un-sliced paired form.
// ------------------------------------------------------------------------------------------------
constexpr int kHalfLdsFloats = 2 * kTileM * kExLd + kCoopWaves * kTileM + 2 * kGammaRow + kZFloats;
constexpr int kCoop2LdsFloats = 2 * kHalfLdsFloats + kTabFloats;

__global__ __launch_bounds__(kCoop2Threads, 2) void card_sample_coop2_kernel(const RunParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w = wave & 3;
    const int htid = tid & (kCoopThreads - 1);                     // thread index inside the half
    float* const ex = lds + half * kHalfLdsFloats;                 // 2 exchange tiles [32][132]
    float* const red = ex + 2 * kTileM * kExLd;                    // [4][32] lin4 partial sums
    float* const gsl = red + kCoopWaves * kTileM;                  // 2 gamma rows
    float* const zb = gsl + 2 * kGammaRow;                         // the tile's normals for 32 draws (see kZFloats)
    float* const tab = lds + 2 * kHalfLdsFloats;                   // small tables, shared by the halves
    const int m = lane & 31, hh = lane >> 5;
    for (int i = tid; i < kTabFloats; i += kCoop2Threads) tab[i] = p.wpk[kOffB + i];

    // ---- which tile this half advances in this slice, from where, and how far
    const long long n_tiles = (p.n_end - p.n_begin + kTileM - 1) / kTileM;
    const int k = p.sl_k, nsg = p.sl_nsingle;
    const long long first_single = (long long)k * nsg;
    long long n_s = n_tiles - first_single;
    n_s = n_s < 0 ? 0 : (n_s > nsg ? nsg : n_s);                   // tiles that run alone in this slice
    long long tile = -1;
    int quota = 0;
    if ((long long)blockIdx.x < n_s) {
        if (half == 0) { tile = first_single + blockIdx.x; quota = p.sl_L; }
    } else {
        const long long j = 2 * ((long long)blockIdx.x - n_s) + half;
        if (j < n_tiles - n_s) { tile = j < first_single ? j : j + n_s; quota = p.sl_S; }
    }
    // every wave of the workgroup walks the same number of barrier rounds: those of half 0 (never fewer than half 1's)
    const int n_iter = (long long)blockIdx.x < n_s ? p.sl_L : p.sl_S;
    const bool active = tile >= 0;
    const int before = k * p.sl_S + ((active && tile / nsg < k) ? p.sl_L - p.sl_S : 0);
    const int th = p.t_hi - before;                                // first timestep of this slice for this tile
    const bool first_seg = before == 0, last_seg = th - quota + 1 <= p.t_lo;

    float A2[64], A3[64];
    if (active) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.wpk + kOffW2 + ((w * 16 + i) * 64 + lane) * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.wpk + kOffW3 + ((w * 16 + i) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { A2[4 * i + e] = a[e]; A3[4 * i + e] = b[e]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) { A2[i] = 0.0f; A3[i] = 0.0f; }
    }
    __syncthreads();
    const float* const tb = tab + 4 * hh;
    const float* const w4l = tab + (kOffW4 - kOffB) + 32 * w + 4 * hh;
    const float b4 = tab[kOffB4 - kOffB];
    const float a10 = hh == 0 ? tab[(kOffW1 - kOffB) + 32 * w + m] : 0.0f;
    float* const exw = ex + m * kExLd + 32 * w + 4 * hh;
    const float* const exr = ex + m * kExLd + 4 * hh;
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;

    const long long row = p.n_begin + (active ? tile : 0) * kTileM + m;
    const bool valid = active && row < p.n_end;
    const long long rc = row < p.n_end ? row : p.n_end - 1;
    const long long rr = p.n_z > 1 ? rc / p.n_z : rc;
    const float mu = p.ytmean[rr];
    const float y0h = p.y0hat[rr];
    const unsigned long long elem = p.elem_offset + (unsigned long long)rc;
    auto static_feature = [&](int kk) -> float {
        if (p.cat_y_pred && kk == 1) return y0h;
        const int xi = kk - 1 - (p.cat_y_pred ? 1 : 0);
        return (p.cat_x && xi < p.x_dim) ? p.x[rr * p.x_dim + xi] : 0.0f;
    };
    f32x16 c1;
    load_bias(c1, tb, w);
    if (active) {
        for (int q = 0; q < p.nqs; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.w1s + ((long long)(w * p.nqs + q) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], static_feature(1 + 8 * q + 2 * e + hh), c1, 0, 0, 0);
        }
    }
    int zgrp = -1;
    auto zfill = [&](int d) {                    // uniform over the half; see card_sample_coop_kernel
        if (__builtin_expect(p.noise || (d >> 5) == zgrp, 1)) return;       // (the refill, once per 32 timesteps, out of line)
        zgrp = d >> 5;
        float zz[4];
        philox_normal4(p.seed, elem, (uint32_t)(8 * zgrp + 2 * w + hh), zz);
#pragma unroll
        for (int j = 0; j < 4; ++j) zb[((2 * w + hh) * 4 + j) * kTileM + m] = zz[j];
    };
    auto draw_lds = [&](int d) -> float { return zb[(((d >> 2) & 7) * 4 + (d & 3)) * kTileM + m]; };
    // (a select of VALUES: `p.noise ? p.noise[..] : zb[..]` as one load compiles to a select of the two ADDRESSES and a flat_load)
    auto draw = [&](int d) -> float {
        float zl = draw_lds(d);
        asm volatile("" : "+v"(zl));
        float zg = 0.0f;
        if (p.noise) zg = p.noise[(long long)(d - draw0) * p.N + rc];
        return p.noise ? zg : zl;
    };
    // (the hot path of a timestep falls through, see card_sample_coop16_kernel: the rarer cases sit out of line)
    const bool st_wave = w == 0 && p.out_mode != CARD_OUT_LAST && !p.eps_only;      // wave-uniform: this wave stores states
    auto store_state = [&](int s, float v, int selected) {
        if (__builtin_expect(st_wave, 0)) {
            const long long slot = p.out_mode == CARD_OUT_ALL ? s - slot_base : selected;
            if (slot >= 0 && valid && hh == 0) p.out[slot * p.N + row] = v;
        }
    };
    float y = 0.0f;
    if (active) {
        if (first_seg && !p.y_init) zfill(0);
        for (int i = htid; i < kGammaRow; i += kCoopThreads) gsl[i] = p.gtab[(long long)th * kGammaRow + i];
    }
    __syncthreads();
    if (active) {
        if (!first_seg) {
            y = p.ybuf[rc - p.n_begin];
        } else if (p.y_init) {
            y = p.y_init[rc];
        } else {
            y = draw(0) + mu;                                 // diffusion_utils.py:109-110
            store_state(0, y, p.out_mode == CARD_OUT_SELECT ? p.state_slot[0] : 0);
        }
    }
    int gcur = 0;

    auto publish = [&](const f32x16& a, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            *reinterpret_cast<f32x4*>(exw + buf * kTileM * kExLd + 8 * q) = v;
        }
    };
    auto hidden = [&](const float (&A)[64], int buf, int layer, const float* gl) -> f32x16 {
        f32x16 acc;
        load_bias(acc, tb + layer * kH, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 h = lds4(exr + buf * kTileM * kExLd + 32 * (i >> 2) + 8 * (i & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * i + e], h[e], acc, 0, 0, 0);
        }
        epilogue_tile<false>(acc, gl + layer * kH, w);
        return acc;
    };

    // (see card_sample_coop16_kernel: every load so far is waited for here, and what a timestep takes from global memory
    // is requested unconditionally at its head and first touched where it is used)
    float mu_r = mu;
    asm volatile("" : "+v"(y), "+v"(mu_r));
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        asm volatile("" : "+v"(A2[i]), "+v"(A2[i + 1]), "+v"(A2[i + 2]), "+v"(A2[i + 3]), "+v"(A2[i + 4]), "+v"(A2[i + 5]), "+v"(A2[i + 6]), "+v"(A2[i + 7]));
        asm volatile("" : "+v"(A3[i]), "+v"(A3[i + 1]), "+v"(A3[i + 2]), "+v"(A3[i + 3]), "+v"(A3[i + 4]), "+v"(A3[i + 5]), "+v"(A3[i + 6]), "+v"(A3[i + 7]));
    }
    float eps_out = 0.0f;
    const int grow = lane < kGammaRow / 16 ? (kGammaRow / 16) * w + lane : 0;     // this lane's 16 bytes of a gamma row
    const float* const zin_base = p.noise ? p.noise + rc : p.ytmean + rr;         // injected noise, or a stand-in address
    const long long zin_stride = p.noise ? p.N : 0;
    for (int it = 0; it < n_iter; ++it) {
        const bool on = active && it < quota;                 // uniform per wave
        const int t = on ? th - it : p.t_lo;                  // (an idle half requests the rows of a valid timestep)
        const bool more = on && it + 1 < quota;
        float z = 0.0f;
        f32x16 acc;
        const float* const gl = gsl + gcur * kGammaRow + 4 * hh;
        const bool noisy = t >= 1 && !p.eps_only;
        const float* const crow = (p.eps_only ? p.gtab : p.coef) + (long long)t * CARD_COEF_COLS;
        f32x4 ca = *reinterpret_cast<const f32x4*>(crow), cb = *reinterpret_cast<const f32x4*>(crow + 4);
        const f32x4 gn = *reinterpret_cast<const f32x4*>(p.gtab + (long long)(more ? t - 1 : t) * kGammaRow + grow * 4);
        float zin = zin_base[(long long)(noisy ? T - t - draw0 : 0) * zin_stride];
        int slot_sel = *(p.out_mode == CARD_OUT_SELECT ? p.state_slot + (T - t) : reinterpret_cast<const int*>(p.wpk));
        if (on) {
            if (noisy) zfill(T - t);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, hh == 0 ? y : 0.0f, c1, 0, 0, 0);
            epilogue_tile<false>(acc, gl, w);
            publish(acc, 0);
        }
        __syncthreads();
        if (on) {
            z = noisy ? draw_lds(T - t) : 0.0f;               // (RNG mode's normal; injected noise is `zin`)
            acc = hidden(A2, 0, 1, gl);
            publish(acc, 1);
        }
        __syncthreads();
        if (on) {
            acc = hidden(A3, 1, 2, gl);
            if (more && lane < kGammaRow / 16) *reinterpret_cast<f32x4*>(gsl + (gcur ^ 1) * kGammaRow + grow * 4) = gn;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = lds4(w4l + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(acc[4 * q + e], wv[e], part);
            }
            part += __shfl_xor(part, 32, 64);
            if (hh == 0) red[w * kTileM + m] = part;
        }
        __syncthreads();
        if (on) {
            const float eps = (((red[m] + red[kTileM + m]) + red[2 * kTileM + m]) + red[3 * kTileM + m]) + b4;
            gcur ^= more ? 1 : 0;
            // (no branch on eps_only: such a call is one timestep, its update runs on the stand-in row and is not stored)
            eps_out = eps;
            asm volatile("" : "+v"(ca), "+v"(cb), "+v"(zin), "+v"(slot_sel));      // first touched here (see the head of the loop)
            const CoefRow cf = {ca[CARD_COEF_INV_SQRT_AB], ca[CARD_COEF_ONE_M_SQRT_AB], ca[CARD_COEF_SOMAB], ca[CARD_COEF_G0],
                                cb[CARD_COEF_G1 - 4], cb[CARD_COEF_G2 - 4], cb[CARD_COEF_SIGMA - 4]};
            y = posterior_update(cf, t, y, mu_r, eps, p.noise ? zin : z);
            store_state(T - t, y, slot_sel);
        }
    }
    if (valid && hh == 0 && w == 0) {
        if (p.eps_only) p.out[row] = eps_out;
        else if (!last_seg) p.ybuf[row - p.n_begin] = y;
        else if (p.out_mode == CARD_OUT_LAST) p.out[row] = y;
    }
}

// ------------------------------------------------------------------------------------------------
// create-time packing kernels
// ------------------------------------------------------------------------------------------------
// A-operand fragment order of a 128x128 layer: element e of lane l in block (j, jj, q) is
// W[32j + (l&31)][32jj + 8q + 4(l>>5) + e] -- the k index a lane needs when the B operand is register
// 4q+e of the previous layer's accumulator tile jj (C/D map: row = (reg&3) + 8(reg>>2) + 4(lane>>5)).
__global__ void pack_hidden_kernel(const float* __restrict__ W, float scale, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kH * kH) return;
    const int e = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) & 3, jj = (idx >> 10) & 3, j = idx >> 12;
    const int n = 32 * j + (l & 31), k = 32 * jj + 8 * q + 4 * (l >> 5) + e;
    out[idx] = W[n * kH + k] * scale;
}

// static columns 1..d_in-1 of W1; k-step 4q+e of group q takes feature 1 + 8q + 2e + (l>>5)
__global__ void pack_static_kernel(const float* __restrict__ W1, int d_in, int nqs, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kH * 8 * nqs) return;
    const int e = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8, q = rest % nqs, j = rest / nqs;
    const int n = 32 * j + (l & 31), k = 1 + 8 * q + 2 * e + (l >> 5);
    out[idx] = k < d_in ? W1[n * d_in + k] : 0.0f;
}

// 16x16x4 fragment order (card_sample_coop16_kernel): element e of lane l in block (w, i = 8 mb + a) is
// W[32w + 16mb + (l&15)][feature(s = 4a + e, g = l>>4)], feature(s, g) = 8(s>>1) + 2(s&1) + (g>>1) + 4(g&1)
__global__ void pack_hidden16_kernel(const float* __restrict__ W, float scale, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kH * kH) return;
    const int e = idx & 3, l = (idx >> 2) & 63, i = (idx >> 8) & 15, w = idx >> 12;
    const int mb = i >> 3, a = i & 7, s = 4 * a + e, g = l >> 4;
    const int nrow = 32 * w + 16 * mb + (l & 15), k = 8 * (s >> 1) + 2 * (s & 1) + (g >> 1) + 4 * (g & 1);
    out[idx] = W[nrow * kH + k] * scale;
}

// static columns of W1 for the 16-chain kernel: [(w, mb), q, e][lane] = W1[32w + 16mb + (l&15)][1 + 8q + 4e + (l>>4)]
__global__ void pack_static16_kernel(const float* __restrict__ W1, int d_in, int nqs, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 8 * nqs * 2 * 64) return;
    const int l = idx & 63, e = (idx >> 6) & 1, rest = idx >> 7, q = rest % nqs, wm = rest / nqs;
    const int nrow = 16 * wm + (l & 15), k = 1 + 8 * q + 4 * e + (l >> 4);
    out[idx] = k < d_in ? W1[nrow * d_in + k] : 0.0f;
}

__global__ void pack_col0_kernel(const float* __restrict__ W1, int d_in, float* __restrict__ out) {
    out[threadIdx.x] = W1[threadIdx.x * d_in];  // 128 threads
}

__global__ void pack_misc_kernel(const float* b1, const float* b2, const float* b3, const float* W4,
                                 const float* b4, float* __restrict__ img) {
    const int i = threadIdx.x;  // 128 threads
    img[kOffB + i] = b1[i];
    img[kOffB + kH + i] = b2[i];
    img[kOffB + 2 * kH + i] = b3[i];
    img[kOffW4 + i] = W4[i] * kLn2;
    if (i < 4) img[kOffB4 + i] = i == 0 ? b4[0] : 0.0f;
}

__global__ void pack_gamma_kernel(const float* g1, const float* g2, const float* g3, int n_embed,
                                  float* __restrict__ gtab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_embed * kGammaRow) return;
    const int t = idx / kGammaRow, r = idx % kGammaRow, l = r / kH, n = r % kH;
    const float* g = l == 0 ? g1 : (l == 1 ? g2 : g3);
    gtab[idx] = g[t * kH + n] * kLog2e;
}


}  // namespace

int guided_create(card_sampler* s, const card_model_desc* d, hipStream_t stream) {
    if (d->hidden != kH) return card_fail(CARD_E_UNSUPPORTED, "card_sampler_create: GUIDED hidden=%d, only 128 is built", d->hidden);
    if (d->y_dim != 1) return card_fail(CARD_E_UNSUPPORTED, "card_sampler_create: GUIDED y_dim=%d, only 1 is built", d->y_dim);
    const int d_in = s->d_in;
    const int nqs = (d_in - 1 + 7) / 8;
    s->nqs = nqs;
    s->lds_bytes = (size_t)(kImgFloats + kWaves * kGammaRow + kWaves * 2 * kH) * sizeof(float);
    HIP_TRY(hipMalloc(&s->wpk, (size_t)kImgFloats * sizeof(float)));
    HIP_TRY(hipMalloc(&s->gtab, (size_t)d->n_embed * kGammaRow * sizeof(float)));
    if (nqs > 0) HIP_TRY(hipMalloc(&s->w1s, (size_t)kH * 8 * nqs * sizeof(float)));
    HIP_TRY(hipMalloc(&s->ws, (size_t)s->num_cu * kWaves * kSlotFloats * sizeof(float)));
    HIP_TRY(hipFuncSetAttribute((const void*)card_sample_guided_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    HIP_TRY(hipFuncSetAttribute((const void*)card_sample_guided_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    // cooperative small-batch forms (CARD_AMD_COOP=off keeps every batch size on the persistent kernel: tests, A/B)
    s->coop_blocks = 0;
    if (!s->tune.coop_off) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)card_sample_coop_kernel, kCoopThreads,
                                                             (size_t)kCoopLdsFloats * sizeof(float)));
        s->coop_blocks = nb < 1 ? 1 : (nb > 2 ? 2 : nb);
        HIP_TRY(hipFuncSetAttribute((const void*)card_sample_coop2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kCoop2LdsFloats * sizeof(float))));
        if (int rc = coop16_prepare()) return rc;
        HIP_TRY(hipMalloc(&s->wpk16, (size_t)kImg16Floats * sizeof(float)));
        if (nqs > 0) HIP_TRY(hipMalloc(&s->w1s16, (size_t)8 * nqs * 2 * 64 * sizeof(float)));
        pack_hidden16_kernel<<<kH * kH / 256, 256, 0, stream>>>(d->W[1], kLn2, s->wpk16);
        pack_hidden16_kernel<<<kH * kH / 256, 256, 0, stream>>>(d->W[2], kLn2, s->wpk16 + kH * kH);
        if (nqs > 0) pack_static16_kernel<<<(8 * nqs * 2 * 64 + 255) / 256, 256, 0, stream>>>(d->W[0], d_in, nqs, s->w1s16);
    }
    pack_hidden_kernel<<<kH * kH / 256, 256, 0, stream>>>(d->W[1], kLn2, s->wpk + kOffW2);
    pack_hidden_kernel<<<kH * kH / 256, 256, 0, stream>>>(d->W[2], kLn2, s->wpk + kOffW3);
    if (int rc = split_create(s, d, stream)) return rc;      // CARD_ARITH_SPLIT_BF16 (reads this image's small tables)
    pack_col0_kernel<<<1, kH, 0, stream>>>(d->W[0], d_in, s->wpk + kOffW1);
    if (nqs > 0) pack_static_kernel<<<(kH * 8 * nqs + 255) / 256, 256, 0, stream>>>(d->W[0], d_in, nqs, s->w1s);
    pack_misc_kernel<<<1, kH, 0, stream>>>(d->b[0], d->b[1], d->b[2], d->W4, d->b4, s->wpk);
    const int ng = d->n_embed * kGammaRow;
    pack_gamma_kernel<<<(ng + 255) / 256, 256, 0, stream>>>(d->gamma[0], d->gamma[1], d->gamma[2], d->n_embed, s->gtab);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

void guided_destroy(card_sampler* s) {
    if (s->wpk) (void)hipFree(s->wpk);
    if (s->gtab) (void)hipFree(s->gtab);
    if (s->w1s) (void)hipFree(s->w1s);
    if (s->ws) (void)hipFree(s->ws);
    if (s->wpk16) (void)hipFree(s->wpk16);
    if (s->w1s16) (void)hipFree(s->w1s16);
    split_destroy(s);
}

// Cost ratios behind the small-batch dispatch.  Each is a ratio of CYCLE counts of kernels whose timestep touches no
// memory beyond LDS / registers (weights in registers or LDS, gamma rows prefetched a step ahead), so it does not move
// with the shader clock or with HBM speed; it would move with the MFMA issue rate of another chip generation.
// Measured on MI355X, tools/bench_small.py (profiles/r03_small_batch_ab.jsonl):
constexpr double kRho16 = 1.74;    // 16-chain tiles: timestep of two tiles sharing a CU (5.27 us) / of one tile alone with every CU busy (3.0 us); measured best of 1.68 / 1.74 / 1.80 / 1.86
constexpr double kRho32 = 1.82;    // 32-chain tiles: measured best of 1.69 / 1.717 / 1.745 / 1.78 / 1.82 / 1.87 after the paired kernel lost its priority changes (10 000 chains 6.85 / 6.75 / 6.67 / 6.62 / 6.57 / 6.58 ms)
// What a timestep of each form costs (us on MI355X, profiles/r03_small_batch_latency.jsonl; used as ratios, see above):
constexpr double kCostPers = 33.6;       // persistent kernel, a round of num_cu * 8 tiles (two waves per SIMD)
constexpr double kCostPersHalf = 18.0;   // ... up to num_cu * 4 tiles (every tile has a SIMD to itself)
constexpr double kCostCoopRound = 8.9;   // cooperative kernel, a round of two 32-chain tiles per CU
constexpr double kCostLone16 = 2.95;     // 16-chain tiles, at most one per CU; time-sliced forms in between: linear up to a round
constexpr double kCostLaunch = 5.0;      // a launch
constexpr double kPlanMinGain = 0.03;    // the last partial round of a call leaves the persistent kernel only if that saves this share of the call

namespace {

// [b, e), less than one round of the cooperative kernel (two 32-chain tiles per CU), on the small-batch forms
int guided_small(const card_sampler* s, const CardRun& r, RunParams& p, long long b, long long e, unsigned long long scratch_bytes, hipStream_t stream) {
    const int n_steps = p.t_hi - p.t_lo + 1;
    p.n_begin = b;
    p.n_end = e;
    const long long n = e - b, tiles_total = (n + kTileM - 1) / kTileM;
    const bool ybuf_fits = (unsigned long long)n * sizeof(float) <= scratch_bytes;   // chain states between time slices
    // Up to two 16-chain tiles per CU (8 192 chains on 256 CUs; BASELINE configs[0] is 5 100): the 16-chain-tile kernel,
    // one tile per CU up to num_cu tiles, the time-sliced single / paired scheme between num_cu and 2 x num_cu.
    // (a CU sharing two 16-chain tiles takes 5.3 us per timestep, one 32-chain tile alone 5.3 us: above 15/8 tiles
    // per CU the 32-chain form below is as fast)
    const long long tiles16 = (n + kT16 - 1) / kT16;
    if (s->coop_blocks == 2 && !s->tune.coop_no16 && s->wpk16 && tiles16 <= 2LL * s->num_cu - s->num_cu / 8) {
        long long n_single = tiles16 <= s->num_cu ? tiles16 : 2LL * s->num_cu - tiles16;
        int K = 1, S = n_steps, L = n_steps;
        if (tiles16 > s->num_cu && n_single > 0 && !p.eps_only && n_steps >= 64 && ybuf_fits) {
            const int K2 = (int)((tiles16 + n_single - 1) / n_single);
            const int S2 = (int)((double)n_steps / ((double)(K2 - 1) + kRho16) + 0.5);
            const int L2 = n_steps - (K2 - 1) * S2;
            if (K2 <= 8 && S2 >= 8 && L2 >= S2) { K = K2; S = S2; L = L2; }
        }
        if (K == 1 && tiles16 > s->num_cu) n_single = 2LL * s->num_cu - tiles16 > 0 ? 2LL * s->num_cu - tiles16 : 0;
        p.ybuf = p.ws;
        p.sl_K = K; p.sl_nsingle = (int)n_single; p.sl_S = S; p.sl_L = L;
        for (int k = 0; k < K; ++k) {
            long long n_s = n_single > 0 ? tiles16 - (long long)k * n_single : 0;
            n_s = n_s > n_single ? n_single : (n_s < 0 ? 0 : n_s);
            const long long grid = n_s + (tiles16 - n_s + 1) / 2;
            p.sl_k = k;
            CARD_LAUNCH(r, "card_sample_coop16_kernel", true,
                        coop16_enqueue(p, (unsigned)grid, stream));
        }
        return CARD_OK;
    }
    // Between one and two tiles per CU (the reference's Year minibatch: 12 800 chains = 400 tiles): the time-sliced
    // paired form, see card_sample_coop2_kernel.  n_single CUs' worth of tiles run alone in each slice; every tile
    // does so in exactly one of the K slices.
    if (s->coop_blocks == 2 && !s->tune.coop_no2 && !p.eps_only && tiles_total > s->num_cu &&
        tiles_total < 2LL * s->num_cu && n_steps >= 64 && ybuf_fits) {
        const long long n_single = 2LL * s->num_cu - tiles_total;
        const int K = (int)((tiles_total + n_single - 1) / n_single);
        const int S = (int)((double)n_steps / ((double)(K - 1) + kRho32) + 0.5);
        const int L = n_steps - (K - 1) * S;
        if (K <= 8 && S >= 8 && L >= S) {
            p.ybuf = p.ws;
            p.sl_K = K; p.sl_nsingle = (int)n_single; p.sl_S = S; p.sl_L = L;
            for (int k = 0; k < K; ++k) {
                long long n_s = tiles_total - (long long)k * n_single;
                n_s = n_s > n_single ? n_single : n_s;
                const long long grid = n_s + (tiles_total - n_s + 1) / 2;
                p.sl_k = k;
                CARD_LAUNCH(r, "card_sample_coop2_kernel", true,
                            card_sample_coop2_kernel<<<dim3((unsigned)grid), dim3(kCoop2Threads), (size_t)kCoop2LdsFloats * sizeof(float), stream>>>(p));
            }
            return CARD_OK;
        }
    }
    const long long cap = (long long)s->num_cu * s->coop_blocks;
    const long long grid = tiles_total < cap ? tiles_total : cap;
    CARD_LAUNCH(r, "card_sample_coop_kernel", true,
                card_sample_coop_kernel<<<dim3((unsigned)grid), dim3(kCoopThreads), (size_t)kCoopLdsFloats * sizeof(float), stream>>>(p));
    return CARD_OK;
}

// what guided_small costs per timestep for n chains (n below a cooperative round): one 16-chain tile per CU, then the
// time-sliced forms up to two 16-chain tiles (5.4 us) and up to two 32-chain tiles per CU
double guided_small_cost(long long n, long long lone, long long round2) {
    constexpr double kCostTwo16 = 5.3;
    if (n <= lone) return kCostLone16;
    if (n <= 2 * lone) return kCostLone16 + (kCostTwo16 - kCostLone16) * (double)(n - lone) / (double)lone;
    return kCostTwo16 + (kCostCoopRound - kCostTwo16) * (double)(n - 2 * lone) / (double)(round2 - 2 * lone);
}

}  // namespace

int guided_run(const card_sampler* s, const CardRun& r, hipStream_t stream) {
    int arith = r.arith;
    if (arith == CARD_ARITH_F32_MFMA && s->tune.arith_split) arith = CARD_ARITH_SPLIT_BF16;
    if (arith == CARD_ARITH_SPLIT_BF16) return split_run(s, r, stream);
    RunParams p;
    memset(&p, 0, sizeof(p));
    p.wpk = s->wpk; p.gtab = s->gtab; p.coef = r.coef; p.x = r.x; p.y0hat = r.y0hat; p.ytmean = r.ytmean;
    p.y_init = r.y_init; p.noise = r.noise; p.out = r.out; p.state_slot = r.state_slot;
    p.N = r.N; p.seed = r.seed; p.elem_offset = r.elem_offset;
    p.T = r.T; p.t_hi = r.t_hi; p.t_lo = r.t_lo;
    p.x_dim = s->x_dim; p.cat_x = s->cat_x; p.cat_y_pred = s->cat_y_pred;
    p.out_mode = r.out_mode; p.eps_only = r.eps_only; p.n_z = r.n_z;
    p.w1s = s->w1s; p.ws = s->ws; p.nqs = s->nqs;
    p.wpk16 = s->wpk16; p.w1s16 = s->w1s16;
    // card_sampler_run_static: scratch comes from the library's static array; `scratch_bytes` bounds what a form may use
    // of it (the sampler's own allocation holds the wave slots, which are larger than any hand-over buffer)
    unsigned long long scratch_bytes = (unsigned long long)s->num_cu * kWaves * kSlotFloats * sizeof(float);
    if (r.lib_scratch) {
        p.ws = static_cast<float*>(r.workspace);
        scratch_bytes = r.workspace_bytes;
    }
    const int n_steps = p.t_hi - p.t_lo + 1;
    if (r.plan)      // layers 2-3, the K = 2 update of layer 1, lin4 (VALU), and the hoisted part once per tile and call
        r.plan->executed = 2.0 * kH * (2 * kH) + 2.0 * kH * 2 + 2.0 * kH + 2.0 * kH * 8 * s->nqs / (double)n_steps;
    // A call is cut into [0, n_pers): whole rounds of the persistent kernel (num_cu * 8 tiles, two waves per SIMD);
    // [n_pers, n_pers + n_coop): whole rounds of the cooperative kernel (two 32-chain tiles per CU, 9.0 ms per 1000
    // timesteps and round against 18.0 / 33.6 ms per 4 / 8 tiles per CU on the persistent kernel: ahead up to 3.5
    // rounds); the rest, less than a cooperative round: the small-batch forms (guided_small).  100 000 chains are one
    // persistent round + two cooperative rounds + 1 696 chains on 16-chain tiles, 54 ms per 1000 timesteps instead of the
    // 68 ms of two persistent rounds.  Results do not depend on the cut: the kernels are bit-identical per chain
    // (tests/test_gpu_sliced.py) and the noise is keyed by the chain's element index.
    const long long round8 = (long long)s->num_cu * kWaves * kTileM;
    long long n_pers = p.N, n_coop = 0;
    if (s->coop_blocks > 0) {
        const long long round2 = (long long)s->num_cu * s->coop_blocks * kTileM, lone = (long long)s->num_cu * kT16;
        const long long k8 = p.N / round8, rest = p.N - k8 * round8, m = rest / round2, tail = rest - m * round2;
        const double c_pers = (rest == 0 ? 0.0 : rest <= round8 / 2 ? kCostPersHalf : kCostPers) * n_steps + (rest && !k8 ? kCostLaunch : 0.0);
        const double c_coop = ((double)m * kCostCoopRound + (tail ? guided_small_cost(tail, lone, round2) : 0.0)) * n_steps +
                              ((m ? 1 : 0) + (tail ? 1 : 0)) * kCostLaunch;
        const double whole = (double)k8 * kCostPers * n_steps + kCostLaunch;
        if (s->tune.coop_force) n_pers = 0;
        else if (rest > 0 && c_pers - c_coop >= kPlanMinGain * ((k8 ? whole : 0.0) + c_pers)) n_pers = k8 * round8;
        n_coop = (p.N - n_pers) / round2 * round2;
    }
    // Persistent kernel: launches of whole waves-worth of tiles so that one launch stays well under a second and every wave
    // gets the same number of tiles (no intra-launch tail except in the last one).
    // in-kernel row tiling with n_z >= 32: the hoisted layer-1 tile has two distinct columns -> LDS-resident form (HOIST = 2)
    const bool two_col = r.n_z >= 32;
    if (n_pers > 0 && !two_col && r.lib_scratch) {
        // only this form needs the per-wave slots, 16 KB per resident wave
        const unsigned long long need = (unsigned long long)s->num_cu * kWaves * kSlotFloats * sizeof(float);
        if (need > r.workspace_bytes)
            return card_fail(CARD_E_UNSUPPORTED, "card_sampler_run_static: %d CUs need %llu bytes of wave slots, the static scratch has %llu",
                             s->num_cu, need, r.workspace_bytes);
    }
    const long long tiles_per_wave = 16;
    const long long chunk = round8 * tiles_per_wave;
    for (long long b = 0; b < n_pers; b += chunk) {
        p.n_begin = b;
        p.n_end = b + chunk < n_pers ? b + chunk : n_pers;
        const long long tiles = (p.n_end - p.n_begin + kTileM - 1) / kTileM;
        const long long grid = tiles < s->num_cu ? tiles : s->num_cu;
        if (two_col)
            CARD_LAUNCH(r, "card_sample_guided_kernel<2>", true, card_sample_guided_kernel<2><<<dim3((unsigned)grid), dim3(kThreads), s->lds_bytes, stream>>>(p));
        else
            CARD_LAUNCH(r, "card_sample_guided_kernel<1>", true, card_sample_guided_kernel<1><<<dim3((unsigned)grid), dim3(kThreads), s->lds_bytes, stream>>>(p));
    }
    if (n_coop > 0) {
        p.n_begin = n_pers;
        p.n_end = n_pers + n_coop;
        CARD_LAUNCH(r, "card_sample_coop_kernel", true,
                    card_sample_coop_kernel<<<dim3((unsigned)(s->num_cu * s->coop_blocks)), dim3(kCoopThreads), (size_t)kCoopLdsFloats * sizeof(float), stream>>>(p));
    }
    if (n_pers + n_coop < p.N) return guided_small(s, r, p, n_pers + n_coop, p.N, scratch_bytes, stream);
    return CARD_OK;
}

