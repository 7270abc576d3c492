// card_split.hip - CARD_ARITH_SPLIT_BF16: the regression sampler (CARD_VARIANT_GUIDED, hidden 128, y_dim 1) with the two
// 128 x 128 layers on the bf16 matrix pipe at fp32 accuracy ("split" arithmetic), an opt-in alternative to the FP32-MFMA
// kernels of card_guided.hip (regression/model.py:37-51, regression/diffusion_utils.py:57-119).
//
// Why.  On gfx950 v_mfma_f32_32x32x2_f32 runs at 256 flop/cycle/CU and shares the SIMD with the epilogue's VALU work
// (tools/ubench: ~0.82 of the FP32-MFMA peak is that kernel's ceiling); v_mfma_f32_32x32x16_bf16 runs 16 x faster and
// overlaps with VALU work.  A float is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits, cut with
// round-to-nearest: v_cvt_pk_bf16_f32), bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so
//     a.b = ah.bh + (ah.bm + am.bh) + (am.bm + ah.bl + al.bh) + [am.bl + al.bm + al.bl]
// with the bracket below 2^-24 |a||b|: six bf16 MFMAs per 16-deep k-step in place of eight fp32 MFMAs, 192 cycles
// instead of 512.  The three groups go to three accumulators (largest terms alone, corrections added last), which
// measures MORE accurate than the fp32 MFMA chain itself (tools/ubench/splitbf16.hip: K = 128 dot products, worst
// error 0.9e-7 of sum|a_k b_k| against 2.7e-7; three products only -- "bf16x3" -- would be 2.7e-6 and is not used).
//
// Form.  Three split copies of W2 and W3 are 192 KB -- more than LDS -- so the weights live in REGISTERS, cooperatively:
// the four waves of a workgroup (one per SIMD, up to 512 registers each) share tiles of 32 chains; wave w owns output
// features [32w, 32w + 32) of every layer and holds the A fragments of its rows (2 layers x 8 k-steps x 3 parts x 4
// registers).  After a layer each wave applies gamma / softplus to its 32 x 32 block (the accumulator layout
// D[feature (r&3) + 8(r>>2) + 4(lane>>5)][chain lane&31]), cuts it into the three bf16 parts and publishes them to LDS
// ALREADY in B-fragment order: k-step s = 2w + c of the next layer takes, on lane half h, the eight features that
// accumulator registers 8c .. 8c+7 of wave w hold on lane half h, so producer and consumer both address LDS as
// base + lane * 16 (one conflict-free ds_write_b128 / ds_read_b128 per fragment) and the matching input-feature order is
// baked into the packed weights.  G = 1..3 tiles advance in lockstep between the three workgroup barriers of a
// timestep; layer 1 (hoisted static part + W1[:,0] y_t), lin4 (per-wave partial sums exchanged through LDS), the
// posterior update and the softplus epilogues stay in fp32 on the VALU exactly as in card_guided.hip.  Noise: the same
// Philox stream per chain as every other kernel (generated once per chain and four draws, by one lane, into LDS).
#include "card_internal.h"

using namespace card;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef CARD_SPLIT_PROFILE
__device__ long long g_split_prof[8 * 8];
#define SPT(i) do { pts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SPT(i) do { } while (0)
#endif

namespace {

constexpr int kH = 128;
constexpr int kWaves = 4;
constexpr int kTileM = 32;
constexpr int kGRow = 3 * kH;
constexpr int kPartBytes = 8 * 64 * 16;              // one bf16 part of a tile's activations: 8 k-steps x 64 lanes x 16 B
constexpr int kActBytes = 3 * kPartBytes;            // 24 KB
// small tables as they sit in the packed image of card_guided.hip from kOffB on: b1 | b2 | b3 | W4 ln2 | b4 (+3) | W1[:,0]
constexpr int kTabB = 0, kTabW4 = 3 * kH, kTabB4 = 4 * kH, kTabW1c = 4 * kH + 4, kTabFloats = 5 * kH + 4;
constexpr long long kGroupsPerWg = 96;               // groups of three tiles a workgroup walks per launch

__device__ __forceinline__ float asf(unsigned u) { return __builtin_bit_cast(float, u); }

// two floats -> the packed bf16 pairs (low half = first value) of their high, middle and low parts
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 hf = {asf(h << 16), asf(h & 0xffff0000u)};
    const f32x2 r = v - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 mf = {asf(m << 16), asf(m & 0xffff0000u)};
    const f32x2 r2 = r - mf;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

}  // namespace

struct SplitParams {
    const unsigned short* wsplit;   // [layer 2][wave 4][part 3][k-step 8][lane 64][8] bf16
    const float* tab;               // kTabFloats
    const float* w1;                // (128, d_in) lin1.weight, row-major
    const float* gtab;              // (n_embed, 3, 128) gamma * log2e
    const float* coef;
    const float* x;
    const float* y0hat;
    const float* ytmean;
    const float* y_init;
    const float* noise;
    float* out;
    const int* state_slot;
    long long N, n_begin, n_end;
    unsigned long long seed, elem_offset;
    int T, t_hi, t_lo;
    int x_dim, cat_x, cat_y_pred, d_in;
    int out_mode, eps_only, n_z;
};

namespace {

// gamma multiply + softplus (log2 units) of a 32 x 32 block, in place
__device__ __forceinline__ void softplus_block(f32x16& a, const float (&g)[16]) {
    __builtin_amdgcn_s_setprio(3);           // VALU stretches go first: the other wave of the SIMD is (ideally) in its MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = softplus_l2(a[r] * g[r]);
}

// cut a block into its three bf16 parts and publish them as the two B fragments (k-steps 2w, 2w + 1) of the next layer
__device__ __forceinline__ void publish_split(const f32x16& a, char* dst /* tile base + (2w * 64 + lane) * 16 */) {
    u32x4 h[2], m[2], l[2];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
        unsigned hh_, mm_, ll_;
        split2(a[2 * pp], a[2 * pp + 1], hh_, mm_, ll_);
        h[pp >> 2][pp & 3] = hh_; m[pp >> 2][pp & 3] = mm_; l[pp >> 2][pp & 3] = ll_;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        *reinterpret_cast<u32x4*>(dst + c * 1024) = h[c];
        *reinterpret_cast<u32x4*>(dst + kPartBytes + c * 1024) = m[c];
        *reinterpret_cast<u32x4*>(dst + 2 * kPartBytes + c * 1024) = l[c];
    }
    __builtin_amdgcn_s_setprio(0);
}

// one 128 -> 32 (this wave's rows) layer of a tile: 8 k-steps x 6 bf16 MFMAs into three accumulators
__device__ __forceinline__ f32x16 layer_split(const bf16x8 (&W)[3][8], const char* act /* tile base + lane * 16 */, const float* tbl) {
    // two accumulators: the bias and the high x high products, and the five correction terms (smallest first within a
    // k-step); they meet once at the end, so the corrections are not rounded away one by one against the large sum
    f32x16 hi, co;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = lds4(tbl + 8 * q);
        hi[4 * q + 0] = b[0]; hi[4 * q + 1] = b[1]; hi[4 * q + 2] = b[2]; hi[4 * q + 3] = b[3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) co[r] = 0.0f;
    // the B fragments of k-step s + 1 are requested before the six MFMAs of k-step s are issued (192 cycles of cover for
    // the LDS latency); the scheduling barriers keep that order (left alone, the scheduler sinks every ds_read next to its
    // first use and each k-step waits out the full latency)
    bf16x8 bh = *reinterpret_cast<const bf16x8*>(act);
    bf16x8 bm = *reinterpret_cast<const bf16x8*>(act + kPartBytes);
    bf16x8 bl = *reinterpret_cast<const bf16x8*>(act + 2 * kPartBytes);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        bf16x8 nh = bh, nm = bm, nl = bl;
        if (s < 7) {
            nh = *reinterpret_cast<const bf16x8*>(act + (s + 1) * 1024);
            nm = *reinterpret_cast<const bf16x8*>(act + kPartBytes + (s + 1) * 1024);
            nl = *reinterpret_cast<const bf16x8*>(act + 2 * kPartBytes + (s + 1) * 1024);
        }
        co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[2][s], bh, co, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0][s], bh, hi, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0][s], bl, co, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1][s], bm, co, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1][s], bh, co, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0][s], bm, co, 0, 0, 0);
        if (s < 7) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);      // the three ds_reads first,
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                 // then the six MFMAs
        __builtin_amdgcn_sched_barrier(0);
        bh = nh; bm = nm; bl = nl;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) hi[r] += co[r];
    return hi;
}

// ------------------------------------------------------------------------------------------------
// The kernel.  Eight waves per workgroup, two per SIMD, one workgroup per CU:
//   waves 0-3 ("front"):  hold the A fragments of W2 and the chains' states.  Per interval: the posterior update of one
//                         tile from the eps that its lin4 partial sums give and layer 1 of its next timestep (VALU
//                         only), then layer 2 of a second tile
//   waves 4-7 ("back"):   hold the A fragments of W3.  Per interval: layer 3 + this wave's lin4 partial sums of a third
//                         tile, the noise of its coming draws and the gamma / coefficient rows of its next timestep
// (the update + layer-1 stage sat on the back waves at first: cycle counters around the stages -- tools/prof_split_phases.py
// on a -DCARD_SPLIT_PROFILE build -- showed them working 7 900 cycles per interval slot while the front waves worked 3 100
// and waited 5 000; moved over, the slot is 6 000 / 4 400 cycles of work and the kernel 8 % faster)
// Three tiles (32 chains each) rotate through the three stages, one workgroup barrier per interval, so every interval
// advances every tile by a third of a timestep and a timestep costs three intervals whatever the stage.  The two waves
// of a SIMD are independent instruction streams: while one sits in its 48 MFMAs the other issues its gamma / softplus /
// split / publish VALU work, which is how the matrix pipe's shadow gets used -- inside ONE wave nothing overlaps (a wave
// issues in order; tools/ubench/interleave_bf16.hip), and getting the compiler to weave two tiles' instruction streams
// together by hand-placed scheduling groups was tried first and lost to spills and hazard nops.
// LDS: layer-1 output (3 bf16 parts) of the tile in stage 1 is written while that of the tile in stage 2 is read, same
// for the layer-2 output one stage later -> two ping-pong pairs (4 x 24 KB) serve three tiles; the hoisted layer-1
// tiles (3 x 16 KB, fp32) live in LDS too, so a wave carries 96 weight registers + its working set (< 256).
// ------------------------------------------------------------------------------------------------
constexpr int kSets = 3;
constexpr int kThreads8 = 512;
constexpr int kC1Floats = kH * kTileM;                // one tile's hoisted layer-1 block: [wave][q][lane][4]
constexpr int kRowFloats = kGRow + CARD_COEF_COLS;   // one timestep's gamma rows (x log2e) + its coefficient row
constexpr int kLdsBytes = 4 * kActBytes + (kSets * kC1Floats + kSets * kWaves * kTileM + kSets * 4 * kTileM + kSets * 2 * kRowFloats + kTabFloats) * 4;

__global__ __launch_bounds__(kThreads8) void card_sample_split_kernel(const SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* const actX = lds_raw;                                               // [2] layer-1 output, split (ping-pong by interval)
    char* const actY = actX + 2 * kActBytes;                                  // [2] layer-2 output, split
    float* const c1l = reinterpret_cast<float*>(actY + 2 * kActBytes);        // [3] hoisted layer-1 tiles
    float* const red = c1l + kSets * kC1Floats;                               // [3][4][32] lin4 partial sums
    float* const zbuf = red + kSets * kWaves * kTileM;                        // [3][4][32] normals of a tile's current draw block
    float* const gsl = zbuf + kSets * 4 * kTileM;                             // [3][2] gamma + coefficient rows of a tile's current / next timestep
    float* const tab = gsl + kSets * 2 * kRowFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int back = wave >> 2, w = wave & 3;                                 // stage group, feature block
    const int m = lane & 31, hh = lane >> 5;
    for (int i = tid; i < kTabFloats; i += kThreads8) tab[i] = p.tab[i];

    // this wave's A fragments: rows 32w .. 32w+31 of W2 (front waves) or W3 (back waves), three parts x eight k-steps,
    // pinned to the accumulation half of the register file (an MFMA reads its A operand from there directly; left to
    // itself the allocator parks them there anyway and copies four registers back before every use)
    bf16x8 Wf[3][8];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            Wf[pt][s] = *reinterpret_cast<const bf16x8*>(p.wsplit + ((((back * kWaves + w) * 3 + pt) * 8 + s) * 64 + lane) * 8);
            asm volatile("" : "+a"(Wf[pt][s]));
        }
    __syncthreads();
    // accumulator register r = 4q + k of this lane holds feature 32w + 8q + 4hh + k: per-feature tables are read as
    // four b128 at (table + 32w + 4hh) + 8q
    const int foff = 32 * w + 4 * hh;
    const float b4 = tab[kTabB4];
    float* const c1w = c1l + (w * 4 * 64 + lane) * 4;                         // + set * kC1Floats + q * 256
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;
    const int n_steps = p.t_hi - p.t_lo + 1;
    const long long n_tiles = (p.n_end - p.n_begin + kTileM - 1) / kTileM;
    const long long n_groups = (n_tiles + kSets - 1) / kSets;
    const bool rng = !p.noise && !p.eps_only;

    auto load16 = [&](const float* src, float (&dst)[16]) {              // 16 per-feature values of this lane
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 8 * q);
            dst[4 * q + 0] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
        }
    };

    for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        bool has[kSets];
        int tile0[kSets];                                                     // first chain of the tile, relative to n_begin (a launch covers < 2^31 chains)
        float mu[kSets], y[kSets], z[kSets];
        const int n_here = (int)(p.n_end - p.n_begin);
#pragma unroll
        for (int g = 0; g < kSets; ++g) {
            const long long tile = grp * kSets + g;
            has[g] = tile < n_tiles;
            tile0[g] = (int)(tile * kTileM);
            y[g] = 0.0f; z[g] = 0.0f; mu[g] = 0.0f;
        }
        auto rci0 = [&](int g) { const int r = tile0[g] + m; return r < n_here ? r : n_here - 1; };
        // one lane per chain draws that chain's four normals of a block: back wave w serves chains 8w .. 8w+7 of a tile
        auto gen_block = [&](int g, uint32_t blk) {
            if (lane < 8) {
                const int n = 8 * w + lane;
                long long rw = p.n_begin + (grp * kSets + g) * kTileM + n;
                rw = rw < p.n_end ? rw : p.n_end - 1;
                float z4[4];
                philox_normal4(p.seed, p.elem_offset + (unsigned long long)rw, blk, z4);
                float* dst = zbuf + (g * 4) * kTileM + n;
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k * kTileM] = z4[k];
            }
        };
        // gamma rows + coefficient row of timestep t: 98 lanes x four floats (the back waves fetch a tile's next row while
        // they run its layer 3, and park it in the other buffer at the end of that stage)
        auto row_fetch = [&](int t) -> f32x4 {
            const int bt = tid - 4 * 64;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bt >= 0 && bt < kGRow / 4) v = *reinterpret_cast<const f32x4*>(p.gtab + (long long)t * kGRow + bt * 4);
            else if (bt >= kGRow / 4 && bt < kRowFloats / 4) v = *reinterpret_cast<const f32x4*>(p.coef + (long long)t * CARD_COEF_COLS + (bt - kGRow / 4) * 4);
            return v;
        };
        auto row_park = [&](int g, int buf, const f32x4& v) {
            const int bt = tid - 4 * 64;
            if (bt >= 0 && bt < kRowFloats / 4) *reinterpret_cast<f32x4*>(gsl + (g * 2 + buf) * kRowFloats + bt * 4) = v;
        };
        if (!back) {
            // hoisted layer 1: c1 = b1 + W1[:, 1:] . (y_0_hat, x) for this lane's 16 features (regression/model.py:38-48),
            // fp32 on the VALU, into LDS
            float c1[kSets][16], y0h[kSets];
            long long rr[kSets];
#pragma unroll
            for (int g = 0; g < kSets; ++g) {
                const long long rc = p.n_begin + rci0(g);
                rr[g] = p.n_z > 1 ? rc / p.n_z : rc;
                y0h[g] = p.y0hat[rr[g]];
                load16(tab + kTabB + foff, c1[g]);
            }
            for (int k = 1; k < p.d_in; ++k) {
                float wk[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) wk[r] = p.w1[(32 * w + (r & 3) + 8 * (r >> 2) + 4 * hh) * p.d_in + k];
#pragma unroll
                for (int g = 0; g < kSets; ++g) {
                    float feat;
                    if (p.cat_y_pred && k == 1) {
                        feat = y0h[g];
                    } else {
                        const int xi = k - 1 - (p.cat_y_pred ? 1 : 0);
                        feat = (p.cat_x && xi < p.x_dim) ? p.x[rr[g] * p.x_dim + xi] : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1[g][r] = fmaf(wk[r], feat, c1[g][r]);
                }
            }
#pragma unroll
            for (int g = 0; g < kSets; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(c1w + g * kC1Floats + q * 256) = f32x4{c1[g][4 * q], c1[g][4 * q + 1], c1[g][4 * q + 2], c1[g][4 * q + 3]};
#pragma unroll
            for (int g = 0; g < kSets; ++g) mu[g] = p.ytmean[rr[g]];
        } else {
            const f32x4 r0 = row_fetch(p.t_hi);
#pragma unroll
            for (int g = 0; g < kSets; ++g) {
                row_park(g, 0, r0);
                if (rng && !p.y_init) gen_block(g, 0u);                       // draw 0 makes y_T
            }
        }
        __syncthreads();
        // Lane-derived indices and addresses are re-derived from a laundered copy of the lane number at the top of every
        // loop trip: hoisted out of the loop they stay live across it (a dozen registers next to the 96 weight registers
        // and the working set) and end up spilled, each reload a memory round trip in the middle of a stage.
        struct { int lane, m, hh, foff, pub, rd; } li;
        auto relane = [&]() {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            li.lane = ln; li.m = ln & 31; li.hh = ln >> 5; li.foff = 32 * w + 4 * li.hh;
            li.pub = (2 * w * 64 + ln) * 16; li.rd = ln * 16;
        };
        relane();
        auto rowi = [&](int g) { return tile0[g] + li.m; };
        auto rci = [&](int g) { const int r = tile0[g] + li.m; return r < n_here ? r : n_here - 1; };
        auto store_state = [&](int g, int s, float v) {
            long long slot;
            if (p.out_mode == CARD_OUT_ALL) slot = s - slot_base;
            else if (p.out_mode == CARD_OUT_SELECT) slot = p.state_slot[s];
            else return;
            float* const ob = p.out + slot * p.N + p.n_begin;                 // scalar base + 32-bit lane offset: no 64-bit lane addresses
            if (slot >= 0 && rowi(g) < n_here && li.hh == 0 && w == 0) ob[rowi(g)] = v;
        };
        if (!back) {
#pragma unroll
            for (int g = 0; g < kSets; ++g) {
                if (p.y_init) {
                    y[g] = (p.y_init + p.n_begin)[rci(g)];
                } else {
                    const float z0 = p.noise ? (p.noise + p.n_begin)[rci(g)] : zbuf[(g * 4 + 0) * kTileM + m];
                    y[g] = z0 + mu[g];                                         // diffusion_utils.py:109-110
                    if (has[g]) store_state(g, 0, y[g]);
                }
            }
        }

        // ---- the three stages of a tile's timestep; i = index of the timestep in this call (t = t_hi - i); the rows of
        // timestep i sit in buffer i & 1 of the tile
        // front waves, stage 0: close timestep i - 1 (eps, posterior update), open timestep i (layer 1)
        auto stage_update_first = [&](int g, int i, int par) {
            if (!has[g] || i < 0 || i > n_steps) return;
            if (i >= 1) {
                const int t = p.t_hi - (i - 1), d = T - t;
                const float* rg = red + g * kWaves * kTileM + li.m;
                const float eps = (((rg[0] + rg[kTileM]) + rg[2 * kTileM]) + rg[3 * kTileM]) + b4;
                if (p.eps_only) {
                    if (rowi(g) < n_here && li.hh == 0 && w == 0) (p.out + p.n_begin)[rowi(g)] = eps;
                    return;
                }
                // regression/diffusion_utils.py:83-89 (t >= 1) / :101-103 (t == 0)
                const float* c = gsl + (g * 2 + ((i - 1) & 1)) * kRowFloats + kGRow;
                const f32x4 ca = lds4(c), cb = lds4(c + 4);
                const float y0r = ca[CARD_COEF_INV_SQRT_AB] * ((y[g] - ca[CARD_COEF_ONE_M_SQRT_AB] * mu[g]) - eps * ca[CARD_COEF_SOMAB]);
                if (t >= 1) {
                    const float zz = p.noise ? z[g] : zbuf[(g * 4 + (d & 3)) * kTileM + li.m];
                    const float yh = (ca[CARD_COEF_G0] * y0r + cb[CARD_COEF_G1 - 4] * y[g]) + cb[CARD_COEF_G2 - 4] * mu[g];
                    y[g] = yh + cb[CARD_COEF_SIGMA - 4] * zz;
                } else {
                    y[g] = y0r;
                }
                store_state(g, T - t, y[g]);
            }
            if (i < n_steps) {
                const int t = p.t_hi - i;
                if (p.noise && t >= 1 && !p.eps_only) z[g] = (p.noise + (long long)(T - t - draw0) * p.N + p.n_begin)[rci(g)];   // used three intervals on
                float gam[16], w1c[16], c1[16];
                load16(gsl + (g * 2 + (i & 1)) * kRowFloats + li.foff, gam);
                load16(tab + kTabW1c + li.foff, w1c);
                const float* c1r = c1l + g * kC1Floats + (w * 4 * 64 + li.lane) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = lds4(c1r + q * 256);
                    c1[4 * q + 0] = v[0]; c1[4 * q + 1] = v[1]; c1[4 * q + 2] = v[2]; c1[4 * q + 3] = v[3];
                }
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = fmaf(w1c[r], y[g], c1[r]);
                softplus_block(a, gam);
                publish_split(a, actX + par * kActBytes + li.pub);
            }
        };
        // front waves, stage 1: layer 2
        auto stage_second = [&](int g, int i, int par) {
            if (!has[g] || i < 0 || i >= n_steps) return;
            float gam[16];
            load16(gsl + (g * 2 + (i & 1)) * kRowFloats + kH + li.foff, gam);
            f32x16 a = layer_split(Wf, actX + (par ^ 1) * kActBytes + li.rd, tab + kTabB + kH + li.foff);
            softplus_block(a, gam);
            publish_split(a, actY + par * kActBytes + li.pub);
        };
        // back waves, stage 2: layer 3, this wave's lin4 partial sums, the coming draws, the next timestep's rows
        auto stage_third = [&](int g, int i, int par) {
            if (!has[g] || i < 0 || i >= n_steps) return;
            const int t = p.t_hi - i, d = T - t;
            const bool more = i + 1 < n_steps;
            f32x4 nxt = {0.f, 0.f, 0.f, 0.f};
            if (more) nxt = row_fetch(t - 1);
            float gam[16], w4r[16];
            load16(gsl + (g * 2 + (i & 1)) * kRowFloats + 2 * kH + li.foff, gam);
            load16(tab + kTabW4 + li.foff, w4r);
            f32x16 a = layer_split(Wf, actY + (par ^ 1) * kActBytes + li.rd, tab + kTabB + 2 * kH + li.foff);
            softplus_block(a, gam);
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(a[r], w4r[r], part);
            part += __shfl_xor(part, 32, 64);
            if (li.hh == 0) red[(g * kWaves + w) * kTileM + li.m] = part;
            __builtin_amdgcn_s_setprio(0);
            // the update that closes this timestep (next interval) consumes draw d: a new block of four starts at
            // d % 4 == 0; a call that starts mid-loop (y_init) has not drawn its first block yet
            if (rng && t >= 1 && ((d & 3) == 0 || (i == 0 && p.y_init))) gen_block(g, (uint32_t)d >> 2);
            if (more) row_park(g, (i + 1) & 1, nxt);
        };

        // interval 3I + k: tile k opens timestep I, tile k - 1 is in layer 2, tile k - 2 in layer 3 (of timestep I or I - 1)
        // (ping-pong parity of the exchange buffers = parity of the interval number 3I + k)
        for (int I = 0; I <= n_steps; ++I) {
            const int e = I & 1, o = e ^ 1;
#ifdef CARD_SPLIT_PROFILE
            long long pts[7];
#endif
            relane();
            SPT(0);
            if (back) { stage_third(1, I - 1, e); } else { stage_update_first(0, I, e); stage_second(2, I - 1, e); }
            SPT(1);
            __syncthreads();
            SPT(2);
            if (back) { stage_third(2, I - 1, o); } else { stage_update_first(1, I, o); stage_second(0, I, o); }
            SPT(3);
            __syncthreads();
            SPT(4);
            if (back) { stage_third(0, I, e); } else { stage_update_first(2, I, e); stage_second(1, I, e); }
            SPT(5);
            __syncthreads();
            SPT(6);
#ifdef CARD_SPLIT_PROFILE
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 6; ++i) g_split_prof[wave * 8 + i] += pts[i + 1] - pts[i];
#endif
        }
        relane();
        if (p.out_mode == CARD_OUT_LAST && !p.eps_only && !back) {
#pragma unroll
            for (int g = 0; g < kSets; ++g)
                if (has[g] && rowi(g) < n_here && li.hh == 0 && w == 0) (p.out + p.n_begin)[rowi(g)] = y[g];
        }
    }
}

// create-time: the three bf16 parts of W2 ln2 and W3 ln2 in A-fragment order.  Lane (m, h) of wave w, k-step s = 2w' + c,
// slot j holds W[32w + m][32w' + (j&3) + 8(2c + (j>>2)) + 4h]: the feature accumulator register 8c + j of wave w' carries
// on lane half h (see publish_split).
__global__ void split_pack_kernel(const float* __restrict__ W2, const float* __restrict__ W3, unsigned short* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * kWaves * 3 * 8 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 7, rest = idx >> 12;
    const int pt = rest % 3, w = (rest / 3) % kWaves, layer = rest / (3 * kWaves);
    const int mm = lane & 31, h = lane >> 5, wp = s >> 1, c = s & 1;
    const int f = 32 * wp + (j & 3) + 8 * (2 * c + (j >> 2)) + 4 * h;
    const float v = (layer ? W3 : W2)[(32 * w + mm) * kH + f] * kLn2;
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    const __bf16 lo = (__bf16)r2;
    const __bf16 pick = pt == 0 ? hi : (pt == 1 ? mid : lo);
    out[idx] = __builtin_bit_cast(unsigned short, pick);
}


}  // namespace

int split_create(card_sampler* s, const card_model_desc* d, hipStream_t stream) {
    const size_t n = (size_t)2 * kWaves * 3 * 8 * 64 * 8;
    HIP_TRY(hipMalloc(&s->wsplit, n * sizeof(unsigned short)));
    HIP_TRY(hipMalloc(&s->W1, (size_t)kH * s->d_in * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(s->W1, d->W[0], (size_t)kH * s->d_in * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipFuncSetAttribute((const void*)card_sample_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    split_pack_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d->W[1], d->W[2], s->wsplit);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

void split_destroy(card_sampler* s) {
    if (s->wsplit) (void)hipFree(s->wsplit);
    if (s->W1) (void)hipFree(s->W1);
}

int split_run(const card_sampler* s, const CardRun& r, hipStream_t stream) {
    SplitParams p;
    memset(&p, 0, sizeof(p));
    p.wsplit = s->wsplit; p.tab = s->wpk + 2 * kH * kH; p.w1 = s->W1; p.gtab = s->gtab; p.coef = r.coef;
    p.x = r.x; p.y0hat = r.y0hat; p.ytmean = r.ytmean; p.y_init = r.y_init; p.noise = r.noise; p.out = r.out;
    p.state_slot = r.state_slot; p.N = r.N; p.seed = r.seed; p.elem_offset = r.elem_offset;
    p.T = r.T; p.t_hi = r.t_hi; p.t_lo = r.t_lo; p.x_dim = s->x_dim; p.cat_x = s->cat_x; p.cat_y_pred = s->cat_y_pred;
    p.d_in = s->d_in; p.out_mode = r.out_mode; p.eps_only = r.eps_only; p.n_z = r.n_z;
    // a workgroup rotates three tiles through the three stages of a timestep; a launch covers kGroupsPerWg such groups
    // per CU (launches stay under ~0.5 s), fewer workgroups than CUs only when the batch has fewer groups
    const long long chunk = (long long)s->num_cu * kGroupsPerWg * kSets * kTileM;
    if (r.plan) r.plan->executed = 6.0 * 2 * kH * (2 * kH);      // bf16 flops: six partial products per k-step of the two hidden layers
    for (long long b = 0; b < r.N; b += chunk) {
        p.n_begin = b;
        p.n_end = b + chunk < r.N ? b + chunk : r.N;
        const long long groups = ((p.n_end - p.n_begin + kTileM - 1) / kTileM + kSets - 1) / kSets;
        const long long grid = groups < s->num_cu ? groups : s->num_cu;
        CARD_LAUNCH(r, "card_sample_split_kernel", true,
                    card_sample_split_kernel<<<dim3((unsigned)grid), dim3(kThreads8), (size_t)kLdsBytes, stream>>>(p));
    }
    return CARD_OK;
}

#ifdef CARD_SPLIT_PROFILE
// profile build only: cycles of workgroup 0, [wave][work0, wait0, work1, wait1, work2, wait2]
extern "C" CARD_API int card_debug_split_prof(long long* host64, int reset) {
    if (host64) HIP_TRY(hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_split_prof), sizeof(long long) * 64));
    if (reset) { long long z[64] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_split_prof), z, sizeof(z))); }
    return CARD_OK;
}
#endif
