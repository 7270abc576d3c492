// card_clscoop.hip - cooperative persistent kernel for the classification ConditionalModel at feature width 128
// (classification/model.py:79-94 after the hoisted encoder; classification/configs/mnist.yml:15, gaussmix.yml:16) with
// up to 32 classes.  On the GEMM back end such a head is five tiny launches per timestep (85 us, launch/latency
// bound); here one launch runs all T timesteps, in the form of card_guided.hip's cooperative kernel:
//   * the four waves of a workgroup (one per SIMD) share a 32-chain tile; wave w owns features [32w, 32w+32) of every
//     hidden layer and keeps its rows of W2, W3 (2 x 64 VGPRs), of lin1's y_t columns and of lin4 in registers
//   * y_t (C values per chain) lives in a chain-major LDS tile ys[32][36]; layer 1 = hoisted static part
//     c1 = b1 + W1[:, C:2C] . y_0_hat (once per tile) + W1[:, 0:C] . y_t on MFMA each step (class order 8q + 4h + e, so
//     one ds_read_b128 of ys feeds four k-steps), then softplus(BN1(gamma1 * .)) and the Hadamard product with this
//     wave's 32 features of x_feat (registers, loaded once per tile)
//   * layers 2, 3 as in the regression kernel (activation exchange through a chain-major LDS tile)
//   * lin4 on MFMA too: the accumulators of layer 3 are, register for register, the B operand of a 32-class x 32-chain
//     tile with K = this wave's 32 features (up to 16 classes: of four 16-class x 16-chain blocks of the four-block
//     16 x 16 x 1 MFMA, half the matrix-pipe time); the four waves' partial tiles are exchanged through LDS and added in wave
//     order; wave w then updates classes [8w, 8w+8) of every chain (posterior update in the reference's fp32 order,
//     noise per (chain, class) element) and writes them back to ys
// Eval-mode BatchNorm is the per-feature affine folded into the gamma table (scale) and a shift table; softplus runs in
// log2 units with ln2 folded into W2, W3, W4, exactly as in the other kernels.
#include "card_internal.h"

using namespace card;

namespace {

constexpr int kF = 128;            // feature width
constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kTileM = 32;
constexpr int kCmax = 32;          // classes, padded
constexpr int kCsmall = 16;        // up to here lin4 runs on the four-block 16 x 16 x 1 MFMA
constexpr int kExLd = kF + 4;      // activation exchange row stride (floats)
constexpr int kYLd = kCmax + 4;    // ys / partial-tile row stride
constexpr int kGRow = 3 * kF;
constexpr int kTab = 6 * kF + kCmax;                 // b1|b2|b3 | shift1|shift2|shift3 (log2 units) | b4
constexpr int kZFloats = 4 * kTileM * kYLd;          // normals of a tile's current draw block: [draw & 3][chain][class]
constexpr int kExBuf = kWaves * kTileM * kYLd;        // one activation exchange buffer [32][132], sized for its alias: the four
                                                      // lin4 partial tiles [4][32][36] (written after layer 2 has read the buffer,
                                                      // read before layer 1 writes it again)
constexpr int kC1Floats = kF * kTileM;                // hoisted layer-1 tile, [wave][q][lane][4]
constexpr int kExBuf1 = kTileM * kExLd;              // the second exchange buffer has no alias: [32][132] only
constexpr int kTileLdsFloats = kExBuf + kExBuf1 + kTileM * kYLd + 2 * kGRow + kZFloats + kC1Floats;   // per tile position
// LDS layout of a tile position per kernel form.  FORM 0: one workgroup per CU (one or two tiles), FORM 1: PAIR, FORM 2:
// PAIR compact (up to kCsmall classes, n_z >= 32): 20-float row pitch for y_t, the noise block and the lin4 partial tiles
// (16 classes + 4: conflict-free for 16 lanes of a b128 access), the hoisted layer-1 tile as the two columns of the at
// most two images a 32-chain tile spans, and with the 23 KB that frees the lin1 / lin4 fragments of the four waves.
constexpr int kC1Ld = kF + 8;                          // row pitch of the two-column hoisted layer-1 tile
template <int FORM> struct ClsLds {
    static constexpr bool compact = FORM == 2;
    static constexpr int YL = compact ? kCsmall + 4 : kYLd;
    static constexpr int EX0 = compact ? kExBuf1 : kExBuf;      // compact: the alias [4][32][20] fits the plain buffer
    static constexpr int Z = 4 * kTileM * YL;
    static constexpr int C1 = compact ? 2 * kC1Ld : kC1Floats;
    static constexpr int FR1 = compact ? kWaves * 2 * 256 : 0;  // lin1 fragments, nq1 <= 2: [wave][q][lane][4]
    static constexpr int FR4 = compact ? kWaves * 4 * 256 : 0;  // lin4 fragments: [wave][q][lane][4]
    static constexpr int XF = compact ? 2 * kC1Ld : 0;          // x_feat of the tile's two images
    static constexpr int tile = EX0 + kExBuf1 + kTileM * YL + 2 * kGRow + Z + C1 + FR1 + FR4 + XF;
};
static_assert(ClsLds<0>::tile == kTileLdsFloats && ClsLds<1>::tile == kTileLdsFloats, "plain layout");
constexpr int kPerCu = 2;                            // tile positions per CU: two one-tile workgroups (PAIR forms)
constexpr long long kChunkTiles = 8192;              // tiles per sliced chunk: bounds the hand-over buffer (262 144 chains)
__host__ __device__ constexpr int lds_floats(int form) { return (form == 2 ? ClsLds<2>::tile : kTileLdsFloats) + kTab; }
constexpr int kPrioValu = 3, kPrioMfma = 0;

}  // namespace

struct ClsCoopParams {
    const float* wpk;     // W2 | W3 fragment images (x ln2)
    const float* w1f;     // lin1 fragments: [part(dyn, stat)][wave][q][64][4]
    const float* w4f;     // lin4 fragments: [wave][q][64][4] (x ln2)
    const float* tabs;    // kTab floats
    const float* gtab;    // (n_embed, 3, 128) gamma * bn_scale * log2e
    const float* coef;
    const float* xfeat;   // (rows, 128)
    const float* y0hat;   // (rows, C)
    const float* ytmean;
    const float* y_init;  // (N, C) or NULL
    const float* noise;
    float* out;
    const int* state_slot;
    long long N;
    unsigned long long seed, elem_offset;
    int T, t_hi, t_lo;
    int C, nq1, guidance;
    int out_mode, eps_only, n_z;
    // time slicing (see the kernel): this launch covers chains [n_begin, n_end) and runs the slots [q0, q1); slot q is
    // segment q / n_tiles of tile q % n_tiles, the call's timesteps being cut into n_seg segments
    long long n_begin, n_end, q0, q1;
    int n_seg, g_eff;
    float* ybuf;          // (n_end - n_begin, C) chain states between the segments of a tile
};

#ifdef CARD_CLS_PROFILE
__device__ long long g_cls_prof[64];
#define PROF_T(i) do { pts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PROF_T(i) do { } while (0)
#endif

namespace {

// BN affine + softplus of a 32-feature x 32-chain tile, staged over its 16 registers (see card_guided.hip)
__device__ __forceinline__ void epilogue_bn(f32x16& acc, const float* gl, const float* sl) {
    // The stages are fenced with empty asm statements that take all 16 values at once: __builtin_amdgcn_sched_barrier
    // alone does not pin pure arithmetic (instruction selection orders it by data dependence only), and where a consumer
    // follows element by element -- the lin4 MFMAs after layer 3 -- the compiler had sunk each element's
    // fma -> exp -> add -> log -> med3 chain in front of its MFMA, every transcendental latency exposed.
    f32x16 xs, e;
    __builtin_amdgcn_s_setprio(kPrioValu);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 g = lds4(gl + 8 * q), sh = lds4(sl + 8 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[4 * q + k] = fmaf(acc[4 * q + k], g[k], sh[k]);
    }
    asm volatile("" : "+v"(xs));
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(xs[r]);
    asm volatile("" : "+v"(e));
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = 1.0f + e[r];
    asm volatile("" : "+v"(e));
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_logf(e[r]);
    asm volatile("" : "+v"(e));
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_fmed3f(e[r], xs[r], kSpThresh);
    asm volatile("" : "+v"(acc));
    __builtin_amdgcn_s_setprio(kPrioMfma);
}

// One workgroup = four waves (one per SIMD) that advance one tile with ONE copy of the weight fragments in registers.
// Two tile positions per CU keep its matrix pipes busy through the barriers and LDS waits of a tile.  Forms:
//   * ONE (FORM 0): at most num_cu tiles -- one workgroup per CU, up to 512 registers per wave, every fragment held.
//   * PAIR (FORM 1): two one-tile workgroups per CU (__launch_bounds__(256, 2): 256 registers per wave, two waves per
//     SIMD).  The 128 W2 / W3 fragment registers leave 128 for everything else, so the lin1 and lin4 fragments are not
//     held: they are re-read from L2 where they are used (two + four 16-byte loads per lane and timestep, covered by the other
//     wave of the SIMD), and the Philox block keeps ~29 values in scratch.  The hardware interleaves the two waves of a
//     SIMD instruction by instruction: every LDS wait, barrier and MFMA -> VALU dependency of one tile is filled with the
//     other tile's work.  17.6 vs 20.0 us per timestep for the reference's MNIST test minibatch against round 2's first
//     form, ONE workgroup advancing two tiles in its one instruction stream (a wave issues in order, so the second tile
//     filled the first one's waits only as far as the compiler interleaved the two; removed in round 3).
//   * PAIR compact (FORM 2; up to 16 classes and n_z >= 32, i.e. every F = 128 config of the reference): the LDS layout
//     ClsLds<2> -- 20-float row pitch, the hoisted layer-1 tile and x_feat as the two columns of the two images a tile
//     spans -- leaves room for the lin1 / lin4 fragments in LDS: no L2 re-reads, 14 values in scratch; 17.05 us.
//
// Time slicing.  A tile is NOT tied to a workgroup for the whole loop: the timesteps of a call are cut into n_seg
// segments and the (tile, segment) pairs are laid out as slots q = segment * n_tiles + tile; one launch ("slice") runs
// num_cu * 2 consecutive slots, one per resident workgroup, so every slice fills the chip whatever
// n_tiles is (250 images x 100 z = 782 tiles used to cost four rounds of 256 tiles, the last one with 14).  A tile's
// segments land in different slices in order (n_tiles >= slots per slice when slicing is on), stream order is the only
// synchronisation, and a chain's state travels through ybuf.  The arithmetic per chain and timestep does not depend
// on the form or on the slicing: results are bit-identical to the un-sliced ONE form.
template <int FORM = 0, bool MULTI = false>
__global__ __launch_bounds__(kThreads, FORM >= 1 ? 2 : 1) void card_sample_cls_coop_kernel(const ClsCoopParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = ClsLds<FORM>;
    constexpr bool PAIR = FORM >= 1, COMPACT = FORM == 2;
    constexpr int G = 1;      // tile positions per workgroup (round 2's two-tile workgroup lost to the PAIR forms and is gone;
                              // the per-position arrays below are what is left of it and fold away)
    constexpr int kTileLds = L::tile, YL = L::YL, EX0 = L::EX0;
    constexpr int kC1q = COMPACT ? 8 : 256;                 // distance between a lane's four quads of the hoisted layer-1 tile
    float* const tab = lds + G * kTileLds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hh = lane >> 5;
    const int C = p.C;
    for (int i = tid; i < kTab; i += kThreads) tab[i] = p.tabs[i];

    // register-resident A fragments of this wave: W2, W3 rows, lin1's y_t columns, lin4's columns
    float A2[64], A3[64], A1[16], A4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.wpk + ((w * 16 + i) * 64 + lane) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.wpk + kF * kF + ((w * 16 + i) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { A2[4 * i + e] = a[e]; A3[4 * i + e] = b[e]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (!PAIR) {                                         // PAIR: re-read from L2 where they are used (256 registers per wave)
            if (q < p.nq1) a = *reinterpret_cast<const f32x4*>(p.w1f + ((w * p.nq1 + q) * 64 + lane) * 4);
            b = *reinterpret_cast<const f32x4*>(p.w4f + ((w * 4 + q) * 64 + lane) * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { A1[4 * q + e] = a[e]; A4[4 * q + e] = b[e]; }
    }
    // COMPACT: this lane's lin1 / lin4 fragments live in LDS (each lane reads back only its own slots: no barrier needed)
    float* const fr1 = lds + L::EX0 + kExBuf1 + kTileM * YL + 2 * kGRow + L::Z + L::C1 + (w * 2 * 64 + lane) * 4;   // + q * 256
    float* const fr4 = lds + L::EX0 + kExBuf1 + kTileM * YL + 2 * kGRow + L::Z + L::C1 + L::FR1 + (w * 4 * 64 + lane) * 4;
    if (COMPACT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 2 && q < p.nq1) *reinterpret_cast<f32x4*>(fr1 + q * 256) = *reinterpret_cast<const f32x4*>(p.w1f + ((w * p.nq1 + q) * 64 + lane) * 4);
            *reinterpret_cast<f32x4*>(fr4 + q * 256) = *reinterpret_cast<const f32x4*>(p.w4f + ((w * 4 + q) * 64 + lane) * 4);
        }
    }
    __syncthreads();
    const float* const tb = tab + 32 * w + 4 * hh;                      // bias of this wave's features: + layer*128 + 8q
    const float* const ts = tab + 3 * kF + 32 * w + 4 * hh;             // shift
    const int c0 = 8 * w + 4 * hh;                                       // first of the 4 classes this lane updates
    const f32x4 b4v = lds4(tab + 6 * kF + c0);
    const bool ys_owner = !COMPACT || c0 < kCsmall;                      // COMPACT rows hold 16 classes: waves 2, 3 own none of them

    const long long n_tiles = (p.n_end - p.n_begin + kTileM - 1) / kTileM;
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;
    const long long NC = p.N * C;
    const int n_steps = p.t_hi - p.t_lo + 1;

    // ---- rounds (MULTI): a workgroup keeps its weight fragments and walks the slots q0 + (blockIdx.x + round * gridDim.x) * g_eff + g.
    // One round when the host launches a workgroup per slot (the time-sliced loops: MULTI = false, the loop below is then
    // straight-line code and costs the hot form no registers); several when a short call -- the
    // reference's per-timestep p_sample (classification/card_classification.py:645-653, :665-670), card_eps_theta -- has more
    // tiles than the CUs hold workgroups: the 128 KB of W2 / W3 fragments are then read once per resident workgroup, not
    // once per tile.
    for (long long first = p.q0 + (long long)blockIdx.x * p.g_eff; first < p.q1; first += (long long)gridDim.x * p.g_eff) {
    // ---- the G tile positions of this workgroup
    float* ex[G]; float* ys[G]; float* red[G]; float* gsl[G]; float* zbuf[G]; float* c1l[G];
    float* exw[G]; const float* exr[G]; float* ysw[G]; const float* ysr[G];
    bool active[G], valid[G], first_seg[G], last_seg[G];
    long long row[G], rc[G], rr[G];
    int th[G], cnt[G], gcur[G];
    f32x16 xf[G];
    float* xfl = nullptr;                                // COMPACT (G = 1): this lane's x_feat values in LDS, + 8 * q
    float yv[G][4], mu[G][4];
    uint32_t zblock[G];
    int n_iter = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* base = lds + g * kTileLds;
        ex[g] = base;                                        // 2 x [32][132] activations
        ys[g] = ex[g] + EX0 + kExBuf1;                       // [32][YL] y_t, chain-major
        red[g] = ex[g];                                      // [4][32][36] lin4 partial tiles: alias of exchange buffer 0
        gsl[g] = ys[g] + kTileM * YL;                        // 2 gamma rows
        zbuf[g] = gsl[g] + 2 * kGRow;                        // [4][32][36] normals of the current draw block
        c1l[g] = zbuf[g] + L::Z + (w * 4 * 64 + lane) * 4;   // this lane's 16 values of the hoisted layer-1 tile: + q * 256 (COMPACT: set in the prologue)
        exw[g] = ex[g] + m * kExLd + 32 * w + 4 * hh;
        exr[g] = ex[g] + m * kExLd + 4 * hh;
        ysw[g] = ys[g] + m * YL + c0;
        ysr[g] = ys[g] + m * YL + 4 * hh;
        const long long q = first + g;
        active[g] = g < p.g_eff && q < p.q1;
        const long long tile = active[g] ? q % n_tiles : 0;
        const int seg = active[g] ? (int)(q / n_tiles) : 0;
        const int s0 = (int)((long long)n_steps * seg / p.n_seg), s1 = (int)((long long)n_steps * (seg + 1) / p.n_seg);
        th[g] = p.t_hi - s0;
        cnt[g] = active[g] ? s1 - s0 : 0;
        first_seg[g] = seg == 0;
        last_seg[g] = seg == p.n_seg - 1;
        n_iter = cnt[g] > n_iter ? cnt[g] : n_iter;
        row[g] = p.n_begin + tile * kTileM + m;
        valid[g] = active[g] && row[g] < p.n_end;
        rc[g] = row[g] < p.n_end ? row[g] : p.n_end - 1;
        rr[g] = p.n_z > 1 ? rc[g] / p.n_z : rc[g];
        gcur[g] = 0;
        zblock[g] = 0xFFFFFFFFu;
    }
    // every wave of the workgroup must walk the same number of barrier rounds
    n_iter = __builtin_amdgcn_readfirstlane(n_iter);

    // Noise.  One Philox call yields the four normals of four consecutive draws of ONE (chain, class) element, so a
    // tile needs 32 C calls per four timesteps.  They are spread over all 256 lanes of the workgroup (element e = chain * C
    // + class on thread e mod 256: 1.25 calls per lane at C = 10) and parked in LDS as [draw & 3][chain][class]; the lane
    // that updates classes c0 .. c0+3 of a chain reads its four normals of a draw with one ds_read_b128.  (Every lane
    // drawing for its own four classes cost 4 calls per lane whatever C: 1 400 cycles per timestep on every wave.)
    const float inv_c = 1.0f / (float)C;
    auto gen_block = [&](int g, uint32_t blk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + kThreads * j;
            if (e < kTileM * C) {
                const int chain = (int)(((float)e + 0.5f) * inv_c), cls = e - chain * C;       // exact for e < 1024
                long long re = row[g] - m + chain;
                re = re < p.n_end ? re : p.n_end - 1;
                float z4[4];
                philox_normal4(p.seed, p.elem_offset + (unsigned long long)(re * C + cls), blk, z4);
                float* zb = zbuf[g] + chain * YL + cls;
#pragma unroll
                for (int k = 0; k < 4; ++k) zb[k * kTileM * YL] = z4[k];
            }
        }
    };
    auto store_state = [&](int g, int s) {
        long long slot;
        if (p.out_mode == CARD_OUT_ALL) slot = s - slot_base;
        else if (p.out_mode == CARD_OUT_SELECT) slot = p.state_slot[s];
        else return;
        if (slot < 0 || !valid[g]) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < C) p.out[slot * NC + row[g] * C + c0 + k] = yv[g][k];
    };

    // ---- per-tile prologue: hoisted static part of layer 1, this wave's x_feat block, the chain state
    const bool rng = !p.noise && !p.eps_only;
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (active[g] && first_seg[g] && !p.y_init && rng) { gen_block(g, 0u); zblock[g] = 0u; }     // draw 0 makes y_T
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!active[g]) continue;
        f32x16 c1;                                           // hoisted static part of layer 1, parked in LDS (each lane reads
                                                             // back only what it wrote itself: no barrier)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = lds4(tb + 8 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c1[4 * q + k] = b[k];
                xf[g][4 * q + k] = p.xfeat[rr[g] * kF + 32 * w + 8 * q + 4 * hh + k];     // (COMPACT: parked in LDS below, dead after that)
            }
        }
        if (p.guidance) {
            for (int q = 0; q < p.nq1; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.w1f + (((kWaves + w) * p.nq1 + q) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 8 * q + 4 * hh + e;
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], c < C ? p.y0hat[rr[g] * C + c] : 0.0f, c1, 0, 0, 0);
                }
            }
        }
        if (COMPACT) {
            // the tile's chains come from at most two input rows (n_z >= 32, checked by the host): one column per row;
            // the lanes of a row write the same values to the same place
            long long r0 = row[g] - m;
            r0 = r0 < p.n_end ? r0 : p.n_end - 1;
            const int sel = (int)(rr[g] - r0 / p.n_z);
            c1l[g] = zbuf[g] + L::Z + sel * kC1Ld + 32 * w + 4 * hh;
            xfl = zbuf[g] + L::Z + L::C1 + L::FR1 + L::FR4 + sel * kC1Ld + 32 * w + 4 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(xfl + 8 * q) = f32x4{xf[g][4 * q + 0], xf[g][4 * q + 1], xf[g][4 * q + 2], xf[g][4 * q + 3]};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(c1l[g] + q * kC1q) = f32x4{c1[4 * q + 0], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) mu[g][k] = c0 + k < C ? p.ytmean[rr[g] * C + c0 + k] : 0.0f;
        if (!first_seg[g]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[g][k] = c0 + k < C ? p.ybuf[(rc[g] - p.n_begin) * C + c0 + k] : 0.0f;
        } else if (p.y_init) {
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[g][k] = c0 + k < C ? p.y_init[rc[g] * C + c0 + k] : 0.0f;
        } else {
            f32x4 z;
            if (p.noise) {
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = c0 + k < C ? p.noise[rc[g] * C + c0 + k] : 0.0f;
            } else {
                z = lds4(zbuf[g] + m * YL + c0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[g][k] = c0 + k < C ? z[k] + mu[g][k] : 0.0f;       // classification/diffusion_utils.py:123-124
            store_state(g, 0);
        }
        if (ys_owner) *reinterpret_cast<f32x4*>(ysw[g]) = f32x4{yv[g][0], yv[g][1], yv[g][2], yv[g][3]};
        for (int i = tid; i < kGRow; i += kThreads) gsl[g][i] = p.gtab[(long long)th[g] * kGRow + i];
    }
    __syncthreads();

    auto publish = [&](int g, const f32x16& a, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(exw[g] + buf * EX0 + 8 * q) = f32x4{a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
    };
    auto hidden = [&](int g, const float (&A)[64], int buf, int layer, const float* gl) -> f32x16 {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = lds4(tb + layer * kF + 8 * q);
            acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 h = lds4(exr[g] + buf * EX0 + 32 * (i >> 2) + 8 * (i & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * i + e], h[e], acc, 0, 0, 0);
        }
        epilogue_bn(acc, gl + layer * kF + 32 * w, ts + layer * kF);
        return acc;
    };

    for (int it = 0; it < n_iter; ++it) {
#ifdef CARD_CLS_PROFILE
        long long pts[9];
#endif
        PROF_T(0);
        bool on[G], more[G];
        f32x4 gn[G];
        float z[G][4];
        f32x4 ca[G], cb[G];
        const float* gl[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            on[g] = it < cnt[g];
            more[g] = it + 1 < cnt[g];
            gl[g] = gsl[g] + gcur[g] * kGRow + 4 * hh;
#pragma unroll
            for (int k = 0; k < 4; ++k) z[g][k] = 0.0f;
            const int t = on[g] ? th[g] - it : th[g];       // (an idle slot requests the rows of a valid timestep)
            if (on[g] && t >= 1 && !p.eps_only) {
                const int d = T - t;
                if (p.noise) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[g][k] = c0 + k < C ? p.noise[(long long)(d - draw0) * NC + rc[g] * C + c0 + k] : 0.0f;
                } else if (((uint32_t)d >> 2) != zblock[g]) {
                    zblock[g] = (uint32_t)d >> 2;
                    gen_block(g, zblock[g]);              // read after the third barrier of this timestep
                }
            }
            // This timestep's coefficient row and the next gamma row are requested here UNCONDITIONALLY (an idle slot and
            // card_eps_theta ask valid stand-ins; every lane asks for the gamma row) and first touched where they are used: behind
            // `if (!p.eps_only)` / `if (more && tid < 96)` the compiler waited for them -- and for every other load in flight --
            // where the branch rejoins, an L2 round trip at the head of every timestep (card_guided.hip,
            // card_sample_coop16_kernel).  Behind the noise block above for the same reason: its branches rejoin with a wait.
            {
                const float* const crow = (p.eps_only ? p.gtab : p.coef) + (long long)t * CARD_COEF_COLS;
                ca[g] = *reinterpret_cast<const f32x4*>(crow);
                cb[g] = *reinterpret_cast<const f32x4*>(crow + 4);
                gn[g] = *reinterpret_cast<const f32x4*>(p.gtab + (long long)(more[g] ? t - 1 : t) * kGRow + (tid < kGRow / 4 ? tid : 0) * 4);
            }
            if (!on[g]) continue;
            // layer 1 (model.py:82-87): c1 + W1[:, 0:C] . y_t, BN, softplus, Hadamard with x_feat
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = lds4(c1l[g] + q * kC1q);
                acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < p.nq1) {
                    const f32x4 yb = lds4(ysr[g] + 8 * q);
                    f32x4 a1 = {A1[4 * q + 0], A1[4 * q + 1], A1[4 * q + 2], A1[4 * q + 3]};
                    if (COMPACT) a1 = lds4(fr1 + (q & 1) * 256);
                    else if (PAIR) a1 = *reinterpret_cast<const f32x4*>(p.w1f + ((w * p.nq1 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)              // k-step (q, e) holds classes 8q + e and 8q + 4 + e: skip all-padding ones
                        if (8 * q + e < C) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], yb[e], acc, 0, 0, 0);
                }
            }
            epilogue_bn(acc, gl[g] + 32 * w, ts);
            if (COMPACT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 xv = lds4(xfl + 8 * q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[4 * q + k] *= xv[k];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= xf[g][r];
            }
            publish(g, acc, 0);
        }
        PROF_T(1);
        __syncthreads();
        PROF_T(2);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!on[g]) continue;
            const f32x16 acc = hidden(g, A2, 0, 1, gl[g]);
            publish(g, acc, 1);
        }
        PROF_T(3);
        __syncthreads();
        PROF_T(4);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!on[g]) continue;
            const f32x16 acc = hidden(g, A3, 1, 2, gl[g]);
            if (more[g] && tid < kGRow / 4) *reinterpret_cast<f32x4*>(gsl[g] + (gcur[g] ^ 1) * kGRow + tid * 4) = gn[g];
            // lin4 (model.py:94): this wave's K = 32 slice of the 32-class x 32-chain tile
            f32x16 part;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = 0.0f;
            if (PAIR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b = COMPACT ? lds4(fr4 + q * 256) : *reinterpret_cast<const f32x4*>(p.w4f + ((w * 4 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) A4[4 * q + e] = b[e];
                }
            }
            if (C <= kCsmall) {
                // Up to 16 classes: v_mfma_f32_16x16x1_4b_f32, four independent 16-class x 16-chain x 1 blocks in half the
                // matrix-pipe time of a 32 x 32 x 2 (32 cycles; tools/ubench/mfma4b.hip).  Block b = lane >> 4 takes, from
                // the same accumulator register, chains 16 (b & 1) .. + 15 and the feature of lane half b >> 1; the
                // result register 4b + i of a lane is class 4 (lane >> 4) + i of chain 16 (b & 1) + (lane & 15).  Blocks b
                // and b + 2 are the two feature halves of one chain group and are added here.
#pragma unroll
                for (int r = 0; r < 16; ++r) part = __builtin_amdgcn_mfma_f32_16x16x1f32(A4[r], acc[r], part, 0, 0, 0);
                float* const rd = red[g] + (w * kTileM + (lane & 15)) * YL + 4 * (lane >> 4);
                *reinterpret_cast<f32x4*>(rd) = f32x4{part[0] + part[8], part[1] + part[9], part[2] + part[10], part[3] + part[11]};
                *reinterpret_cast<f32x4*>(rd + 16 * YL) = f32x4{part[4] + part[12], part[5] + part[13], part[6] + part[14], part[7] + part[15]};
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) part = __builtin_amdgcn_mfma_f32_32x32x2f32(A4[r], acc[r], part, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(red[g] + (w * kTileM + m) * YL + 8 * q + 4 * hh) =
                        f32x4{part[4 * q + 0], part[4 * q + 1], part[4 * q + 2], part[4 * q + 3]};
            }
        }
        PROF_T(5);
        __syncthreads();
        PROF_T(6);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!on[g]) continue;
            gcur[g] ^= more[g] ? 1 : 0;
            if (COMPACT && !ys_owner) continue;            // up to 16 classes: waves 2 and 3 own none (wave-uniform)
            const int t = th[g] - it;
            f32x4 eps = lds4(red[g] + m * YL + c0);
#pragma unroll
            for (int ww = 1; ww < kWaves; ++ww) {
                const f32x4 v = lds4(red[g] + (ww * kTileM + m) * YL + c0);
                eps += v;
            }
            eps += b4v;
            if (p.eps_only) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (valid[g] && c0 + k < C) p.out[row[g] * C + c0 + k] = eps[k];
                continue;
            }
            // posterior update per (chain, class), classification/diffusion_utils.py:80-86 (t >= 1) / :98-100 (t == 0)
            asm volatile("" : "+v"(ca[g]), "+v"(cb[g]));        // first touched here (see the head of the loop)
            const float ci = ca[g][CARD_COEF_INV_SQRT_AB], co = ca[g][CARD_COEF_ONE_M_SQRT_AB], cs = ca[g][CARD_COEF_SOMAB], g0 = ca[g][CARD_COEF_G0];
            const float g1 = cb[g][CARD_COEF_G1 - 4], g2 = cb[g][CARD_COEF_G2 - 4], sg = cb[g][CARD_COEF_SIGMA - 4];
            if (t >= 1 && !p.noise) {
                const f32x4 zz = lds4(zbuf[g] + ((T - t) & 3) * kTileM * YL + m * YL + c0);
#pragma unroll
                for (int k = 0; k < 4; ++k) z[g][k] = zz[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y0r = ci * ((yv[g][k] - co * mu[g][k]) - eps[k] * cs);
                float v = y0r;
                if (t >= 1) v = ((g0 * y0r + g1 * yv[g][k]) + g2 * mu[g][k]) + sg * z[g][k];
                yv[g][k] = c0 + k < C ? v : 0.0f;
            }
            store_state(g, T - t);
            if (ys_owner) *reinterpret_cast<f32x4*>(ysw[g]) = f32x4{yv[g][0], yv[g][1], yv[g][2], yv[g][3]};
        }
        PROF_T(7);
        __syncthreads();
        PROF_T(8);
#ifdef CARD_CLS_PROFILE
        if (blockIdx.x == 0 && lane == 0)
            for (int i = 0; i < 8; ++i) g_cls_prof[w * 8 + i] += pts[i + 1] - pts[i];
#endif
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!valid[g] || p.eps_only) continue;
        if (!last_seg[g]) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < C) p.ybuf[(row[g] - p.n_begin) * C + c0 + k] = yv[g][k];
        } else if (p.out_mode == CARD_OUT_LAST) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < C) p.out[row[g] * C + c0 + k] = yv[g][k];
        }
    }
    if constexpr (!MULTI) break;
    __syncthreads();        // (a workgroup without a timestep in this round has passed no barrier since its prologue)
    }   // rounds
}

// ------------------------------------------------------------------------------------------------
// create-time packing
// ------------------------------------------------------------------------------------------------
// W2 / W3: element e of lane l in block (j, i = 4jj + q) is W[32j + (l&31)][32jj + 8q + 4(l>>5) + e] * ln2
__global__ void cls_pack_hidden_kernel(const float* __restrict__ W, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kF * kF) return;
    const int e = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) & 3, jj = (idx >> 10) & 3, j = idx >> 12;
    out[idx] = W[(32 * j + (l & 31)) * kF + 32 * jj + 8 * q + 4 * (l >> 5) + e] * kLn2;
}

// lin1: part 0 = y_t columns [0, C), part 1 = y_0_hat columns [C, 2C); k-step (q, e) of lane half h takes class 8q + 4h + e
__global__ void cls_pack_first_kernel(const float* __restrict__ W1, int C, int d_in, int nq1, float* __restrict__ out) {
    const int total = 2 * kWaves * nq1 * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8, q = rest % nq1, j = (rest / nq1) % kWaves, part = rest / (nq1 * kWaves);
        const int c = 8 * q + 4 * (l >> 5) + e, col = part * C + c;
        out[idx] = (c < C && col < d_in) ? W1[(32 * j + (l & 31)) * d_in + col] : 0.0f;
    }
}

// lin4: A[class = l&31][k]: k-step (q, e) of lane half h takes feature 32w + 8q + 4h + e (layer 3's accumulator order)
__global__ void cls_pack_head_kernel(const float* __restrict__ W4, int C, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kWaves * 4 * 256) return;
    const int e = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) & 3, j = idx >> 10;
    // C <= kCsmall: the four-block 16 x 16 x 1 form, A[block = l >> 4][class = l & 15] with the feature half of block >> 1 = l >> 5
    const int cls = C <= kCsmall ? (l & 15) : (l & 31);
    out[idx] = cls < C ? W4[cls * kF + 32 * j + 8 * q + 4 * (l >> 5) + e] * kLn2 : 0.0f;
}

__global__ void cls_pack_tabs_kernel(const float* b1, const float* b2, const float* b3, const float* s1, const float* s2,
                                     const float* s3, const float* b4, int C, float* __restrict__ tabs) {
    const int i = threadIdx.x;   // 128 threads
    tabs[i] = b1[i]; tabs[kF + i] = b2[i]; tabs[2 * kF + i] = b3[i];
    tabs[3 * kF + i] = s1[i] * kLog2e; tabs[4 * kF + i] = s2[i] * kLog2e; tabs[5 * kF + i] = s3[i] * kLog2e;
    if (i < kCmax) tabs[6 * kF + i] = i < C ? b4[i] : 0.0f;
}

__global__ void cls_pack_gamma_kernel(const float* g1, const float* g2, const float* g3, const float* s1, const float* s2,
                                      const float* s3, int n_embed, float* __restrict__ gtab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_embed * kGRow) return;
    const int t = idx / kGRow, r = idx % kGRow, l = r / kF, n = r % kF;
    gtab[idx] = (l == 0 ? g1 : (l == 1 ? g2 : g3))[t * kF + n] * (l == 0 ? s1 : (l == 1 ? s2 : s3))[n] * kLog2e;
}

}  // namespace

bool clscoop_supports(const card_model_desc* d, const CardTune& tune) {
    if (tune.cls_off) return false;                     // CARD_AMD_CLSCOOP=off: the GEMM back end instead (tests, A/B)
    return d->variant == CARD_VARIANT_BN_HADAMARD && d->hidden == kF && d->x_dim == kF && d->y_dim <= kCmax &&
           d->bn_scale[0] && d->bn_scale[1] && d->bn_scale[2] && d->bn_shift[0] && d->bn_shift[1] && d->bn_shift[2];
}

int clscoop_create(card_sampler* s, const card_model_desc* d, hipStream_t stream) {
    const int C = d->y_dim, nq1 = (C + 7) / 8, d_in = s->d_in;
    s->nqs = nq1;
    HIP_TRY(hipMalloc(&s->wpk, (size_t)2 * kF * kF * sizeof(float)));
    HIP_TRY(hipMalloc(&s->w1s, (size_t)2 * kWaves * nq1 * 256 * sizeof(float)));
    HIP_TRY(hipMalloc(&s->W4, (size_t)kWaves * 4 * 256 * sizeof(float)));
    HIP_TRY(hipMalloc(&s->bias, (size_t)kTab * sizeof(float)));
    HIP_TRY(hipMalloc(&s->gtab, (size_t)d->n_embed * kGRow * sizeof(float)));
    HIP_TRY(hipMalloc(&s->ws, (size_t)kChunkTiles * kTileM * C * sizeof(float)));      // chain states between time slices
    s->lds_bytes = (size_t)lds_floats(0) * sizeof(float);
    {
        const void* plain[] = {(const void*)card_sample_cls_coop_kernel<0>, (const void*)card_sample_cls_coop_kernel<1>,
                               (const void*)card_sample_cls_coop_kernel<1, true>};
        for (const void* fn : plain) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_floats(0) * sizeof(float))));
        const void* comp[] = {(const void*)card_sample_cls_coop_kernel<2>, (const void*)card_sample_cls_coop_kernel<2, true>};
        for (const void* fn : comp) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_floats(2) * sizeof(float))));
    }
    s->coop_blocks = 1;
    cls_pack_hidden_kernel<<<kF * kF / 256, 256, 0, stream>>>(d->W[1], s->wpk);
    cls_pack_hidden_kernel<<<kF * kF / 256, 256, 0, stream>>>(d->W[2], s->wpk + kF * kF);
    cls_pack_first_kernel<<<8, 256, 0, stream>>>(d->W[0], C, d_in, nq1, s->w1s);
    cls_pack_head_kernel<<<kWaves * 4, 256, 0, stream>>>(d->W4, C, s->W4);
    cls_pack_tabs_kernel<<<1, kF, 0, stream>>>(d->b[0], d->b[1], d->b[2], d->bn_shift[0], d->bn_shift[1], d->bn_shift[2], d->b4, C, s->bias);
    const int ng = d->n_embed * kGRow;
    cls_pack_gamma_kernel<<<(ng + 255) / 256, 256, 0, stream>>>(d->gamma[0], d->gamma[1], d->gamma[2], d->bn_scale[0],
                                                              d->bn_scale[1], d->bn_scale[2], d->n_embed, s->gtab);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

void clscoop_destroy(card_sampler* s) {
    float* bufs[] = {s->wpk, s->w1s, s->W4, s->bias, s->gtab, s->ws};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
}

int clscoop_run(const card_sampler* s, const CardRun& r, hipStream_t stream) {
    ClsCoopParams p;
    memset(&p, 0, sizeof(p));
    p.wpk = s->wpk; p.w1f = s->w1s; p.w4f = s->W4; p.tabs = s->bias; p.gtab = s->gtab; p.coef = r.coef;
    p.xfeat = r.x; p.y0hat = r.y0hat; p.ytmean = r.ytmean; p.y_init = r.y_init; p.noise = r.noise; p.out = r.out;
    p.state_slot = r.state_slot; p.N = r.N; p.seed = r.seed; p.elem_offset = r.elem_offset;
    p.T = r.T; p.t_hi = r.t_hi; p.t_lo = r.t_lo; p.C = s->y_dim; p.nq1 = s->nqs; p.guidance = s->cat_y_pred;
    p.out_mode = r.out_mode; p.eps_only = r.eps_only; p.n_z = r.n_z;
    p.g_eff = 1;
    // hand-over buffer of the time slices: the sampler's own, or the library's static array (card_sampler_run_static)
    p.ybuf = s->ws;
    long long chunk_tiles = kChunkTiles;
    if (r.lib_scratch) {
        p.ybuf = static_cast<float*>(r.workspace);
        const long long fit = (long long)(r.workspace_bytes / ((size_t)kTileM * s->y_dim * sizeof(float)));
        chunk_tiles = fit < chunk_tiles ? fit : chunk_tiles;
    }
    const long long P = (long long)s->num_cu * kPerCu;          // slots one launch runs at a time: two one-tile workgroups per CU
    const int n_steps = r.t_hi - r.t_lo + 1;
    const int C = s->y_dim;
    // compact LDS layout with the lin1 / lin4 fragments in LDS: up to 16 classes and tiles that span at most two input rows
    const bool compact = C <= kCsmall && r.n_z >= kTileM;
    if (r.plan) {
        // layer 1: the k-steps (q, e) with a real class 8q + e, two classes each; two F x F layers; lin4 as 16 (four-block
        // MFMA) or 32 class columns; the hoisted y_0_hat part of layer 1 once per tile and call
        int k1 = 0;
        for (int q = 0; q < s->nqs; ++q)
            for (int e = 0; e < 4; ++e) k1 += 8 * q + e < C ? 2 : 0;
        r.plan->executed = 2.0 * kF * (k1 + 2 * kF + (C <= kCsmall ? kCsmall : kCmax)) + 2.0 * kF * 8 * s->nqs / (double)n_steps;
    }
    // slots [q0, q1): at most num_cu of them as one-tile workgroups with every fragment in registers (one per CU); more as
    // the PAIR forms, two workgroups per CU, walking the slots beyond what the chip holds at once in rounds
    auto launch = [&]() -> int {
        const long long n = p.q1 - p.q0;
        if (n <= s->num_cu) {
            CARD_LAUNCH(r, "card_sample_cls_coop_kernel<ONE>", true,
                        card_sample_cls_coop_kernel<0><<<dim3((unsigned)n), dim3(kThreads), lds_floats(0) * sizeof(float), stream>>>(p));
            return CARD_OK;
        }
        const long long pn = n < P ? n : P;
        const dim3 pgrid((unsigned)pn);
        const bool multi = n > pn;
        if (compact && multi)
            CARD_LAUNCH(r, "card_sample_cls_coop_kernel<PAIR compact, rounds>", true,
                        card_sample_cls_coop_kernel<2, true><<<pgrid, dim3(kThreads), lds_floats(2) * sizeof(float), stream>>>(p));
        else if (compact)
            CARD_LAUNCH(r, "card_sample_cls_coop_kernel<PAIR compact>", true,
                        card_sample_cls_coop_kernel<2><<<pgrid, dim3(kThreads), lds_floats(2) * sizeof(float), stream>>>(p));
        else if (multi)
            CARD_LAUNCH(r, "card_sample_cls_coop_kernel<PAIR, rounds>", true,
                        card_sample_cls_coop_kernel<1, true><<<pgrid, dim3(kThreads), lds_floats(1) * sizeof(float), stream>>>(p));
        else
            CARD_LAUNCH(r, "card_sample_cls_coop_kernel<PAIR>", true,
                        card_sample_cls_coop_kernel<1><<<pgrid, dim3(kThreads), lds_floats(1) * sizeof(float), stream>>>(p));
        return CARD_OK;
    };
    const long long tiles_total = (r.N + kTileM - 1) / kTileM;
    for (long long t0 = 0; t0 < tiles_total; t0 += chunk_tiles) {
        const long long nt = tiles_total - t0 < chunk_tiles ? tiles_total - t0 : chunk_tiles;
        p.n_begin = t0 * kTileM;
        p.n_end = (t0 + nt) * kTileM < r.N ? (t0 + nt) * kTileM : r.N;
        // segments: as many as make the last, partly filled slice cheap, at >= 32 timesteps each; one segment (whole
        // loops in one launch) when slicing is off (CARD_AMD_CLSCOOP=noslice) or useless
        int n_seg = 1;
        if (!s->tune.cls_noslice && !r.eps_only && nt > P && n_steps >= 64 && chunk_tiles >= P) {
            double best = 1e300;
            for (int ns = 1; ns <= 16 && n_steps / ns >= 32; ++ns) {
                const long long slices = (nt * ns + P - 1) / P;
                const double cost = (double)slices * (double)((n_steps + ns - 1) / ns) + 4.0 * (double)slices;   // + launch overhead in timestep units
                if (cost < best) { best = cost; n_seg = ns; }
            }
        }
        p.n_seg = n_seg;
        const long long slots = nt * n_seg;
        // a slice must not hold two segments of one tile: slices of at most nt slots (P <= nt whenever n_seg > 1).
        // Un-sliced: ONE launch whose workgroups walk all tiles (short calls such as the reference's per-timestep p_sample
        // are launch- and prologue-bound, so fewer launches win).
        const long long per = n_seg > 1 ? P : slots;
        for (long long q0 = 0; q0 < slots; q0 += per) {
            p.q0 = q0;
            p.q1 = q0 + per < slots ? q0 + per : slots;
            if (int rc = launch()) return rc;
        }
    }
    return CARD_OK;
}

#ifdef CARD_CLS_PROFILE
// profile build only (tools/prof_cls_phases.sh): cycles per phase of workgroup 0, [wave][phase]
extern "C" CARD_API int card_debug_cls_prof(long long* host64, int reset) {
    if (host64) HIP_TRY(hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_cls_prof), sizeof(long long) * 64));
    if (reset) { long long z[64] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_cls_prof), z, sizeof(z))); }
    return CARD_OK;
}
#endif
