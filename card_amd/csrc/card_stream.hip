// card_stream.hip - persistent sampler kernel for regression ConditionalGuidedModel networks whose hidden
// layers do not fit in LDS (hidden = 256: 2 x 256 KB of fp32 weights; BASELINE.json configs[4], width-256 leg).
//
// Same idea as card_guided.hip -- a wave owns its chains for all T timesteps, activations stay in registers in
// the transposed (feature x chain) layout so that one layer's MFMA accumulators are the next layer's B operands
// -- with two changes forced by the weight size:
//   * weights are STREAMED (the eight-wave form for full batches; the four-wave forms for small batches take the same
//     fragments straight from L2, see the kernel's comment): the three layers of a timestep are cut into 32 KB slabs
//     (8 k-steps x all output features, already in A-fragment order in global memory, L2-resident) that every wave of
//     the workgroup consumes in lockstep from a three-buffer LDS ring; slab i+2 is fetched global -> registers while slab i
//     is computed, one barrier per slab placed MID-slab so that no wave drains its MFMA queue around it and
//     A fragments are prefetched straight across slab boundaries.  The slab sequence is the same every
//     timestep (layer 1 | layer 2 | layer 3), so the ring simply cycles.
//   * tiles are 16 chains wide (v_mfma_f32_16x16x4_f32: D[feature 4g+r][chain n], lane = 16g + n): the
//     256-feature activation vector of 16 chains is 64 VGPRs, the accumulators another 64, which leaves room
//     for two waves per SIMD.  k-step ks of a hidden layer takes register ks of the previous layer's
//     accumulators as its B operand, i.e. input feature 16(ks>>2) + 4g + (ks&3) from lane group g; that k order
//     is baked into the packed weights.
// Layer 1 runs on MFMA every step from its own slab(s) (K padded to 4*KS1); softplus in log2 units with ln2
// folded into W2, W3, W4 exactly as in card_guided.hip; posterior update unfused fp32 in the reference's order.
#include <type_traits>

#include "card_internal.h"

using namespace card;

namespace {

constexpr int kWavesFull = 8;      // 2 per SIMD, one workgroup per CU
constexpr int kWavesSmall = 4;     // small batches: 1 per SIMD, so that num_cu * 64 chains already put a tile on every SIMD
constexpr int kTile = 16;          // chains per wave tile
constexpr int kSlabKs = 8;         // k-steps (of 4) per slab
constexpr int kGroupsPerWg = 8;    // tile groups (128 chains) one workgroup walks per launch

template <int H>
struct Geo {
    static constexpr int NT = H / 16;               // 16-feature output tiles
    static constexpr int NQ = NT / 4;               // ds_read_b128 per k-step (4 tiles each)
    static constexpr int KSH = H / 4;               // k-steps of a hidden layer
    static constexpr int NSH = KSH / kSlabKs;       // slabs of a hidden layer
    static constexpr int SLAB = kSlabKs * NT * 64;  // floats per slab
    static constexpr int TABS = 4 * H + 4;          // b1 | b2 | b3 | W4*ln2 | b4 (+3 pad)
    static constexpr int GROW = 3 * H;              // one timestep's gamma*log2e
    static constexpr int RING = 3;                   // LDS ring buffers
    static constexpr int LDS_FLOATS = RING * SLAB + TABS + 2 * GROW;
};

}  // namespace

struct StreamParams {
    const float* wimg;    // (NS, SLAB) weight slabs in fragment order
    const float* tabs;    // Geo::TABS floats
    const float* gtab;    // (n_embed, 3, H) gamma * log2e
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
    int x_dim, cat_x, cat_y_pred;
    int out_mode, eps_only, n_z;
    // time-sliced paired form (kWaves = 8, SPLIT = 4): launch sl_k of a call cut into K slices, see the kernel
    float* ybuf;          // (n_end - n_begin) chain states handed from one slice to the next
    int sl_k, sl_nsingle, sl_S, sl_L;
};

namespace {

// The forms of one kernel:
//  * kWaves = 8, a tile per wave (two waves per SIMD): the slab ring described above; every A fragment that enters the CU
//    is used by eight waves.  Batches above num_cu * 64 chains.
//  * kWaves = 4 (one wave per SIMD), DIRECT (round 3): no ring.  Below num_cu * 64 chains a workgroup holds at most four
//    tiles, the ring buys little reuse and costs a lone wave a 32 KB store + a barrier per slab (timing-only builds:
//    19 % of the loop, most of it the ds_write_b128 burst holding up the fragment reads queued behind it).  Every wave
//    requests its own A fragments global -> registers, one buffer load of 1 KB per MFMA group, DD k-steps ahead of the
//    MFMAs that consume them (L2-resident weights, 13-16 TB/s chip-wide: within what the L2s deliver); ring position P
//    of a timestep lives in register set P % DD, timestep after timestep.
//      SPLIT = 1: a tile per wave, four tiles per workgroup, one barrier per timestep (for the gamma row).
//      SPLIT = 4: a tile-timestep is a dependent chain of 2 112 MFMAs for the wave that owns it however small the batch,
//        so below num_cu * 32 chains four waves (on four SIMDs) share ONE 16-chain tile: wave `part` computes output
//        features [64 part, 64 part + 64) of every layer from its quarter of the A fragments, and after each epilogue
//        the waves trade their activations through LDS (same lane to same lane: the register layout is identical in all
//        of them), so every wave again holds the whole 256-feature vector as B operands; lin4 and the update run
//        redundantly in all four.  Per output element the FMA chain over k is unchanged: bit-identical to SPLIT = 1.
//        The form stays within 256 registers, so two such workgroups fit a CU (num_cu * 16 < chains <= num_cu * 32) and
//        cover each other's waits.
//      kWaves = 8, SPLIT = 4 (PAIR): two shared tiles ("halves") in one workgroup, one workgroup per CU, for calls between
//        one and two tiles per CU, which are cut into time slices (see `tile` / `quota` below and stream_run).
//    (Round-3 forms this replaced, per 1000 timesteps at x_dim 13: the four-wave ring 43.9 ms at 16 384 chains -> 33.5;
//    a shared tile on the four-wave ring 16.0 ms at 4 096 chains -> 9.2; two shared tiles on an eight-wave ring 20.7 ms at
//    8 192 chains -> 17.2.)
constexpr int kDirectDepth = 16;

template <int H, int KS1, int kWaves, int SPLIT = 1>
__global__ __launch_bounds__(kWaves * 64, (kWaves == 8 || SPLIT == 4) ? 2 : 1) void card_sample_stream_kernel(const StreamParams p) {
    using G = Geo<H>;
    constexpr int kThreads = kWaves * 64;
    constexpr int NT = G::NT, NQ = G::NQ, NSH = G::NSH, SLAB = G::SLAB;
    static_assert(SPLIT == 1 || SPLIT == 4, "a tile belongs to one wave or is shared by four");
    constexpr bool PAIR = kWaves == 8 && SPLIT == 4;        // two shared tiles ("halves") per workgroup, time-sliced
    constexpr bool DIRECT = kWaves == 4 || PAIR;
    constexpr int HALVES = PAIR ? 2 : 1, kHalfThreads = kThreads / HALVES;
    constexpr bool L1RES = SPLIT == 4 && KS1 <= 4;          // DIRECT: layer-1 fragments stay in registers (else they take their turn in the ring)
    constexpr int DP1 = L1RES ? 0 : KS1;                    // ring positions of layer 1
    constexpr int DPT = DP1 + 2 * G::KSH;                   // ring positions per timestep
    constexpr int DD = SPLIT == 1 ? 4 : DPT % kDirectDepth == 0 ? kDirectDepth : kDirectDepth / 2;   // ring depth in k-steps (a k-step is 16 / SPLIT MFMAs)
    static_assert(!DIRECT || DPT % DD == 0, "a k-step keeps its ring slot from timestep to timestep");
    constexpr int NTW = NT / SPLIT;                       // output tiles (of 16 features) this wave computes
    constexpr int NQW = NQ / SPLIT;                       // its ds_read_b128 per k-step
    constexpr int TPW = kWaves / SPLIT;                   // tiles per workgroup
    constexpr int NS1 = (KS1 + kSlabKs - 1) / kSlabKs;
    constexpr int NS = NS1 + 2 * NSH;                     // slabs per timestep
    constexpr int NST = SLAB / (kThreads * 4);            // b128 words of a slab each thread moves global -> ring
    static_assert(SLAB == NST * kThreads * 4, "a slab is a whole number of b128 per thread");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const wbuf = lds;
    float* const tabs = lds + (DIRECT ? 0 : G::RING * SLAB);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    const int part = wave % SPLIT, tl = wave / SPLIT;    // which share of which of the workgroup's tiles
    const int half = PAIR ? tl : 0, htid = PAIR ? (tid & (kHalfThreads - 1)) : tid;
    float* const gbuf = tabs + G::TABS + half * 2 * G::GROW;                                   // two gamma rows (per half)
    float* const xb = tabs + G::TABS + HALVES * 2 * G::GROW + half * (2 * NT * 256) + lane * 4;   // SPLIT = 4: the tile's two activation exchange blocks, + j * 256
    int xpar = 0;                                        // ... and which of them the next epilogue uses

    // PAIR: which tile this half advances in this launch, for how many timesteps, from which timestep.  The scheme of
    // card_sample_coop2_kernel (card_guided.hip): with n_tiles between one and two per CU the call is cut into K launches;
    // in launch k the tiles [k nsg, (k + 1) nsg) run ALONE in a workgroup for sl_L timesteps (the other half only keeps
    // the barriers company) while all others share workgroups for sl_S timesteps, (K - 1) sl_S + sl_L = timesteps of the
    // call; every tile is single in exactly one launch, every CU busy in every launch; chain states cross launches in ybuf.
    [[maybe_unused]] long long tile = -1;
    [[maybe_unused]] int quota = 0, n_iter = 0;
    int th = p.t_hi;
    [[maybe_unused]] bool first_seg = true, last_seg = true;
    if constexpr (PAIR) {
        const long long n_tiles = (p.n_end - p.n_begin + kTile - 1) / kTile;
        const int k = p.sl_k, nsg = p.sl_nsingle;
        const long long first_single = (long long)k * nsg;
        long long n_s = n_tiles - first_single;
        n_s = n_s < 0 ? 0 : (n_s > nsg ? nsg : n_s);
        if ((long long)blockIdx.x < n_s) {
            if (half == 0) { tile = first_single + blockIdx.x; quota = p.sl_L; }
        } else {
            const long long j = 2 * ((long long)blockIdx.x - n_s) + half;
            if (j < n_tiles - n_s) { tile = j < first_single ? j : j + n_s; quota = p.sl_S; }
        }
        n_iter = (long long)blockIdx.x < n_s ? p.sl_L : p.sl_S;
        const int before = k * p.sl_S + ((tile >= 0 && nsg > 0 && tile / nsg < k) ? p.sl_L - p.sl_S : 0);
        th = p.t_hi - before;
        first_seg = before == 0;
        last_seg = th - quota + 1 <= p.t_lo;
    }
    const bool active = !PAIR || tile >= 0;

    // slabs come through a buffer resource with a scalar offset: no per-lane 64-bit addresses to keep (or spill)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wimg), 0, NS * SLAB * 4, 0x00020000);
    const int voff = tid * 16;
    f32x4 stage[NST];         // the slab after the one being computed, on its way global -> registers -> ring
    // words [i0, i1) of a slab: global -> registers, registers -> ring
    auto slab_fetch = [&](int slab, int i0, int i1) {
        const int so = slab * (SLAB * 4);
#pragma unroll
        for (int i = i0; i < i1; ++i)
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, so + i * kThreads * 16, 0));
    };
    auto stage_to_ring = [&](int buf, int i0, int i1) {
        float* dst = wbuf + buf * SLAB + tid * 4;
#pragma unroll
        for (int i = i0; i < i1; ++i) *reinterpret_cast<f32x4*>(dst + i * kThreads * 4) = stage[i];
    };

    // prologue: tables, gamma row of the first step, slab 0 in ring buffer 0, slab 1 in flight
    for (int i = tid * 4; i < G::TABS; i += kThreads * 4)
        *reinterpret_cast<f32x4*>(tabs + i) = *reinterpret_cast<const f32x4*>(p.tabs + i);
    for (int i = htid; i < G::GROW; i += kHalfThreads) gbuf[i] = p.gtab[(long long)th * G::GROW + i];
    // DIRECT: this wave's A fragments of one k-step, NQW x 1 KB: word q, lane l holds output features 16 (4 (NQW part + q) + e) + (l & 15), e = 0..3
    f32x4 frag[DIRECT ? DD : 1][NQW], frag1[DIRECT && L1RES ? KS1 : 1];
    const int voffd = lane * 16 + part * NQW * 1024;
    auto frag_load = [&](int ks, int q) __attribute__((always_inline)) {      // k-step ks of the packed image (slabs of kSlabKs)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            wrsrc, voffd, ((ks / kSlabKs) * SLAB + ((ks % kSlabKs) * NQ + q) * 256) * 4, 0));
    };
    // k-step (of the packed image) at ring position P
    auto pos_ks = [](int P) __attribute__((always_inline)) { return P < DP1 ? P : NS1 * kSlabKs + (P - DP1); };
    if constexpr (DIRECT) {
        if constexpr (L1RES) {
            static_assert(NQW == 1, "one b128 of A fragments per wave and k-step");
#pragma unroll
            for (int s = 0; s < KS1; ++s) frag1[s] = frag_load(s, 0);
        }
#pragma unroll
        for (int P = 0; P < DD; ++P)
#pragma unroll
            for (int q = 0; q < NQW; ++q) frag[P][q] = frag_load(pos_ks(P), q);
    } else {
        slab_fetch(0, 0, NST);
        stage_to_ring(0, 0, NST);
        slab_fetch(1, 0, NST);
    }
    __syncthreads();

    int cur = 0;      // ring buffer holding the slab being computed
    int nxt = 2;      // index (in the per-timestep cycle of NS slabs) of the slab fetched next
    int gcur = 0;     // gamma buffer of the current step
    // ring form: A fragments of the k-step about to run, always one k-step ahead of the MFMAs (also across slab, layer
    // and timestep boundaries: the slab sequence is one endless stream)
    constexpr int NA = 2, DA = NA - 1;
    // (the one form the spread fill makes spill: two layer-1 slabs' worth of inputs)
    constexpr bool SPREAD = KS1 != 8;
    static_assert(!DIRECT || KS1 % 4 == 0, "direct forms: ring positions per timestep divisible by the ring depth (the host runs KS1 = 2 models on the KS1 = 4 kernel)");
    f32x4 afrag[NA][NQ];
    if constexpr (!DIRECT) {
#pragma unroll
        for (int d = 0; d < DA; ++d)
#pragma unroll
            for (int q = 0; q < NQ; ++q) afrag[d][q] = lds4(wbuf + lane * 4 + (d * NQ + q) * 256);
    }

    const float* const bl = tabs + 4 * g;
    const float* const w4l = tabs + 3 * H + 4 * g;
    const float b4 = tabs[4 * H];
    // lin4 = four partial dot products over 64 features each (the features of one `part`), added as (q0 + q1) + (q2 + q3),
    // in every form: a wave that holds a quarter of the features (SPLIT = 4) computes its q and the four waves trade one
    // float per lane instead of the layer-3 activations, and keeps its W4 slice in registers (16; with all 64 of a whole
    // tile in registers the one-wave-per-tile direct form measured 1 % slower).
    f32x4 w4r[SPLIT == 4 ? NTW : 1];
    if constexpr (SPLIT == 4) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) w4r[j] = lds4(w4l + 16 * (part * NTW + j));
    }
    float* const qx = tabs + G::TABS + HALVES * (2 * G::GROW + 2 * NT * 256) + half * 256 + lane * 4;   // SPLIT = 4: [lane][part] partial dot products
    const long long n_rows = p.n_end - p.n_begin;
    const long long n_groups = (n_rows + TPW * kTile - 1) / (TPW * kTile);
    const int iters = PAIR ? 1 : (int)((n_groups + gridDim.x - 1) / gridDim.x);   // same for every workgroup: barriers inside
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;

    for (int it = 0; it < iters; ++it) {
        const long long group = (long long)it * gridDim.x + blockIdx.x;
        const long long row = p.n_begin + (PAIR ? (active ? tile : 0) : group * TPW + tl) * kTile + n;
        const bool valid = active && row < p.n_end;            // workgroups past the end run on clamped chains
        const long long rc = row < p.n_end ? row : p.n_end - 1;
        const long long rr = p.n_z > 1 ? rc / p.n_z : rc;
        const float mu = p.ytmean[rr];
        const float y0h = p.y0hat[rr];
        const unsigned long long elem = p.elem_offset + (unsigned long long)rc;
        // input column k of cat(y_t, [y_0_hat], [x]) (regression/model.py:38-47); column 0 is y_t
        auto input_col = [&](int k) -> float {
            if (k == 0) return 0.0f;
            if (p.cat_y_pred && k == 1) return y0h;
            const int xi = k - 1 - (p.cat_y_pred ? 1 : 0);
            return (p.cat_x && xi < p.x_dim) ? p.x[rr * p.x_dim + xi] : 0.0f;
        };
        float xin[KS1];                                        // layer-1 B operands: k-step s takes column 4s + g
#pragma unroll
        for (int s = 0; s < KS1; ++s) xin[s] = input_col(4 * s + g);

        float zc[4];
        uint32_t zblock = 0xFFFFFFFFu;
        auto draw_rng = [&](int d) -> float {
            const uint32_t blk = (uint32_t)d >> 2;
            if (blk != zblock) {
                philox_normal4(p.seed, elem, blk, zc);
                zblock = blk;
            }
            return pick4(zc, d & 3);
        };
        auto store_state = [&](int s, float v, int selected) {
            long long slot;
            if (p.out_mode == CARD_OUT_ALL) slot = s - slot_base;
            else if (p.out_mode == CARD_OUT_SELECT) slot = selected;
            else return;
            if (slot >= 0 && valid && g == 0 && part == 0) p.out[slot * p.N + row] = v;
        };
        float y;
        if (PAIR && !first_seg) {
            y = p.ybuf[rc - p.n_begin];
        } else if (p.y_init) {
            y = p.y_init[rc];
        } else {
            y = (p.noise ? p.noise[rc] : draw_rng(0)) + mu;    // regression/diffusion_utils.py:109-110
            store_state(0, y, p.out_mode == CARD_OUT_SELECT ? p.state_slot[0] : 0);
        }

        // (every load so far is waited for HERE: a value still in flight at the head of the timestep loop is waited for
        // inside it, timestep after timestep, together with the weight requests issued before it)
        float mu_r = mu;
        asm volatile("" : "+v"(y), "+v"(mu_r));
#pragma unroll
        for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(xin[s]));
        const int n_it = PAIR ? n_iter : p.t_hi - p.t_lo + 1;
        for (int i = 0; i < n_it; ++i) {
            const int t = th - i;
            if constexpr (PAIR) {
                if (!(active && i < quota)) {          // an idle half: the three barriers of a timestep
                    __syncthreads(); __syncthreads(); __syncthreads();
                    continue;
                }
            }
            // next gamma row: t-1, or t_hi again for the next group of chains
            const int tn = PAIR ? (i + 1 < quota ? t - 1 : t) : (t > p.t_lo ? t - 1 : p.t_hi);
            f32x4 gn = {0.f, 0.f, 0.f, 0.f};
            if (htid < G::GROW / 4) gn = *reinterpret_cast<const f32x4*>(p.gtab + (long long)tn * G::GROW + htid * 4);
            // Everything the tail of the timestep takes from global memory is requested here, unconditionally (from a
            // stand-in address where the mode has no such input): a load behind a branch makes the compiler wait for ALL
            // outstanding loads where the branch rejoins, which drained the weight requests in flight once per timestep.
            const float* const crow = (p.eps_only ? p.gtab : p.coef) + (long long)t * CARD_COEF_COLS;
            f32x4 ca = *reinterpret_cast<const f32x4*>(crow), cb = *reinterpret_cast<const f32x4*>(crow + 4);
            const bool noisy = t >= 1 && !p.eps_only;
            float zin = *(p.noise && noisy ? p.noise + (long long)(T - t - draw0) * p.N + rc : p.ytmean + rr);
            int slot_sel = *(p.out_mode == CARD_OUT_SELECT ? p.state_slot + (T - t) : reinterpret_cast<const int*>(p.tabs));
            const float zrng = (noisy && !p.noise) ? draw_rng(T - t) : 0.0f;
            const float* const gl = gbuf + gcur * G::GROW + 4 * g;

            f32x4 acc[NTW];
            float h[4 * NT];
            [[maybe_unused]] f32x4 gmr[SPLIT == 4 ? NTW : 1];      // SPLIT = 4: the layer's gamma values, read ahead of its MFMAs (a lone wave waits out every LDS read placed next to its use)
            auto load_bias = [&](int layer) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = lds4(bl + layer * H + 16 * (part * NTW + j));
                if constexpr (SPLIT == 4) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) gmr[j] = lds4(gl + layer * H + 16 * (part * NTW + j));
                }
            };
            // One slab = NKS k-steps: acc[j] += A(slab)[j][ks] * b[OFF + ks].  Ring protocol (3 buffers, ONE barrier per
            // slab, placed mid-slab where every wave has MFMA work on both sides of it):
            //   first half: slab q+1 (fetched during slab q-1) registers -> buffer (cur+1)%3, last read during slab q-2,
            //          which every wave had finished before it passed the barrier inside slab q-1; each register set is
            //          refilled from slab q+2 as soon as it is written.  SPREAD: a share of the words behind each
            //          k-step's fragment reads instead of all of them at the top, where the 32 KB a workgroup stores at
            //          once held up the fragment reads queued behind them: 133.8 -> 132.0 ms per 1000 timesteps at 65 536
            //          chains (and -7 % / -18 % on the four-wave ring forms of the time)
            //   mid:   barrier: slab q+1 is now visible to all waves
            //   end:   the last k-step prefetches the first A fragments of slab q+1 -- no bubble at the boundary
            auto slab = [&](auto nks_tag, auto off_tag, const auto& b) {
                constexpr int NKS = decltype(nks_tag)::value, OFF = decltype(off_tag)::value;
                static_assert(NKS % NA == 0, "the fragment ring restarts at every slab");
                // the barrier: mid-slab, and before the first fragment of the NEXT slab is asked for (k-step NKS - DA)
                constexpr int BAR = NKS / 2 < NKS - DA ? NKS / 2 : NKS - DA;
                const int nb = cur == G::RING - 1 ? 0 : cur + 1;
                const float* wl = wbuf + cur * SLAB + lane * 4;
                const float* wn = wbuf + nb * SLAB + lane * 4;
                if constexpr (!SPREAD) {
                    stage_to_ring(nb, 0, NST);
                    slab_fetch(nxt, 0, NST);
                }
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    if (s == BAR) __syncthreads();
                    asm volatile("" ::: "memory");       // keep the scheduler from hoisting a whole slab's LDS reads
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        afrag[(s + DA) % NA][q] = lds4(s + DA < NKS ? wl + ((s + DA) * NQ + q) * 256 : wn + ((s + DA - NKS) * NQ + q) * 256);
                    if (SPREAD && s < BAR) {
                        stage_to_ring(nb, s * NST / BAR, (s + 1) * NST / BAR);
                        slab_fetch(nxt, s * NST / BAR, (s + 1) * NST / BAR);
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s % NA][q][e], b[OFF + s], acc[4 * q + e], 0, 0, 0);
                    }
                }
                nxt = nxt + 1 == NS ? 0 : nxt + 1;
                cur = nb;
            };
            // DIRECT: the k-step at ring position P: its fragments were asked for DD k-steps ago; the register set is refilled
            // at once with the k-step that uses it next
            auto kstep_direct = [&](int P, float bv) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < NQW; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[P % DD][q][e], bv, acc[4 * q + e], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NQW; ++q) frag[P % DD][q] = frag_load(pos_ks((P + DD) % DPT), q);
                __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks the requests to two k-steps ahead of their use)
            };
            // SPLIT = 4, layer 3: softplus and this wave's share of lin4 in one go; the partial sums are traded, not the activations
            auto epilogue_lin4_split = [&]() -> float {
                float d = 0.0f;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const f32x4 gm = gmr[j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) d = fmaf(softplus_l2(acc[j][r] * gm[r]), w4r[j][r], d);
                }
                qx[part] = d;
                __syncthreads();
                const f32x4 q = lds4(qx);
                return (q[0] + q[1]) + (q[2] + q[3]);
            };
            auto epilogue = [&](int layer) {
                if constexpr (SPLIT == 1) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f32x4 gm = lds4(gl + layer * H + 16 * j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[4 * j + r] = softplus_l2(acc[j][r] * gm[r]);
                    }
                } else {
                    // this wave's share to the tile's exchange block, then the whole vector back in register order (own
                    // share included: static register indices).  Two blocks in turn: a block is rewritten two exchanges
                    // later, behind the barrier of the exchange in between, which a wave passes only after its reads below
                    // have returned.
                    float* const xw = xb + xpar * (NT * 256);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const f32x4 gm = gmr[j];
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = softplus_l2(acc[j][r] * gm[r]);
                        *reinterpret_cast<f32x4*>(xw + (part * NTW + j) * 256) = v;
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f32x4 v = lds4(xw + j * 256);
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[4 * j + r] = v[r];
                    }
                    xpar ^= 1;
                }
            };

            // the NSH slabs of one hidden layer; slab S multiplies registers h[8S .. 8S+7]
            auto hidden_slabs = [&](auto first) {
                auto body = [&](auto self, auto S_tag) -> void {
                    constexpr int S = decltype(S_tag)::value;
                    if constexpr (S < NSH) {
                        slab(std::integral_constant<int, kSlabKs>{}, std::integral_constant<int, S * kSlabKs>{}, h);
                        self(self, std::integral_constant<int, S + 1>{});
                    }
                };
                body(body, first);
            };
            // ---- layer 1 (regression/model.py:48)
            if (g == 0) xin[0] = y;
            load_bias(0);
            if constexpr (DIRECT) {
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
                    if constexpr (L1RES) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag1[s][e], xin[s], acc[e], 0, 0, 0);
                    } else {
                        kstep_direct(s, xin[s]);
                    }
                }
                epilogue(0);
                // the next gamma row, published here: the two exchange barriers still ahead in this timestep lie between
                // this store and its readers, and the barrier just passed between the row's last readers and this store
                if (SPLIT > 1 && htid < G::GROW / 4) *reinterpret_cast<f32x4*>(gbuf + (gcur ^ 1) * G::GROW + htid * 4) = gn;
                // ---- layers 2, 3 (:49-50)
#pragma unroll
                for (int layer = 1; layer <= 2; ++layer) {
                    load_bias(layer);
#pragma unroll
                    for (int ks = 0; ks < G::KSH; ++ks) kstep_direct(DP1 + (layer - 1) * G::KSH + ks, h[ks]);
                    if (SPLIT == 1 || layer == 1) epilogue(layer);
                }
            } else {
                if constexpr (NS1 >= 1) slab(std::integral_constant<int, (KS1 < 8 ? KS1 : 8)>{}, std::integral_constant<int, 0>{}, xin);
                if constexpr (NS1 >= 2) slab(std::integral_constant<int, (KS1 - 8 < 8 ? KS1 - 8 : 8)>{}, std::integral_constant<int, 8>{}, xin);
                if constexpr (NS1 >= 3) slab(std::integral_constant<int, (KS1 - 16 < 8 ? KS1 - 16 : 8)>{}, std::integral_constant<int, 16>{}, xin);
                static_assert(NS1 <= 3, "layer 1 is built for at most 3 slabs (d_in <= 96)");
                epilogue(0);
                // ---- layers 2, 3 (:49-50)
#pragma unroll
                for (int layer = 1; layer <= 2; ++layer) {
                    load_bias(layer);
                    hidden_slabs(std::integral_constant<int, 0>{});
                    epilogue(layer);
                }
            }
            // ---- lin4 (:51): per-lane partial over its 64 features (four quarter sums), then across the four lane groups
            float dot;
            if constexpr (SPLIT == 4) {
                dot = epilogue_lin4_split();
            } else {
                float q[4];
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    float d = 0.0f;
#pragma unroll
                    for (int j = pq * (NT / 4); j < (pq + 1) * (NT / 4); ++j) {
                        const f32x4 w = lds4(w4l + 16 * j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) d = fmaf(h[4 * j + r], w[r], d);
                    }
                    q[pq] = d;
                }
                dot = (q[0] + q[1]) + (q[2] + q[3]);
            }
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
            const float eps = dot + b4;

            if (p.eps_only) {
                if (valid && g == 0 && part == 0) p.out[row] = eps;
            } else {
                // (the values requested at the top are first touched here: pinned, or the compiler moves a copy -- and with
                // it the wait for every load in flight -- to the head of the timestep)
                asm volatile("" : "+v"(ca), "+v"(cb), "+v"(zin), "+v"(slot_sel));
                const CoefRow cf = {ca[CARD_COEF_INV_SQRT_AB], ca[CARD_COEF_ONE_M_SQRT_AB], ca[CARD_COEF_SOMAB], ca[CARD_COEF_G0],
                                    cb[CARD_COEF_G1 - 4], cb[CARD_COEF_G2 - 4], cb[CARD_COEF_SIGMA - 4]};
                y = posterior_update(cf, t, y, mu_r, eps, p.noise ? zin : zrng);
                store_state(T - t, y, slot_sel);
            }
            // publish the next gamma row (read after the next step's first slab barrier)
            // (DIRECT with a tile per wave has no barrier of its own in the timestep: one here, behind the store; the
            // row's last readers passed the previous timestep's)
            if (!(DIRECT && SPLIT > 1) && htid < G::GROW / 4) *reinterpret_cast<f32x4*>(gbuf + (gcur ^ 1) * G::GROW + htid * 4) = gn;
            if constexpr (DIRECT && SPLIT == 1) __syncthreads();
            gcur ^= 1;
        }
        if (!p.eps_only && valid && g == 0 && part == 0) {
            if (PAIR && !last_seg) p.ybuf[row - p.n_begin] = y;
            else if (p.out_mode == CARD_OUT_LAST) p.out[row] = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// create-time packing
// ------------------------------------------------------------------------------------------------
// slab q of the per-timestep cycle, element ((s*NQ + jq)*64 + l)*4 + e  =  W[out][in] with
//   out = 16(4jq + e) + (l & 15),  g = l >> 4,  ks = 8*S + s
//   layer 1 (q < NS1):  in = 4ks + g (zero past d_in)
//   layers 2, 3:        in = 16(ks >> 2) + 4g + (ks & 3), scaled by ln2 (inputs are softplus / ln2)
__global__ void stream_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ W3,
                                   int H, int d_in, int ns1, float* __restrict__ out) {
    const int NT = H / 16, NQ = NT / 4, NSH = H / 32, SLAB = kSlabKs * NT * 64;
    const long long total = (long long)(ns1 + 2 * NSH) * SLAB;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx / SLAB), r = (int)(idx % SLAB);
        const int e = r & 3, l = (r >> 2) & 63, sq = r >> 8, jq = sq % NQ, s = sq / NQ;
        const int o = 16 * (4 * jq + e) + (l & 15), g = l >> 4;
        float v;
        if (q < ns1) {
            const int in = 4 * (kSlabKs * q + s) + g;
            v = in < d_in ? W1[(long long)o * d_in + in] : 0.0f;
        } else {
            const int layer = (q - ns1) / NSH, S = (q - ns1) % NSH, ks = kSlabKs * S + s;
            const int in = 16 * (ks >> 2) + 4 * g + (ks & 3);
            v = (layer == 0 ? W2 : W3)[(long long)o * H + in] * kLn2;
        }
        out[idx] = v;
    }
}

__global__ void stream_tabs_kernel(const float* b1, const float* b2, const float* b3, const float* W4, const float* b4,
                                   int H, float* __restrict__ tabs) {
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        tabs[i] = b1[i];
        tabs[H + i] = b2[i];
        tabs[2 * H + i] = b3[i];
        tabs[3 * H + i] = W4[i] * kLn2;
    }
    if (threadIdx.x < 4) tabs[4 * H + threadIdx.x] = threadIdx.x == 0 ? b4[0] : 0.0f;
}

__global__ void stream_gamma_kernel(const float* g1, const float* g2, const float* g3, int n_embed, int H,
                                    float* __restrict__ gtab) {
    const long long total = (long long)n_embed * 3 * H;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(idx % H), l = (int)((idx / H) % 3);
        const long long t = idx / (3LL * H);
        gtab[idx] = (l == 0 ? g1 : (l == 1 ? g2 : g3))[t * H + f] * kLog2e;
    }
}

typedef void (*stream_kernel_t)(const StreamParams);

// smallest built layer-1 depth (in k-steps of 4 input columns) that holds d_in
int stream_ks1_for(int d_in) {
    const int need = (d_in + 3) / 4;
    for (int ks : {2, 4, 8, 16, 24})
        if (need <= ks) return ks;
    return 0;
}

template <int W, int S>
stream_kernel_t stream_kernel_pick(int ks1) {
    // (the packed image zero-fills a layer-1 slab beyond d_in and the kernel feeds zeros for columns beyond d_in, so a
    // model packed for two k-steps also runs, with the same results, on the kernel that walks four)
    constexpr bool direct = W == kWavesSmall || S == 4;
    if (direct && ks1 == 2) ks1 = 4;
    switch (ks1) {
        case 2: if constexpr (!direct) return card_sample_stream_kernel<256, 2, W, S>; else return nullptr;
        case 4: return card_sample_stream_kernel<256, 4, W, S>;
        case 8: return card_sample_stream_kernel<256, 8, W, S>;
        case 16: return card_sample_stream_kernel<256, 16, W, S>;
        case 24: return card_sample_stream_kernel<256, 24, W, S>;
        default: return nullptr;
    }
}

// waves per workgroup (8: slab ring | 4: fragments straight from L2) and, with 4, waves per tile (1 | 4)
stream_kernel_t stream_kernel_for(int H, int ks1, int waves, int split = 1) {
    if (H != 256) return nullptr;
    if (waves == kWavesFull) return split == 1 ? stream_kernel_pick<kWavesFull, 1>(ks1) : stream_kernel_pick<kWavesFull, 4>(ks1);
    return split == 4 ? stream_kernel_pick<kWavesSmall, 4>(ks1) : stream_kernel_pick<kWavesSmall, 1>(ks1);
}

// dynamic LDS of a form: tables + two gamma rows; + the slab ring (8 waves) or the two activation exchange blocks and the
// lin4 partial sums (shared tile)
size_t stream_lds_bytes(int H, int waves, int split) {
    const size_t tabs = (size_t)(4 * H + 4 + 2 * 3 * H) * sizeof(float);
    const size_t xchg = (size_t)(2 * (H / 16) * 256 + 256) * sizeof(float);       // one shared tile's exchange blocks + partial sums
    if (waves == kWavesFull && split == 4) return (size_t)(4 * H + 4) * sizeof(float) + 2 * ((size_t)2 * 3 * H * sizeof(float) + xchg);   // two halves
    if (waves == kWavesFull) return tabs + (size_t)3 * kSlabKs * (H / 16) * 64 * sizeof(float);
    return tabs + (split == 4 ? xchg : 0);
}

// A form of the kernel and what one ROUND of it costs (every resident tile advances one timestep), relative to the
// shared-tile form with one workgroup per CU.  Measured per 1000 timesteps at x_dim 13 (profiles/r03_small_batch_latency_h256.jsonl:
// 66.2 / 33.5 / 17.2 / 9.2 ms; x_dim 50: 71.8 / 36.4 / 18.3 / 10.0): ratios of cycle counts of MFMA-bound loops fed from L2,
// they do not move with the shader clock.
struct StreamForm { int waves, split, wg_per_cu; double round_cost; };
struct StreamSeg { int form; long long chains; };
constexpr double kRound8 = 7.2, kRound4 = 3.65, kRoundS2 = 1.87, kRoundS1 = 1.0;
constexpr double kRhoPair = 1.90;        // timestep of two shared tiles in one paired workgroup / of a tile alone on its CU
constexpr double kLaunchCost = 0.5;      // a launch (~5 us) in the same unit (a shared-tile round is ~9.2 us per timestep)

// Can [`chains`] be run as the time-sliced paired form (between one and two shared tiles per CU)?  K, S, L: its launches,
// the timesteps of a paired / of the single launch of a tile.
bool stream_sliceable(long long chains, int num_cu, int n_steps, int* K, int* S, int* L) {
    const long long tiles = (chains + kTile - 1) / kTile, n_single = 2LL * num_cu - tiles;
    if (tiles <= num_cu || n_single < num_cu / 8 || n_steps < 64) return false;
    *K = (int)((tiles + n_single - 1) / n_single);
    *S = (int)((double)n_steps / ((double)(*K - 1) + kRhoPair) + 0.5);
    *L = n_steps - (*K - 1) * *S;
    return *K <= 8 && *S >= 8 && *L >= *S;
}

// Cheapest cut of N chains into segments: whole rounds of the forms (ordered by falling round capacity cap[f]; forms[f].waves
// == 0: not available) and one rest, which runs as a partial round of any form or, where `slice_ok` and stream_sliceable,
// time-sliced on the paired form (recorded as a segment of form `f_pair`, the two-workgroups-per-CU form, which the run
// loop slices).  Exhaustive over a handful of round counts.
double stream_plan(const StreamForm* forms, int nf, const long long* cap, long long N, int steps, bool slice_ok, int f_pair, int num_cu,
                   StreamSeg* out, int* n_out) {
    double best = 1e300;
    *n_out = 0;
    const int kmax[4] = {0, 2, 1, 2};       // rounds tried of forms 1..3; form 0 (the largest): what fits, or one less
    const long long K0 = N / cap[0];
    int k[4];
    for (k[0] = (int)(K0 > 0 ? K0 - 1 : 0); k[0] <= (int)K0; ++k[0])
        for (k[1] = 0; k[1] <= (nf > 1 ? kmax[1] : 0); ++k[1])
            for (k[2] = 0; k[2] <= (nf > 2 ? kmax[2] : 0); ++k[2])
                for (k[3] = 0; k[3] <= (nf > 3 ? kmax[3] : 0); ++k[3]) {
                    long long used = 0;
                    double cost = 0.0;
                    for (int f = 0; f < nf; ++f) {
                        used += (long long)k[f] * cap[f];
                        if (k[f]) cost += (double)k[f] * forms[f].round_cost * steps + kLaunchCost;
                    }
                    if (used > N) continue;
                    const long long rest = N - used;
                    int rest_form = -1;
                    double rest_cost = 0.0;
                    if (rest > 0) {
                        rest_cost = 1e300;
                        for (int g = 0; g < nf; ++g) {
                            // (equal costs: the rest on the smallest form that holds it, whole rounds on the larger ones)
                            const double c = (double)((rest + cap[g] - 1) / cap[g]) * forms[g].round_cost * steps + kLaunchCost + 1e-9 * (double)cap[g];
                            if (c < rest_cost) { rest_cost = c; rest_form = g; }
                        }
                        int K, S, L;
                        if (slice_ok && f_pair >= 0 && stream_sliceable(rest, num_cu, steps, &K, &S, &L)) {
                            // K launches of L timesteps each at the pace of a lone tile (+ 10 % for the paired workgroups' lockstep)
                            const double c = 1.1 * (double)K * L * kRoundS1 + K * kLaunchCost;
                            if (c < rest_cost) { rest_cost = c; rest_form = f_pair; }
                        }
                    }
                    if (cost + rest_cost < best) {
                        best = cost + rest_cost;
                        int n = 0;
                        for (int f = 0; f < nf; ++f)
                            if (k[f]) out[n++] = {f, (long long)k[f] * cap[f]};
                        if (rest > 0) out[n++] = {rest_form, rest};
                        *n_out = n;
                    }
                }
    return best;
}

}  // namespace

bool stream_supports(const card_model_desc* d, int d_in, const CardTune& tune) {
    if (tune.stream_off) return false;                  // CARD_AMD_STREAM=off: the GEMM back end instead (tests, A/B)
    return d->variant == CARD_VARIANT_GUIDED && d->y_dim == 1 && d->hidden == 256 && stream_ks1_for(d_in) != 0;
}

int stream_create(card_sampler* s, const card_model_desc* d, hipStream_t stream) {
    const int H = d->hidden, d_in = s->d_in;
    const int ks1 = stream_ks1_for(d_in);
    stream_kernel_t kern = stream_kernel_for(H, ks1, kWavesFull);
    if (!kern) return card_fail(CARD_E_UNSUPPORTED, "card_sampler_create: no streamed kernel for hidden=%d d_in=%d", H, d_in);
    const int ns1 = (ks1 + kSlabKs - 1) / kSlabKs, NSH = H / 32, SLAB = kSlabKs * (H / 16) * 64;
    s->k1 = ks1;
    s->kernel = reinterpret_cast<void*>(kern);
    s->lds_bytes = stream_lds_bytes(H, kWavesFull, 1);
    HIP_TRY(hipMalloc(&s->wpk, (size_t)(ns1 + 2 * NSH) * SLAB * sizeof(float)));
    HIP_TRY(hipMalloc(&s->w1s, (size_t)(4 * H + 4) * sizeof(float)));
    HIP_TRY(hipMalloc(&s->gtab, (size_t)d->n_embed * 3 * H * sizeof(float)));
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    // workgroups of the shared-tile form that fit a CU (two: it is built within 256 registers)
    int occ = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)stream_kernel_for(H, ks1, kWavesSmall, 4), kWavesSmall * 64,
                                                         stream_lds_bytes(H, kWavesSmall, 4)));
    s->coop_blocks = occ < 1 ? 1 : occ > 2 ? 2 : occ;
    // the time-sliced paired form: 84 KB of LDS, chain states between its launches (at most two tiles per CU)
    HIP_TRY(hipFuncSetAttribute((const void*)stream_kernel_for(H, ks1, kWavesFull, 4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)stream_lds_bytes(H, kWavesFull, 4)));
    HIP_TRY(hipMalloc(&s->ws, (size_t)2 * s->num_cu * kTile * sizeof(float)));
    stream_pack_kernel<<<512, 256, 0, stream>>>(d->W[0], d->W[1], d->W[2], H, d_in, ns1, s->wpk);
    stream_tabs_kernel<<<1, 256, 0, stream>>>(d->b[0], d->b[1], d->b[2], d->W4, d->b4, H, s->w1s);
    stream_gamma_kernel<<<512, 256, 0, stream>>>(d->gamma[0], d->gamma[1], d->gamma[2], d->n_embed, H, s->gtab);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

void stream_destroy(card_sampler* s) {
    if (s->wpk) (void)hipFree(s->wpk);
    if (s->w1s) (void)hipFree(s->w1s);
    if (s->gtab) (void)hipFree(s->gtab);
    if (s->ws) (void)hipFree(s->ws);
}

int stream_run(const card_sampler* s, const CardRun& r, hipStream_t stream) {
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.wimg = s->wpk; p.tabs = s->w1s; p.gtab = s->gtab; p.coef = r.coef; p.x = r.x; p.y0hat = r.y0hat; p.ytmean = r.ytmean;
    p.y_init = r.y_init; p.noise = r.noise; p.out = r.out; p.state_slot = r.state_slot;
    p.N = r.N; p.seed = r.seed; p.elem_offset = r.elem_offset;
    p.T = r.T; p.t_hi = r.t_hi; p.t_lo = r.t_lo;
    p.x_dim = s->x_dim; p.cat_x = s->cat_x; p.cat_y_pred = s->cat_y_pred;
    p.out_mode = r.out_mode; p.eps_only = r.eps_only; p.n_z = r.n_z;
    // Forms (256 CUs): eight waves per workgroup, a tile each, weights through the LDS slab ring (a round = num_cu * 128
    // chains); four waves (one per SIMD) that take their fragments straight from L2, a tile each (num_cu * 64); ONE tile
    // shared by the four waves of a workgroup, two such workgroups per CU (num_cu * 32) or one (num_cu * 16).  A call is cut
    // into segments of whole rounds (stream_plan above): 20 000 chains are one round of the second form + one of the
    // last, not one round of the first with 39 % of the CUs idle.  Results do not depend on the cut (a chain's arithmetic
    // is the same in every form, its noise keyed by its element index).
    // CARD_AMD_STREAM=big: only the first form; nosplit: never a shared tile (tests, A/B).
    StreamForm forms[4];
    int nf = 0;
    forms[nf++] = {kWavesFull, 1, 1, kRound8};
    if (!s->tune.stream_big) {
        forms[nf++] = {kWavesSmall, 1, 1, kRound4};
        if (!s->tune.stream_nosplit) {
            if (s->coop_blocks >= 2) forms[nf++] = {kWavesSmall, 4, 2, kRoundS2};
            forms[nf++] = {kWavesSmall, 4, 1, kRoundS1};
        }
    }
    long long cap[4];
    for (int f = 0; f < nf; ++f) cap[f] = (long long)s->num_cu * forms[f].wg_per_cu * (forms[f].waves / forms[f].split) * kTile;
    StreamSeg seg[6];
    int nseg = 0, f_pair = -1;
    for (int f = 0; f < nf; ++f)
        if (forms[f].split == 4 && forms[f].wg_per_cu == 2) f_pair = f;
    const int n_steps = r.t_hi - r.t_lo + 1;
    const bool slice_ok = !s->tune.stream_noslice && !p.eps_only;
    stream_plan(forms, nf, cap, p.N, n_steps, slice_ok, f_pair, s->num_cu, seg, &nseg);
    if (r.plan)      // layer 1 on MFMA every step with K padded to 4 * ks1 columns, two H x H layers, lin4 (VALU)
        r.plan->executed = 2.0 * s->hidden * (4 * ((forms[seg[0].form].waves == kWavesSmall && s->k1 == 2) ? 4 : s->k1) + 2 * s->hidden) + 2.0 * s->hidden;
    long long b = 0;
    for (int i = 0; i < nseg; ++i) {
        const StreamForm& f = forms[seg[i].form];
        stream_kernel_t kern = stream_kernel_for(s->hidden, s->k1, f.waves, f.split);
        const size_t lds_bytes = stream_lds_bytes(s->hidden, f.waves, f.split);
        // launches of at most kGroupsPerWg rounds: every workgroup walks the same number of tile groups (the kernel's
        // barriers need that) and one launch stays well under a second
        const long long group = f.waves / f.split * kTile, slots = (long long)s->num_cu * f.wg_per_cu;
        const long long chunk = slots * kGroupsPerWg * group, end = b + seg[i].chains;
        // Between one and two shared tiles per CU (5 000 chains = 313 tiles on 256 CUs): two independent workgroups on 57
        // CUs and one on the others leave those idle for half of the launch.  The call is cut in TIME instead, as
        // card_guided.hip does it for hidden 128: K launches of the paired form (two tiles per 8-wave workgroup, one
        // workgroup per CU), in each of which 2 num_cu - tiles of the tiles run alone for sl_L timesteps and the others in
        // pairs for sl_S, sl_L = kRhoPair sl_S.  Same arithmetic per chain and timestep: bit-identical.
        int K, S, L;
        if (f.split == 4 && f.wg_per_cu == 2 && slice_ok && stream_sliceable(seg[i].chains, s->num_cu, n_steps, &K, &S, &L) &&
            (!r.lib_scratch || (unsigned long long)seg[i].chains * sizeof(float) <= r.workspace_bytes)) {
            const long long tiles = (seg[i].chains + kTile - 1) / kTile, n_single = 2LL * s->num_cu - tiles;
            stream_kernel_t kp = stream_kernel_for(s->hidden, s->k1, kWavesFull, 4);
            p.n_begin = b; p.n_end = end;
            p.ybuf = r.lib_scratch ? static_cast<float*>(r.workspace) : s->ws;
            p.sl_nsingle = (int)n_single; p.sl_S = S; p.sl_L = L;
            for (int k = 0; k < K; ++k) {
                long long n_s = tiles - (long long)k * n_single;
                n_s = n_s > n_single ? n_single : (n_s < 0 ? 0 : n_s);
                const long long grid = n_s + (tiles - n_s + 1) / 2;
                p.sl_k = k;
                CARD_LAUNCH(r, "card_sample_stream_kernel (8 waves, two time-sliced tiles over 4 waves each)", true,
                            kp<<<dim3((unsigned)grid), dim3(kWavesFull * 64), stream_lds_bytes(s->hidden, kWavesFull, 4), stream>>>(p));
            }
            b = end;
            continue;
        }
        for (; b < end; b += chunk) {
            p.n_begin = b;
            p.n_end = b + chunk < end ? b + chunk : end;
            const long long groups = (p.n_end - p.n_begin + group - 1) / group;
            const long long grid = groups < slots ? groups : slots;
            CARD_LAUNCH(r, f.split == 4 ? "card_sample_stream_kernel (4 waves, tile over 4 waves)"
                           : f.waves == kWavesSmall ? "card_sample_stream_kernel (4 waves)" : "card_sample_stream_kernel", true,
                        kern<<<dim3((unsigned)grid), dim3(f.waves * 64), lds_bytes, stream>>>(p));
        }
        b = end;
    }
    return CARD_OK;
}
