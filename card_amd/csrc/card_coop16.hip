// card_coop16.hip - the 16-chain-tile cooperative kernel of the hidden-128 regression sampler (small batches: BASELINE
// configs[0], 5 100 chains).  A translation unit of its own because it is compiled with -falign-loops=64: with the head of
// its timestep loop on an instruction-fetch line of its own a Boston loop takes 3.640 instead of 3.663 ms, while the same
// flag costs the persistent kernel 0.2 % (profiles/r03_ubench.txt).  Dispatch, packing and the other forms: card_guided.hip.
#include "card_guided.h"

using namespace card;

namespace {

// ------------------------------------------------------------------------------------------------
// Fewer than two 32-chain tiles per CU: 16-chain tiles (card_sample_coop16_kernel).
// A tile-timestep of the kernels above is a dependent chain of 130 v_mfma_f32_32x32x2_f32 per wave whatever the batch,
// so 5 100 chains (BASELINE configs[0], 160 tiles) leave 96 CUs idle and take as long as 8 192.  This kernel is
// card_sample_coop2_kernel (two cooperative halves per workgroup, time-sliced between one and two tiles per CU) on
// 16-chain tiles with v_mfma_f32_16x16x4_f32 (D[feature 4g + r][chain n] on lane 16g + n): half the MFMA time per
// tile-timestep and twice as many tiles to spread.
// It is BIT-IDENTICAL to the 32-chain kernels: on gfx950 both fp32 MFMA shapes are plain sequential fp32 FMA chains over
// k in index order (tools/ubench/mfma_order.hip: 0 mismatches against a host fmaf chain), so it is enough to feed the
// input features of every layer in the same order.  The 32-chain kernels consume (f, f + 4) per MFMA for
// f = 8i + e, i = 0..15, e = 0..3; here MFMA s (k = lane group g) takes feature 8(s>>1) + 2(s&1) + (g>>1) + 4(g&1).
// The activation exchange tile is stored in that order (float 64(g&1) + 8a + 4(g>>1) + j for MFMA s = 4a + j), so a lane still fetches
// the B operands of four MFMAs with one ds_read_b128; lin4's partial sums follow the other kernels' association
// (two 16-feature chains per wave handed between lane groups with __shfl_xor, then added).
// ------------------------------------------------------------------------------------------------

// gamma multiply + softplus of this wave's 32 features x 16 chains (two 16-feature accumulator blocks), staged.  The
// gamma values arrive in registers: the caller reads them from LDS BEFORE the layer's MFMAs, so their LDS latency is
// covered by the matrix work instead of being waited out at the head of every epilogue (a lone wave per SIMD hides nothing)
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[2], const f32x4 (&gm)[2]) {
    float xs[8], e[8];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[4 * mb + k] = acc[mb][k] * gm[mb][k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = __builtin_amdgcn_exp2f(xs[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = 1.0f + e[r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = __builtin_amdgcn_logf(e[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r >> 2][r & 3] = __builtin_amdgcn_fmed3f(e[r], xs[r], kSpThresh);
}

#ifdef CARD_COOP16_PROFILE
__device__ long long g_c16_prof[64];
#define PROF16(i) do { pts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PROF16(i) do { } while (0)
#endif

__global__ __launch_bounds__(kCoop2Threads, 2) void card_sample_coop16_kernel(const RunParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w = wave & 3;
    const int htid = tid & (kCoopThreads - 1);
    float* const ex = lds + half * kHalf16LdsFloats;               // 2 exchange tiles [16][132], permuted feature order
    float* const red = ex + 2 * kT16 * kExLd;                      // [4][16] lin4 partial sums
    float* const gsl = red + kCoopWaves * kT16;                    // 2 gamma rows
    float* const zb = gsl + 2 * kGammaRow;                         // the tile's normals for 64 draws
    float* const tab = lds + 2 * kHalf16LdsFloats;
    const int n = lane & 15, g = lane >> 4;
    for (int i = tid; i < kTabFloats; i += kCoop2Threads) tab[i] = p.wpk[kOffB + i];

    // ---- which tile this half advances in this slice (same scheme as card_sample_coop2_kernel)
    const long long n_tiles = (p.n_end - p.n_begin + kT16 - 1) / kT16;
    const int k = p.sl_k, nsg = p.sl_nsingle;
    const long long first_single = (long long)k * nsg;
    long long n_s = n_tiles - first_single;
    n_s = n_s < 0 ? 0 : (n_s > nsg ? nsg : n_s);                   // nsg = 0: every tile shares a CU
    long long tile = -1;
    int quota = 0;
    if ((long long)blockIdx.x < n_s) {
        if (half == 0) { tile = first_single + blockIdx.x; quota = p.sl_L; }
    } else {
        const long long j = 2 * ((long long)blockIdx.x - n_s) + half;
        if (j < n_tiles - n_s) { tile = j < first_single ? j : j + n_s; quota = p.sl_S; }
    }
    const int n_iter = (long long)blockIdx.x < n_s ? p.sl_L : p.sl_S;
    const bool active = tile >= 0;
    const int before = k * p.sl_S + ((active && nsg > 0 && tile / nsg < k) ? p.sl_L - p.sl_S : 0);
    const int th = p.t_hi - before;
    const bool first_seg = before == 0, last_seg = th - quota + 1 <= p.t_lo;

    // this wave's A fragments: A[mb][s] = W[32w + 16mb + (lane & 15)][feature(s, lane >> 4)]
    float A2[64], A3[64];
    if (active) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {                   // i = 8 mb + a: MFMAs s = 4a .. 4a + 3 of block mb
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.wpk16 + ((w * 16 + i) * 64 + lane) * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.wpk16 + kH * kH + ((w * 16 + i) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { A2[4 * i + e] = a[e]; A3[4 * i + e] = b[e]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) { A2[i] = 0.0f; A3[i] = 0.0f; }
    }
    __syncthreads();
    const float* const tb = tab + 32 * w + 4 * g;                                  // bias: + layer * kH + 16 mb
    const float* const w4l = tab + (kOffW4 - kOffB) + 32 * w + 4 * g;              // + 16 mb
    const float b4 = tab[kOffB4 - kOffB];
    // column 0 of W1 for this lane's eight features 32w + 16mb + 4g + r: layer 1 = c1 + W1[:, 0] * y_t is one fmaf per
    // accumulator register here.  The other kernels issue it as a K = 2 MFMA whose only non-zero product is the same
    // (w, y): the MFMA is a sequential fmaf chain over k (tools/ubench/mfma_order.hip), fmaf(0, 0, x) = x, so the bits are
    // the same (up to the sign of a zero, which exp2 does not see) -- and a lone wave does not wait out an MFMA's latency
    // between the update and the first epilogue
    f32x4 w1c[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) w1c[mb] = lds4(tab + (kOffW1 - kOffB) + 32 * w + 16 * mb + 4 * g);
    // loop-invariant LDS tables this lane needs every timestep, held in registers (LDS reads are not hoisted over the
    // loop's barriers by the compiler, and a read placed next to its use costs a full LDS latency in the dependent chain)
    f32x4 w4r[2], bias2[2], bias3[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        w4r[mb] = lds4(w4l + 16 * mb);
        bias2[mb] = lds4(tb + kH + 16 * mb);
        bias3[mb] = lds4(tb + 2 * kH + 16 * mb);
    }
    // Exchange tile, row n = chain: the four B operands (j = 0..3) that lane group gc consumes for MFMAs 4a .. 4a + 3 sit
    // at float 64 (gc & 1) + 8a + 4 (gc >> 1) + j.  A ds_read_b128 is served in 16-lane groups that hold every chain n
    // once, from lane groups {0, 1} or {2, 3}: the two groups of a pair are 64 floats = one whole bank row apart, so the
    // 16-byte slot is n + 2a + (gc >> 1) mod 16 -- conflict-free (the first layout, 16a + 4gc + j, put lanes (gc, n) and
    // (gc + 1, n - 1) on one slot: 27 % of the LDS-array cycles were conflict cycles, profiles/r02_pmc_small_batch.jsonl).
    // publish: accumulator register r of block mb is feature 32w + 16mb + 4g + r = the operand of MFMA s = 4a + j for
    // lane group gc with a = 2w + mb, gc = (g & 1) + 2 (r >> 1) ... after lanes l and l + 32 (groups g and g + 2) have
    // traded halves (v_permlane32_swap) each lane holds j = 0..3 of ONE consumer group gc = g and writes one aligned
    // b128 (8 consecutive lanes = 8 consecutive chains at one column: conflict-free)
    float* const exw = ex + n * kExLd + 64 * (g & 1) + 4 * (g >> 1) + 16 * w;      // + 8 mb
    const float* const exr = ex + n * kExLd + 64 * (g & 1) + 4 * (g >> 1);         // + 8a: B operands of MFMAs 4a .. 4a+3
    const int T = p.T;
    const int draw0 = p.y_init ? (T - p.t_hi) : 0;
    const int slot_base = draw0;

    const long long row = p.n_begin + (active ? tile : 0) * kT16 + n;
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
    // hoisted layer 1: c1 = b1 + W1[:, 1:] . (y_0_hat, x); MFMA s takes static feature 1 + 4s + g (the 32-chain
    // kernels' order 1, 2, 3, ...)
    f32x4 c1[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) c1[mb] = lds4(tb + 16 * mb);
    if (active) {
        for (int q = 0; q < p.nqs; ++q) {                  // 8 static features per q: two MFMAs per block
            float bq[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) bq[e] = static_feature(1 + 8 * q + 4 * e + g);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a = p.w1s16[((long long)((w * 2 + mb) * p.nqs + q) * 2 + e) * 64 + lane];
                    c1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[e], c1[mb], 0, 0, 0);
                }
            }
        }
    }
    // the 16 (wave, lane group) copies of a chain draw 16 different blocks: one generator call per lane covers 64
    // timesteps of the tile (see kZFloats above)
    int zgrp = -1;
    auto zfill = [&](int d) {                    // uniform over the half
        if (__builtin_expect(p.noise || (d >> 6) == zgrp, 1)) return;       // (the refill, once per 64 timesteps, out of line)
        zgrp = d >> 6;
        float zz[4];
        philox_normal4(p.seed, elem, (uint32_t)(16 * zgrp + 4 * w + g), zz);
#pragma unroll
        for (int j = 0; j < 4; ++j) zb[((4 * w + g) * 4 + j) * kT16 + n] = zz[j];
    };
    auto draw_lds = [&](int d) -> float { return zb[(((d >> 2) & 15) * 4 + (d & 3)) * kT16 + n]; };
    // (a select of VALUES: `p.noise ? p.noise[..] : zb[..]` as one load compiles to a select of the two ADDRESSES and a flat_load)
    auto draw = [&](int d) -> float {
        float zl = draw_lds(d);
        asm volatile("" : "+v"(zl));
        float zg = 0.0f;
        if (p.noise) zg = p.noise[(long long)(d - draw0) * p.N + rc];
        return p.noise ? zg : zl;
    };
    // A lone wave per SIMD pays for every TAKEN branch of its timestep (nothing else to issue while the instruction buffer
    // refills), so the hot path -- RNG noise, last state only, or a wave that stores nothing -- falls through: the rarer
    // cases sit out of line behind __builtin_expect.
    const bool st_wave = w == 0 && p.out_mode != CARD_OUT_LAST && !p.eps_only;      // wave-uniform: this wave stores states
    auto store_state = [&](int s, float v, int selected) {
        if (__builtin_expect(st_wave, 0)) {
            const long long slot = p.out_mode == CARD_OUT_ALL ? s - slot_base : selected;
            if (slot >= 0 && valid && g == 0) p.out[slot * p.N + row] = v;
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
    // layer 1's gamma values of the NEXT timestep are read right behind the barrier that publishes their row (the third
    // of a timestep), ahead of the update chain, instead of at the head of the next timestep
    f32x4 gm1[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) gm1[mb] = lds4(gsl + 4 * g + 32 * w + 16 * mb);

    auto publish = [&](const f32x4 (&a)[2], int buf) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            // lower lanes end up with (own r0, own r2, partner r0, partner r2), upper lanes with (partner r1, partner r3,
            // own r1, own r3): v_permlane32_swap exchanges the first operand's lanes 32-63 with the second's lanes 0-31.
            // (Inline asm: ROCm 7.2's __builtin_amdgcn_permlane32_swap loses its second result when both results feed a
            // vector build -- tools/ubench/permlane.hip; the s_nop cover the VALU-write -> permlane read distance.)
            float x0 = a[mb][0], x1 = a[mb][1], y0 = a[mb][2], y1 = a[mb][3];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                         : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
            const f32x4 v = {x0, y0, x1, y1};
            *reinterpret_cast<f32x4*>(exw + buf * kT16 * kExLd + 8 * mb) = v;
        }
    };
    auto hidden = [&](const float (&A)[64], int buf, int layer, const float* gl, const f32x4 (&bias)[2], f32x4 (&acc)[2]) {
        f32x4 gm[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            acc[mb] = bias[mb];
            gm[mb] = lds4(gl + layer * kH + 32 * w + 16 * mb);       // used after the MFMAs
        }
        // The B operands of a k-group (one ds_read_b128 feeds 8 MFMAs) are read TWO groups ahead: left to itself the
        // compiler issues each pair of reads right where it waits for them, and a lone wave sits out the LDS latency behind
        // every sixteenth MFMA (~70 cycles, four times a layer).  sched_barrier keeps the read where it is written.
        const float* const hb = exr + buf * kT16 * kExLd;
        f32x4 hq[3];
        hq[0] = lds4(hb);
        hq[1] = lds4(hb + 8);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (a + 2 < 8) hq[(a + 2) % 3] = lds4(hb + 8 * (a + 2));
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 h = hq[a % 3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * a + e], h[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[32 + 4 * a + e], h[e], acc[1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        epilogue16(acc, gm);
    };

    // Every load issued so far is waited for HERE: a value still in flight at the head of the timestep loop is waited for
    // inside it, timestep after timestep (vmcnt(N) at each use of a weight register, together with whatever the timestep
    // itself has requested by then).
    float mu_r = mu;
    asm volatile("" : "+v"(y), "+v"(mu_r));
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        asm volatile("" : "+v"(A2[i]), "+v"(A2[i + 1]), "+v"(A2[i + 2]), "+v"(A2[i + 3]), "+v"(A2[i + 4]), "+v"(A2[i + 5]), "+v"(A2[i + 6]), "+v"(A2[i + 7]));
        asm volatile("" : "+v"(A3[i]), "+v"(A3[i + 1]), "+v"(A3[i + 2]), "+v"(A3[i + 3]), "+v"(A3[i + 4]), "+v"(A3[i + 5]), "+v"(A3[i + 6]), "+v"(A3[i + 7]));
    }

    // injected noise, or a stand-in address (see the head of the loop): one address form, so no branch around its arithmetic
    float eps_out = 0.0f;
    const int grow = lane < kGammaRow / 16 ? (kGammaRow / 16) * w + lane : 0;     // this lane's 16 bytes of a gamma row
    const float* const zin_base = p.noise ? p.noise + rc : p.ytmean + rr;
    const long long zin_stride = p.noise ? p.N : 0;
    for (int it = 0; it < n_iter; ++it) {
#ifdef CARD_COOP16_PROFILE
        long long pts[8];
#endif
        PROF16(0);
        const bool on = active && it < quota;
        const int t = on ? th - it : p.t_lo;          // (an idle half requests the rows of a valid timestep)
        const bool more = on && it + 1 < quota;
        float z = 0.0f;
        f32x4 acc[2];
        const float* const gl = gsl + gcur * kGammaRow + 4 * g;
        // Everything the timestep takes from global memory is requested here, unconditionally (from a stand-in address
        // where the mode has no such input), and first touched where it is used: a load behind a branch is waited for --
        // with every other load in flight -- where the branch rejoins, i.e. the coefficient row cost a lone tile an L2
        // round trip at the head of every timestep.
        const float* const crow = (p.eps_only ? p.gtab : p.coef) + (long long)t * CARD_COEF_COLS;
        f32x4 ca = *reinterpret_cast<const f32x4*>(crow), cb = *reinterpret_cast<const f32x4*>(crow + 4);
        // (every lane loads: behind `if (htid < kGammaRow / 4)` the load is one the compiler cannot count, and its next
        // register-reuse wait falls one short -- on the coefficient row just requested)
        // (a quarter of the row per wave, lanes 0-23: every wave has lanes that store it below, so none of them takes the
        // branch around an empty store)
        const f32x4 gn = *reinterpret_cast<const f32x4*>(p.gtab + (long long)(more ? t - 1 : t) * kGammaRow + grow * 4);
        const bool noisy = t >= 1 && !p.eps_only;
        float zin = zin_base[(long long)(noisy ? T - t - draw0 : 0) * zin_stride];
        int slot_sel = *(p.out_mode == CARD_OUT_SELECT ? p.state_slot + (T - t) : reinterpret_cast<const int*>(p.wpk));
        if (on) {
            if (noisy) zfill(T - t);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mb][r] = fmaf(w1c[mb][r], y, c1[mb][r]);
            epilogue16(acc, gm1);
            publish(acc, 0);
        }
        PROF16(1);
        __syncthreads();
        PROF16(2);
        if (on) {
            z = noisy ? draw_lds(T - t) : 0.0f;          // (RNG mode's normal; injected noise is `zin`)
            hidden(A2, 0, 1, gl, bias2, acc);
            publish(acc, 1);
        }
        PROF16(3);
        __syncthreads();
        PROF16(4);
        if (on) {
            hidden(A3, 1, 2, gl, bias3, acc);
            if (more && lane < kGammaRow / 16) *reinterpret_cast<f32x4*>(gsl + (gcur ^ 1) * kGammaRow + grow * 4) = gn;
            // lin4 in the 32-chain kernels' association: the chain over features 8q + e (q, e ascending) lives on lane
            // groups 0 -> 2 -> 0 -> 2, the one over 8q + 4 + e on 1 -> 3 -> 1 -> 3; each finished chain is added to the other
            // (the hand-over between lanes l and l + 32 is one v_permlane32_swap instead of a ds_bpermute round trip; lane
            // groups whose turn it is not compute on values that the next hand-over overwrites)
            float part = 0.0f;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                if (ph > 0) {
                    float px = part, py = part;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(px), "+v"(py));
                    part = (ph & 1) ? px : py;          // odd: lanes 32-63 receive; even: lanes 0-31 receive
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(acc[ph >> 1][e], w4r[ph >> 1][e], part);
            }
            // part + (part of lane ^ 16): v_permlane16_swap trades the odd 16-lane rows of its first operand with the even
            // rows of its second, so two copies of part become (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3);
            // their sum is what part += __shfl_xor(part, 16) gives (fp32 addition commutes) without its LDS round trip
            {
                float pa = part, pb = part;
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(pa), "+v"(pb));
                part = pa + pb;
            }
            if (g == 2) red[w * kT16 + n] = part;
        }
        PROF16(5);
        __syncthreads();
        PROF16(6);
        if (on) {
            gcur ^= more ? 1 : 0;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) gm1[mb] = lds4(gsl + gcur * kGammaRow + 4 * g + 32 * w + 16 * mb);
            const float eps = (((red[n] + red[kT16 + n]) + red[2 * kT16 + n]) + red[3 * kT16 + n]) + b4;
            // (no branch on eps_only here: such a call is one timestep, its update runs on the stand-in row, is not stored,
            // and eps leaves the loop in eps_out)
            eps_out = eps;
            asm volatile("" : "+v"(ca), "+v"(cb), "+v"(zin), "+v"(slot_sel));      // first touched here (see the head of the loop)
            const CoefRow cf = {ca[CARD_COEF_INV_SQRT_AB], ca[CARD_COEF_ONE_M_SQRT_AB], ca[CARD_COEF_SOMAB], ca[CARD_COEF_G0],
                                cb[CARD_COEF_G1 - 4], cb[CARD_COEF_G2 - 4], cb[CARD_COEF_SIGMA - 4]};
            y = posterior_update(cf, t, y, mu_r, eps, p.noise ? zin : z);
            store_state(T - t, y, slot_sel);
        }
#ifdef CARD_COOP16_PROFILE
        PROF16(7);
        asm volatile("" : "+v"(y));
        if (blockIdx.x == 0 && lane == 0 && half == 0) {
            for (int i = 0; i < 7; ++i) g_c16_prof[w * 8 + i] += pts[i + 1] - pts[i];
            g_c16_prof[w * 8 + 7] += 1;
        }
#endif
    }
    if (valid && g == 0 && w == 0) {
        if (p.eps_only) p.out[row] = eps_out;
        else if (!last_seg) p.ybuf[row - p.n_begin] = y;
        else if (p.out_mode == CARD_OUT_LAST) p.out[row] = y;
    }
}

}  // namespace

int coop16_prepare() {
    HIP_TRY(hipFuncSetAttribute((const void*)card_sample_coop16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kCoop16LdsFloats * sizeof(float))));
    return CARD_OK;
}

void coop16_enqueue(const GuidedParams& p, unsigned grid, hipStream_t stream) {
    card_sample_coop16_kernel<<<dim3(grid), dim3(kCoop2Threads), (size_t)kCoop16LdsFloats * sizeof(float), stream>>>(p);
}

#ifdef CARD_COOP16_PROFILE
// profile build only (tools/prof_c16_phases.py): cycles per phase of workgroup 0's first half, [wave][phase], [wave][7] = timesteps
extern "C" CARD_API int card_debug_c16_prof(long long* host64, int reset) {
    if (host64) HIP_TRY(hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_c16_prof), sizeof(long long) * 64));
    if (reset) { long long z[64] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_c16_prof), z, sizeof(z))); }
    return CARD_OK;
}
#endif
