// card_guided.h - what the hidden-128 regression kernels share across translation units (card_guided.hip: the persistent
// kernel, the 32-chain cooperative kernels, create / run; card_coop16.hip: the 16-chain-tile cooperative kernel, which is
// built with its timestep loop aligned to an instruction-fetch line).
#pragma once
#include "card_internal.h"

namespace {

constexpr int kH = 128;            // hidden width (regression/model.py:32-35)
constexpr int kWaves = 8;          // waves per workgroup: 2 per SIMD, one workgroup per CU
constexpr int kThreads = kWaves * 64;
constexpr int kTileM = 32;         // chains per wave tile (the MFMA's 32 columns)

// LDS image (floats), copied verbatim from the packed device buffer at kernel start.
constexpr int kOffW2 = 0;                 // 128x128 in fragment order
constexpr int kOffW3 = kH * kH;           // 128x128 in fragment order
constexpr int kOffB = 2 * kH * kH;        // b1 | b2 | b3
constexpr int kOffW4 = kOffB + 3 * kH;    // W4 * ln2
constexpr int kOffB4 = kOffW4 + kH;       // b4 (+3 pad)
constexpr int kOffW1 = kOffB4 + 4;        // column 0 of W1 (the y_t column)
constexpr int kSlotFloats = kTileM * kH;  // one wave's hoisted layer-1 tile (32 chains x 128 features)
constexpr int kImgFloats = kOffW1 + kH;
constexpr int kGammaRow = 3 * kH;         // one timestep's gamma*log2e for the 3 layers

}  // namespace

struct GuidedParams {
    const float* wpk;     // packed LDS image
    const float* gtab;    // (n_embed, 3, 128) gamma * log2e
    const float* coef;    // (T, 8)
    const float* x;
    const float* y0hat;
    const float* ytmean;
    const float* y_init;
    const float* noise;
    float* out;
    const int* state_slot;
    long long N;          // total rows (stride of one output/noise slab)
    long long n_begin;    // this launch covers rows [n_begin, n_end)
    long long n_end;
    unsigned long long seed;
    unsigned long long elem_offset;
    int T, t_hi, t_lo;
    int x_dim, cat_x, cat_y_pred;
    int out_mode;
    int eps_only;         // write eps_theta(y_init, t_hi) and stop
    int n_z;              // chains per input row (x, y_0_hat, y_T_mean are indexed by chain / n_z)
    // hoisted layer 1 (HOIST kernels): static columns of W1 in fragment order + per-wave scratch slots
    const float* w1s;     // (4, nqs, 64, 4): W1[32j + (l&31)][1 + 8q + 2e + (l>>5)]
    float* ws;            // grid * 8 waves * 4096 floats
    int nqs;              // ceil((d_in - 1) / 8)
    // time-sliced cooperative launches (card_sample_coop2_kernel): this launch is slice sl_k of sl_K
    int sl_k, sl_K, sl_nsingle, sl_S, sl_L;
    float* ybuf;          // (n_end - n_begin) chain states handed from one slice to the next
    const float* wpk16;   // card_sample_coop16_kernel: W2 | W3 in 16x16x4 fragment order (x ln2)
    const float* w1s16;   // ... and the static columns of W1: [(w, mb), q, e][64]
};


namespace {
typedef GuidedParams RunParams;

// Wave priorities of the two kinds of phase.  With equal priorities the two waves of a SIMD are issued alternately and
// an epilogue's VALU ops get threaded between the partner's MFMAs, each MFMA <-> VALU switch costing ~14 cycles
// (tools/ubench/interleave); with different priorities one wave runs its block through (measured +2 %).
#ifndef CARD_PRIO_VALU
#define CARD_PRIO_VALU 3
#endif
#ifndef CARD_PRIO_MFMA
#define CARD_PRIO_MFMA 0
#endif
constexpr int kPrioValu = CARD_PRIO_VALU, kPrioMfma = CARD_PRIO_MFMA;

// the cooperative kernels: four waves (one per SIMD) share a tile
constexpr int kCoopWaves = 4;
constexpr int kCoopThreads = kCoopWaves * 64;
constexpr int kExLd = kH + 4;                          // exchange tile row stride (floats): conflict-free b128
constexpr int kTabFloats = kOffW1 + kH - kOffB;        // b1|b2|b3|W4|b4|W1[:,0] as they sit in the packed image
constexpr int kCoop2Threads = 2 * kCoopThreads;
// 16-chain tiles (card_coop16.hip)
constexpr int kT16 = 16;
constexpr int kZ16Floats = 16 * 4 * kT16;              // [block & 15][draw & 3][chain]: 64 draws of the tile's 16 chains
constexpr int kHalf16LdsFloats = 2 * kT16 * kExLd + kCoopWaves * kT16 + 2 * kGammaRow + kZ16Floats;
constexpr int kCoop16LdsFloats = 2 * kHalf16LdsFloats + kTabFloats;
constexpr int kImg16Floats = 2 * kH * kH;                 // W2 | W3 in 16x16x4 fragment order

}  // namespace

// card_coop16.hip: once per process before the first launch (dynamic LDS size), and one launch of card_sample_coop16_kernel
int coop16_prepare();
void coop16_enqueue(const GuidedParams& p, unsigned grid, hipStream_t stream);
