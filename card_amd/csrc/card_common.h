// card_common.h - shared device helpers of libcard_hip.so (gfx950).
#ifndef CARD_COMMON_H
#define CARD_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>

#include "card_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace card {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kSpThresh = 20.0f * 1.4426950408889634f;  // softplus threshold 20 in log2 units

// ------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10 (Salmon et al. 2011) + Box-Muller.  Restated in oracle/card_oracle.py:philox_normal.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f; }

// four standard normals for (element, draw block).  Box-Muller on the hardware transcendentals: v_log_f32 is log2,
// v_sin_f32 / v_cos_f32 take revolutions, so sqrt(-2 ln u1) = v_sqrt(v_log(u1) * (-2 ln 2)) and the angle 2 pi u2 is u2
// itself.  Exhaustively over the 2^23 uniforms u01 can return (tools/ubench/boxmuller.hip, profiles/r03_ubench.txt): the
// radius within 4.9e-7, sine and cosine within 1.26e-7 of double precision, a normal within 1.5e-6 (the parity test
// allows 2e-6 against the float64 restatement); 17 instructions per pair where logf / sqrtf / sincospif took ~70.
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned long long elem, uint32_t block,
                                               float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), block, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u01(r[2 * p])) * -1.3862943611198906f);
        const float u2 = u01(r[2 * p + 1]);
        z[2 * p] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * p + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
}

__device__ __forceinline__ float pick4(const float (&z)[4], int i) {
    return i == 0 ? z[0] : (i == 1 ? z[1] : (i == 2 ? z[2] : z[3]));
}

// softplus(x)/ln2 with x given in log2 units: F.softplus(beta=1, threshold=20) (regression/model.py:48-50)
__device__ __forceinline__ float softplus_l2(float xs) {
    const float e = __builtin_amdgcn_exp2f(xs);
    const float l = __builtin_amdgcn_logf(1.0f + e);
    // xs > threshold ? xs : l as ONE v_med3_f32: l >= xs always (l = +inf once exp2 overflows), and l <= threshold
    // (to an ulp) whenever xs <= threshold, so the median of (l, xs, threshold) is l below the threshold and xs above
    return __builtin_amdgcn_fmed3f(l, xs, kSpThresh);
}

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// One row of the (T, 8) coefficient table, requested at the TOP of a timestep: asked for where it is used (after lin4) the
// load sat in the dependent chain of every timestep with its full L2 latency -- twice, the second half of the row behind
// the t >= 1 branch (ISA of round 2's kernels).
struct CoefRow { float inv_sab, om_sab, somab, g0, g1, g2, sigma; };
__device__ __forceinline__ CoefRow load_coef(const float* coef, int t) {
    const float* c = coef + (long long)t * CARD_COEF_COLS;
    const f32x4 a = *reinterpret_cast<const f32x4*>(c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(c + 4);
    return {a[CARD_COEF_INV_SQRT_AB], a[CARD_COEF_ONE_M_SQRT_AB], a[CARD_COEF_SOMAB], a[CARD_COEF_G0],
            b[CARD_COEF_G1 - 4], b[CARD_COEF_G2 - 4], b[CARD_COEF_SIGMA - 4]};
}
// posterior update, regression/diffusion_utils.py:83-89 (t >= 1) / :101-103 (t == 0), unfused fp32 in the reference's order
__device__ __forceinline__ float posterior_update(const CoefRow& c, int t, float y, float mu, float eps, float z) {
    const float y0r = c.inv_sab * ((y - c.om_sab * mu) - eps * c.somab);
    if (t < 1) return y0r;
    const float yh = (c.g0 * y0r + c.g1 * y) + c.g2 * mu;
    return yh + c.sigma * z;
}


}  // namespace card

// host-side error plumbing (defined in card_api.hip)
int card_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return card_fail(CARD_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#endif
