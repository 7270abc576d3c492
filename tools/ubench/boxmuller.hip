// Box-Muller on the hardware transcendentals: accuracy of  sqrt(-2 ln u1)  as v_sqrt_f32(v_log_f32(u1) * (-2 ln 2))  and of
// sin / cos(2 pi u2) as v_sin_f32 / v_cos_f32 (which take revolutions), exhaustively over the 2^23 uniforms the
// generator can produce ((k + 0.5) * 2^-23), against double precision; and the issue cost of a generator call (Philox4x32-10 +
// two Box-Muller pairs) with the library's logf / sqrtf / sincospif against the hardware forms.
// The bound that matters: tests/test_gpu_parity.py holds the draws within 2e-6 (absolute) of the float64 restatement.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "card_hip.h"
#include "../../card_amd/csrc/card_common.h"

__global__ void table(float* rad, float* sn, float* cs) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const float u = ((float)k + 0.5f) * 1.1920928955078125e-07f;
    rad[k] = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);
    sn[k] = __builtin_amdgcn_sinf(u);
    cs[k] = __builtin_amdgcn_cosf(u);
}

template <int FAST>
__global__ void calls(float* out, long long* cyc, int n) {
    float acc = 0.f;
    uint32_t dep = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        float z[4];
        if (FAST) {
            uint32_t r[4];
            card::philox4x32_10(threadIdx.x + dep, 0u, it, 0u, 17u, 4u, r);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(card::u01(r[2 * p])) * -1.3862943611198906f);
                const float u2 = card::u01(r[2 * p + 1]);
                z[2 * p] = rad * __builtin_amdgcn_cosf(u2);
                z[2 * p + 1] = rad * __builtin_amdgcn_sinf(u2);
            }
        } else {
            card::philox_normal4(17u | (4ull << 32), threadIdx.x + dep, it, z);
        }
        acc += (z[0] + z[1]) + (z[2] + z[3]);
        dep = __float_as_uint(acc) & 1u;                       // the next call waits for this one
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

__global__ void addloop(uint32_t* out, long long* cyc, int n) {
    uint32_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, 0x9E3779B9, %0" : "+v"(v[i]));
    const long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    const uint32_t N = 1u << 23;
    float *rad, *sn, *cs;
    hipMalloc(&rad, N * 4); hipMalloc(&sn, N * 4); hipMalloc(&cs, N * 4);
    table<<<N / 256, 256>>>(rad, sn, cs);
    std::vector<float> hr(N), hs(N), hc(N);
    hipMemcpy(hr.data(), rad, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), sn, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), cs, N * 4, hipMemcpyDeviceToHost);
    double er = 0, es = 0, ec = 0, rr = 0;
    uint32_t kr = 0, ks = 0, kc = 0;
    for (uint32_t k = 0; k < N; ++k) {
        const double u = ((double)k + 0.5) * 1.1920928955078125e-07;
        const double r = sqrt(-2.0 * log(u)), s = sin(2.0 * M_PI * u), c = cos(2.0 * M_PI * u);
        const double dr = fabs(hr[k] - r), ds = fabs(hs[k] - s), dc = fabs(hc[k] - c);
        if (dr > er) { er = dr; kr = k; }
        if (dr / r > rr) rr = dr / r;
        if (ds > es) { es = ds; ks = k; }
        if (dc > ec) { ec = dc; kc = k; }
    }
    printf("radius  sqrt(-2 ln u): max abs error %.3e (k = %u, radius %.4f), max relative %.3e\n", er, kr, hr[kr], rr);
    printf("v_sin_f32(u) vs sin(2 pi u): max abs error %.3e (k = %u)\n", es, ks);
    printf("v_cos_f32(u) vs cos(2 pi u): max abs error %.3e (k = %u)\n", ec, kc);
    printf("bound on a normal: radius error + 5.7711 * trig error = %.3e (the test allows 2e-6)\n", er + 5.7711 * (es > ec ? es : ec));
    float* o; long long* c;
    hipMalloc(&o, 64 * 4); hipMalloc(&c, 8);
    const int n = 4000;
    long long a, t0, t1;
    for (int rep = 0; rep < 2; ++rep) {
        addloop<<<1, 64>>>((uint32_t*)o, c, n); hipMemcpy(&a, c, 8, hipMemcpyDeviceToHost);
        calls<0><<<1, 64>>>(o, c, n); hipMemcpy(&t0, c, 8, hipMemcpyDeviceToHost);
        calls<1><<<1, 64>>>(o, c, n); hipMemcpy(&t1, c, 8, hipMemcpyDeviceToHost);
    }
    const double tick = 4.0 * 8 * n / (double)a;
    printf("generator call (4 normals per lane), logf / sqrtf / sincospif: %.0f cycles\n", t0 * tick / n);
    printf("generator call (4 normals per lane), v_log / v_sqrt / v_sin / v_cos: %.0f cycles\n", t1 * tick / n);
    return 0;
}
