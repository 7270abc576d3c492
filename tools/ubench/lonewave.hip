// Micro-benchmark: what does one wave alone on a SIMD lose in the hidden-layer loop of card_guided.hip?
// One "step" = two 128x128 layers on a 32-chain tile (512 MFMAs), in variants that drop one ingredient each.
//   bit 0: A fragments from LDS (else a constant register)     bit 1: softplus epilogue (else plain copy)
//   bit 2: dependent chain per output tile (else 4 interleaved accumulators, one per output tile)
//   bit 3: A fragments pinned one group ahead   bit 4: epilogue staged over the tile   bit 5 / 6: s_setprio around the epilogue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../card_amd/csrc/card_common.h"
using namespace card;

template <int V>
__global__ __launch_bounds__(512, 2) void k(int steps, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 128 * 128 + 512; i += blockDim.x) lds[i] = 1e-3f * (float)((i * 7) & 63) - 0.03f;
    __syncthreads();
    const float* wl = lds + lane * 4;
    const float* gl = lds + 2 * 128 * 128 + 4 * (lane >> 5);
    f32x16 h[4], o[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) h[j][r] = 0.01f * (lane + r + j);
    f32x4 aset[8];
    for (int i = 0; i < 8; ++i) aset[i] = f32x4{0.001f * (lane + i), 0.002f * (lane - i), 0.0005f * i, 0.003f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const float* w = wl + layer * 128 * 128;
            if constexpr (V & 4) {
                f32x4 pre = lds4(w);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.1f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        f32x4 a = aset[(i + 3 * j) & 7];
                        if constexpr ((V & 1) && !(V & 8)) a = lds4(w + (j * 16 + i) * 256);
                        if constexpr ((V & 1) && (V & 8)) {          // pinned one-group-ahead prefetch
                            a = pre;
                            asm volatile("" ::: "memory");
                            pre = lds4(w + ((j * 16 + i + 1) & 63) * 256);
                            asm volatile("" ::: "memory");
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], h[i >> 2][4 * (i & 3) + e], acc, 0, 0, 0);
                    }
                    if constexpr ((V & 2) && !(V & 16)) {
                        if constexpr (V & 32) __builtin_amdgcn_s_setprio(3);
                        if constexpr (V & 64) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 g = lds4(gl + 32 * j + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[4 * q + e] = softplus_l2(acc[4 * q + e] * g[e]);
                        }
                        if constexpr (V & 32) __builtin_amdgcn_s_setprio(0);
                        if constexpr (V & 64) __builtin_amdgcn_s_setprio(3);
                    }
                    if constexpr ((V & 2) && (V & 16)) {
                        if constexpr (V & 32) __builtin_amdgcn_s_setprio(3);
                        if constexpr (V & 64) __builtin_amdgcn_s_setprio(0);
                        float xs[16], ev[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 g = lds4(gl + 32 * j + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) xs[4 * q + e] = acc[4 * q + e] * g[e];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) ev[r] = __builtin_amdgcn_exp2f(xs[r]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) ev[r] = 1.0f + ev[r];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) ev[r] = __builtin_amdgcn_logf(ev[r]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_fmed3f(ev[r], xs[r], kSpThresh);
                        if constexpr (V & 32) __builtin_amdgcn_s_setprio(0);
                        if constexpr (V & 64) __builtin_amdgcn_s_setprio(3);
                    }
                    o[j] = acc;
                }
            } else {
                f32x16 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.1f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float a = aset[(i + 3 * j) & 7][e];
                            if constexpr (V & 1) a = w[((j * 16 + i) * 64) * 4 + e];
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[i >> 2][4 * (i & 3) + e], acc[j], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (V & 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 g = lds4(gl + 32 * j + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = softplus_l2(acc[j][4 * q + e] * g[e]);
                        }
                    }
                    o[j] = acc[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = o[j];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int j = 0; j < 4; ++j) sum += h[j][0] + h[j][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, int waves) {
    float* out; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int steps = 2000;
    const size_t ldsb = (2 * 128 * 128 + 512) * 4;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    k<V><<<256, waves * 64, ldsb>>>(steps, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<V><<<256, waves * 64, ldsb>>>(steps, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz; use wall time and the nominal 2.4 GHz for cycles
    printf("%-44s waves/WG %d: %.2f us/step = %.0f cycles/step (512 MFMA = 32768)\n", name, waves, ms * 1e3 / steps, ms * 1e-3 / steps * 2.4e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int waves : {4, 8}) {
        run<4>("chain, reg A, no epilogue", waves);
        run<5>("chain, LDS A, no epilogue", waves);
        run<6>("chain, reg A, softplus epilogue", waves);
        run<7>("chain, LDS A, softplus epilogue (kernel)", waves);
        run<13>("chain, LDS A prefetched, no epilogue", waves);
        run<15>("chain, LDS A prefetched, softplus epilogue", waves);
        run<15 + 16>("  + staged epilogue", waves);
        run<15 + 32>("  + epilogue at raised priority", waves);
        run<15 + 16 + 32>("  + staged, epilogue at raised priority", waves);
        run<15 + 16 + 64>("  + staged, MFMA chain at raised priority", waves);
        run<0>("4 accumulators, reg A, no epilogue", waves);
        run<1>("4 accumulators, LDS A, no epilogue", waves);
        run<3>("4 accumulators, LDS A, softplus epilogue", waves);
    }
    return 0;
}
