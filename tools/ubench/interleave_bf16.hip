// Micro-benchmark: within ONE wave, how many VALU ops hide in the shadow of a v_mfma_f32_32x32x16_bf16 (gfx950)?
// (the bf16 twin of interleave.hip; SHAPE 2 = bf16 32x32x16, SHAPE 0 = f32 32x32x2 for reference)
// Each wave runs: { 1 MFMA (chain of NACC accumulators round-robin); K independent VALU ops } x iters.
// 1 or 2 waves per SIMD.  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int OP, int NACC, int SHAPE>
__global__ __launch_bounds__(512, 2) void k(int iters, float* out, unsigned long long* cyc) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    f32x16 acc[NACC]; f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (f32x16){0}; acc4[i] = (f32x4){0}; }
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 2) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[u % NACC], 0, 0, 0);
            else if (SHAPE == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
            else acc4[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u % NACC], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (OP == 0) x[v % 8] = fmaf(x[v % 8], b, a);
                if (OP == 1) x[v % 8] = __builtin_amdgcn_exp2f(x[v % 8]);
                if (OP == 2) { x[v % 8] = (v & 3) == 1 ? __builtin_amdgcn_exp2f(x[v % 8]) : fmaf(x[v % 8], b, a); }   // 1 transcendental in 4
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7] + acc4[i][1];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int OP, int NACC, int SHAPE>
double run(int threads) {
    const int iters = 5000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned long long> h(256 * 8);
    for (int rep = 0; rep < 2; ++rep) k<K, OP, NACC, SHAPE><<<256, threads>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; int nw = threads / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) m += h[b * 8 + w];
    (void)hipFree(out); (void)hipFree(cyc);
    return m / (256 * nw) / (iters * 8.0);
}

#define ROW(K) printf("K=%2d | bf16 32x32x16 1 acc, 1w/SIMD: fma %.1f exp %.1f mix %.1f | 2 acc: fma %.1f exp %.1f mix %.1f | 2 acc, 2w/SIMD (per-wave cyc per own MFMA): fma %.1f mix %.1f | f32 32x32x2: fma %.1f\n", K, \
    run<K,0,1,2>(256), run<K,1,1,2>(256), run<K,2,1,2>(256), run<K,0,2,2>(256), run<K,1,2,2>(256), run<K,2,2,2>(256), run<K,0,2,2>(512), run<K,2,2,2>(512), run<K,0,1,0>(256));
int main() {
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8) ROW(10) ROW(12) ROW(16)
    return 0;
}
