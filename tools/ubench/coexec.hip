// Micro-benchmark: do f32-input MFMA and VALU work from two waves on one SIMD overlap on gfx950?
// 8 waves per workgroup (waves w and w+4 share a SIMD), one workgroup per CU.
// mode 0: waves 0-3 run MFMA, waves 4-7 idle; mode 1: waves 0-3 idle, 4-7 run VALU; mode 2: both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MF, int VA>
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float* out, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_m = wave < 4 && (mode == 0 || mode == 2);
    const bool do_v = wave >= 4 && (mode == 1 || mode == 2);
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_m) {
        f32x16 acc = {0}; f32x4 acc4 = {0};
        s16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (MF == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                if (MF == 1) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
                if (MF == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[5] + acc4[1];
    }
    if (do_v) {
        float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (VA == 0) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
                if (VA == 1) { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3); }
                if (VA == 2) { x0 = x0 > b ? x0 : a; x1 = x1 > b ? x1 : a; x2 = x2 > b ? a : x2; x3 = x3 > b ? a : x3; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MF, int VA>
void run(const char* name, int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned long long> h(256 * 8);
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) k<MF, VA><<<256, 512>>>(mode, iters, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
        res[mode][0] = m / (256 * 4) / (iters * 16.0); res[mode][1] = v / (256 * 4) / (iters * 16.0);
    }
    printf("%-28s  MFMA-wave cyc/mfma: alone %.1f  with-VALU %.1f | VALU-wave cyc/(4 ops): alone %.1f  with-MFMA %.1f\n",
           name, res[0][0], res[2][0], res[1][1], res[2][1]);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<0, 0>("f32 32x32x2  + v_fma", it);
    run<0, 1>("f32 32x32x2  + v_exp", it);
    run<0, 2>("f32 32x32x2  + v_cmp/cndmask", it);
    run<1, 0>("f32 16x16x4  + v_fma", it);
    run<1, 1>("f32 16x16x4  + v_exp", it);
    run<2, 0>("bf16 32x32x16 + v_fma", it);
    run<2, 1>("bf16 32x32x16 + v_exp", it);
    return 0;
}
