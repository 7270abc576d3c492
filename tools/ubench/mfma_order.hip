// Does v_mfma_f32_16x16x4_f32 over k = (k0, k1, k2, k3) give the same bits as two chained v_mfma_f32_32x32x2_f32 over
// (k0, k1) then (k2, k3) -- i.e. are both "sequential fp32 FMA over k in index order"?  Decides whether a 16-chain-tile
// kernel can be bit-identical to the 32-chain-tile kernels.  Also checks against a plain fmaf chain on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: 32 x 4 (row-major), B: 4 x 32, C0: 32 x 32.  out32: D via two 32x32x2; out16: D[0:16][0:16] via one 16x16x4
__global__ void k(const float* A, const float* B, const float* C0, float* out32, float* out16) {
    const int lane = threadIdx.x;
    {
        const int m = lane & 31, hh = lane >> 5;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = C0[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + m];
        // A operand: lane supplies A[i = lane&31][k = lane>>5]; B operand: B[k = lane>>5][n = lane&31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * 4 + hh], B[hh * 32 + m], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * 4 + 2 + hh], B[(2 + hh) * 32 + m], acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + m] = acc[r];
    }
    {
        const int n = lane & 15, g = lane >> 4;
        f32x4 acc;
        for (int r = 0; r < 4; ++r) acc[r] = C0[(4 * g + r) * 32 + n];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * 4 + g], B[g * 32 + n], acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out16[(4 * g + r) * 16 + n] = acc[r];
    }
}

int main() {
    float hA[128], hB[128], hC[1024], o32[1024], o16[256];
    float *A, *B, *C, *O32, *O16;
    hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&C, 4096); hipMalloc(&O32, 4096); hipMalloc(&O16, 1024);
    long mism = 0, mism_host = 0, mism_host16 = 0, total = 0;
    srand(1);
    for (int trial = 0; trial < 200; ++trial) {
        const float scale = trial % 3 == 0 ? 1e-3f : (trial % 3 == 1 ? 1.0f : 1e3f);
        for (int i = 0; i < 128; ++i) { hA[i] = scale * ((float)rand() / RAND_MAX - 0.5f); hB[i] = (float)rand() / RAND_MAX - 0.5f; }
        for (int i = 0; i < 1024; ++i) hC[i] = scale * 0.3f * ((float)rand() / RAND_MAX - 0.5f);
        hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice); hipMemcpy(C, hC, 4096, hipMemcpyHostToDevice);
        k<<<1, 64>>>(A, B, C, O32, O16);
        hipMemcpy(o32, O32, 4096, hipMemcpyDeviceToHost); hipMemcpy(o16, O16, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int n = 0; n < 16; ++n) {
                float ref = hC[i * 32 + n];
                for (int kk = 0; kk < 4; ++kk) ref = fmaf(hA[i * 4 + kk], hB[kk * 32 + n], ref);
                ++total;
                if (memcmp(&o32[i * 32 + n], &o16[i * 16 + n], 4)) ++mism;
                if (memcmp(&o32[i * 32 + n], &ref, 4)) ++mism_host;
                if (memcmp(&o16[i * 16 + n], &ref, 4)) ++mism_host16;
            }
    }
    printf("elements %ld: 32x32x2 chain vs 16x16x4 mismatches %ld; 32x32x2 vs host fmaf chain %ld; 16x16x4 vs host fmaf chain %ld\n",
           total, mism, mism_host, mism_host16);
    return 0;
}
