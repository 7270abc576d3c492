// The publish step of card_sample_coop16_kernel in isolation: two ds_write_b64 per lane (old) against v_permlane32_swap + one
// ds_write_b128 (new).  With __builtin_amdgcn_permlane32_swap (ROCm 7.2) the second result is lost when both results feed a
// vector build (the b128 holds the first result twice): the kernel uses inline asm instead.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* oa, float* ob) {
    __shared__ __attribute__((aligned(16))) float A[16 * 132], B[16 * 132];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    for (int i = lane; i < 16 * 132; i += 64) { A[i] = -1; B[i] = -1; }
    __syncthreads();
    float a[4];
    for (int r = 0; r < 4; ++r) a[r] = 1000 * g + 10 * n + r;
    float* d = A + n * 132 + 4 * (g & 1) + 2 * (g >> 1);
    *reinterpret_cast<float2*>(d) = float2{a[0], a[2]};
    *reinterpret_cast<float2*>(d + 8) = float2{a[1], a[3]};
    const u32x2 sx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]), false, false);
    const u32x2 sy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3]), false, false);
    const f32x4 v = {__builtin_bit_cast(float, sx[0]), __builtin_bit_cast(float, sy[0]), __builtin_bit_cast(float, sx[1]), __builtin_bit_cast(float, sy[1])};
    *reinterpret_cast<f32x4*>(B + n * 132 + 4 * (g & 1) + 8 * (g >> 1)) = v;
    __syncthreads();
    for (int i = lane; i < 16 * 132; i += 64) { oa[i] = A[i]; ob[i] = B[i]; }
}
int main() {
    float *a, *b; static float ha[16 * 132], hb[16 * 132];
    if (hipMalloc(&a, sizeof(ha)) || hipMalloc(&b, sizeof(hb))) return 1;
    k<<<1, 64>>>(a, b);
    if (hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost) || hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost)) return 1;
    int bad = 0;
    for (int i = 0; i < 16 * 132; ++i) if (ha[i] != hb[i]) { if (bad < 12) printf("row %d slot %d: old %g new %g\n", i / 132, i % 132, ha[i], hb[i]); ++bad; }
    printf("mismatches %d\nrow 3 old:", bad);
    for (int i = 0; i < 16; ++i) printf(" %g", ha[3 * 132 + i]);
    printf("\nrow 3 new:");
    for (int i = 0; i < 16; ++i) printf(" %g", hb[3 * 132 + i]);
    printf("\n");
    return 0;
}
