// v_mfma_f32_16x16x1_4b_f32 (four independent 16 x 16 x 1 blocks per instruction): operand / result layout probe and
// issue cost next to v_mfma_f32_32x32x2_f32.  Question behind it: can lin4 of the F = 128 classification kernel (16 or
// fewer classes, 32 chains, layer 3's accumulators as the B operand) run in half the matrix-pipe time?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    d = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(1 + l), (float)(100 + l), d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = d[r];
}

template <int KIND>
__global__ void chain(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, d, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += d[r];
    out[l] = s;
    if (l == 0) *cyc = t1 - t0;
}

int main() {
    float* o; long long* c;
    hipMalloc(&o, 64 * 16 * 4); hipMalloc(&c, 8);
    probe<<<1, 64>>>(o);
    float h[64 * 16];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int bb = 0; bb < 4; ++bb)
            for (int r = 0; r < 4; ++r) {
                const float want = (float)(1 + 16 * bb + 4 * (l >> 4) + r) * (float)(100 + 16 * bb + (l & 15));
                if (h[l * 16 + 4 * bb + r] != want) {
                    if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, 4 * bb + r, h[l * 16 + 4 * bb + r], want);
                    ++bad;
                }
            }
    printf("layout hypothesis (A: block l>>4 row l&15; B: block l>>4 col l&15; D reg 4b+r = block b, row 4(l>>4)+r, col l&15): %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);
    const int n = 2000;
    long long hc;
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) chain<0><<<1, 64>>>(o, c, n); else chain<1><<<1, 64>>>(o, c, n);
            hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f clock ticks per dependent MFMA (s_memtime units)\n", kind == 0 ? "32x32x2_f32   " : "16x16x1_4b_f32", (double)hc / (16.0 * n));
    }
    return 0;
}
