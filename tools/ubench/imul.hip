// Issue cost of the 32-bit integer multiplies behind Philox4x32 on gfx950: v_mul_lo_u32 + v_mul_hi_u32 (what the
// compiler emits for __umulhi(a, b) and a * b) against one v_mad_u64_u32 that yields both halves, and a whole
// Philox4x32-10 call written both ways.  One wave, independent instruction streams, cycles from s_memtime scaled by a
// v_add_u32 loop (4 cycles per instruction).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ void ops(uint32_t* out, long long* cyc, int n) {
    uint32_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, 0x9E3779B9, %0" : "+v"(v[i]));
            if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(0xD2511F53u));
            if (KIND == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(0xD2511F53u));
            if (KIND == 3) {
                unsigned long long p, c;
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(c) : "v"(v[i]), "v"(0xD2511F53u));
                v[i] = (uint32_t)p ^ (uint32_t)(p >> 32);
            }
            if (KIND == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(0x511F53u));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int FORM>
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (FORM == 0) {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        } else {
            unsigned long long p0, p1, cc;
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cc) : "v"(c0), "v"(0xD2511F53u));
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cc) : "v"(c2), "v"(0xCD9E8D57u));
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        }
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

template <int FORM>
__global__ void calls(uint32_t* out, long long* cyc, int n) {
    uint32_t acc[4] = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        uint32_t o[4];
        philox<FORM>(threadIdx.x, it, acc[0], 0u, 17u, 4u, o);      // dependent on the last call: no overlap between calls
        for (int i = 0; i < 4; ++i) acc[i] ^= o[i];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    uint32_t* o; long long* c;
    hipMalloc(&o, 64 * 4); hipMalloc(&c, 8);
    const int n = 4000;
    long long hc[5];
    const char* names[5] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (+ xor)", "v_mul_u32_u24"};
    for (int rep = 0; rep < 2; ++rep) {
        ops<0><<<1, 64>>>(o, c, n); hipMemcpy(&hc[0], c, 8, hipMemcpyDeviceToHost);
        ops<1><<<1, 64>>>(o, c, n); hipMemcpy(&hc[1], c, 8, hipMemcpyDeviceToHost);
        ops<2><<<1, 64>>>(o, c, n); hipMemcpy(&hc[2], c, 8, hipMemcpyDeviceToHost);
        ops<3><<<1, 64>>>(o, c, n); hipMemcpy(&hc[3], c, 8, hipMemcpyDeviceToHost);
        ops<4><<<1, 64>>>(o, c, n); hipMemcpy(&hc[4], c, 8, hipMemcpyDeviceToHost);
    }
    const double tick = 4.0 * 8 * n / (double)hc[0];              // cycles per s_memtime tick, from the v_add_u32 loop
    for (int k = 0; k < 5; ++k) printf("%-24s %.1f cycles per instruction (8 independent streams, one wave)\n", names[k], hc[k] * tick / (8.0 * n));
    uint32_t h0[64], h1[64];
    long long p0, p1;
    for (int rep = 0; rep < 2; ++rep) {
        calls<0><<<1, 64>>>(o, c, n); hipMemcpy(&p0, c, 8, hipMemcpyDeviceToHost); hipMemcpy(h0, o, sizeof(h0), hipMemcpyDeviceToHost);
        calls<1><<<1, 64>>>(o, c, n); hipMemcpy(&p1, c, 8, hipMemcpyDeviceToHost); hipMemcpy(h1, o, sizeof(h1), hipMemcpyDeviceToHost);
    }
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h0[i] != h1[i];
    printf("Philox4x32-10 call, mul_hi + mul_lo : %.0f cycles\n", p0 * tick / n);
    printf("Philox4x32-10 call, v_mad_u64_u32   : %.0f cycles (%s)\n", p1 * tick / n, bad ? "DIFFERENT BITS" : "same bits");
    return bad != 0;
}
