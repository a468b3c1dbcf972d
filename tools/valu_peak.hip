// Microbenchmark: sustained wave64 issue rate of the integer VALU ops the game kernel leans on.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/valu_peak.hip && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234567, a3 = a0 + 77, a4 = a1 + 5, a5 = a2 + 9, a6 = a3 ^ 3, a7 = a0 * 7;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) { // v_add_u32
            a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
        } else if (OP == 1) { // v_mul_lo_u32
            a0 *= a1; a1 *= a2; a2 *= a3; a3 *= a4; a4 *= a5; a5 *= a6; a6 *= a7; a7 *= a0;
        } else if (OP == 2) { // v_mul_hi_u32
            a0 = __umulhi(a0, a1); a1 = __umulhi(a1, a2); a2 = __umulhi(a2, a3); a3 = __umulhi(a3, a4);
            a4 = __umulhi(a4, a5); a5 = __umulhi(a5, a6); a6 = __umulhi(a6, a7); a7 = __umulhi(a7, a0) | 0x80000001u;
        } else if (OP == 3) { // v_mad_u64_u32
            b0 = (uint64_t)(uint32_t)b1 * (uint32_t)b2 + b0; b1 = (uint64_t)(uint32_t)b2 * (uint32_t)b3 + b1;
            b2 = (uint64_t)(uint32_t)b3 * (uint32_t)b0 + b2; b3 = (uint64_t)(uint32_t)b0 * (uint32_t)b1 + b3;
            b0 = (uint64_t)(uint32_t)b1 * (uint32_t)b2 + b0; b1 = (uint64_t)(uint32_t)b2 * (uint32_t)b3 + b1;
            b2 = (uint64_t)(uint32_t)b3 * (uint32_t)b0 + b2; b3 = (uint64_t)(uint32_t)b0 * (uint32_t)b1 + b3;
        } else if (OP == 4) { // v_xor / v_lshl mix
            a0 ^= a1 << 3; a1 ^= a2 >> 5; a2 ^= a3 << 7; a3 ^= a4 >> 2; a4 ^= a5 << 1; a5 ^= a6 >> 9; a6 ^= a7 << 4; a7 ^= a0 >> 6;
        } else if (OP == 5) { // v_cndmask (compare + select)
            a0 = a1 > a2 ? a3 : a0; a1 = a2 > a3 ? a4 : a1; a2 = a3 > a4 ? a5 : a2; a3 = a4 > a5 ? a6 : a3;
            a4 = a5 > a6 ? a7 : a4; a5 = a6 > a7 ? a0 : a5; a6 = a7 > a0 ? a1 : a6; a7 = a0 > a1 ? a2 : a7;
        } else if (OP == 6) { // v_mul_u32_u24
            a0 = __umul24(a0, a1); a1 = __umul24(a1, a2); a2 = __umul24(a2, a3); a3 = __umul24(a3, a4);
            a4 = __umul24(a4, a5); a5 = __umul24(a5, a6); a6 = __umul24(a6, a7); a7 = __umul24(a7, a0) | 3;
        } else if (OP == 7) { // 64-bit multiply low (what the compiler emits for u64*u64)
            b0 *= b1; b1 *= b2; b2 *= b3; b3 *= b0; b0 *= b1 | 1; b1 *= b2 | 1; b2 *= b3 | 1; b3 *= b0 | 1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(b0 + b1 + b2 + b3);
}

template <int OP>
void run(const char *name, int ops_per_iter) {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 8; // 8 blocks of 256 per CU -> 8 waves / SIMD
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)grid * 256 * ITER * ops_per_iter;
    double waveinst = lane_ops / 64;
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / waveinst;
    printf("%-28s %8.3f ms  %8.2f Tlane-op/s  %6.2f cycles/wave-inst/SIMD (at 2.4 GHz)\n", name, ms, lane_ops / ms * 1e-9, cyc);
    hipFree(d);
}

int main() {
    run<0>("v_add_u32", 8);
    run<4>("v_xor+shift (2 ops)", 16);
    run<5>("v_cmp+v_cndmask (2 ops)", 16);
    run<6>("v_mul_u32_u24", 8);
    run<1>("v_mul_lo_u32", 8);
    run<2>("v_mul_hi_u32", 8);
    run<3>("v_mad_u64_u32", 8);
    run<7>("u64*u64 low (compiler seq)", 8);
    return 0;
}
