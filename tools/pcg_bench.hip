// Cycles per PCG64DXSM draw on gfx950 at the game kernel's occupancy (6 waves per SIMD): the compiler's lowering of the 128-bit expression
// against hand-laid variants of the draw (tools/pcg_hand.h FK_PCG_DRAW and alternatives).  Every variant's state after the loop is checked against the
// plain C version.  Event time x the shader clock measured inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/pcg_bench tools/pcg_bench.hip && tools/pcg_bench [iterations]
#define FK_PCG_HAND // the hand-laid draw (tools/pcg_hand.h) beside the shipped pcg_next64_plain
#include "../farkle_ii_amd/csrc/fk_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fk;

#define SCRATCH "vcc", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

// V2: the output's 64-bit product with one v_mad_u64_u32 into a scratch pair + a move (instead of v_mul_lo + v_mul_hi)
#define DRAW_V2(LO, HI)                                                                                 \
    "v_mul_lo_u32 v24, %3, %17\n\t"                                                                     \
    "v_xor_b32_e32 v25, %2, %3\n\t"                                                                     \
    "v_mad_u64_u32 v[22:23], vcc, v25, %17, 0\n\t"                                                      \
    "v_mul_lo_u32 v25, v25, %18\n\t"                                                                    \
    "v_mad_u64_u32 v[16:17], %11, %0, %17, %15\n\t"                                                     \
    "v_add3_u32 v23, v23, v25, v24\n\t"                                                                 \
    "v_mad_u64_u32 v[18:19], vcc, %1, %17, 0\n\t"                                                       \
    "v_xor_b32_sdwa v22, v23, v22 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" \
    "v_or_b32_e32 v25, 1, %0\n\t"                                                                       \
    "v_addc_co_u32_e64 v19, vcc, v19, 0, %11\n\t"                                                       \
    "v_mul_lo_u32 v23, v23, v25\n\t"                                                                    \
    "v_mad_u64_u32 v[18:19], %12, %0, %18, v[18:19]\n\t"                                                \
    "v_mul_lo_u32 v27, v22, %1\n\t"                      /* h0 s1 -> high word of the addend pair */      \
    "v_mov_b32_e32 v26, 0\n\t"                                                                          \
    "v_mad_u64_u32 v[20:21], vcc, %2, %17, %16\n\t"                                                     \
    "v_mad_u64_u32 v[26:27], vcc, v22, v25, v[26:27]\n\t" /* h0 l0 + (h0 s1 << 32) */                    \
    "v_mad_u64_u32 v[20:21], vcc, %1, %18, v[20:21]\n\t"                                                \
    "v_mul_lo_u32 v25, %2, %18\n\t"                                                                     \
    "v_mov_b32_e32 " LO ", v26\n\t"                                                                     \
    "v_add_u32_e32 " HI ", v27, v23\n\t"                                                                \
    "v_addc_co_u32_e64 v25, vcc, v25, v24, %12\n\t"                                                     \
    "v_mov_b32_e32 %0, v16\n\t"                                                                         \
    "v_add_co_u32_e32 %1, vcc, v17, v18\n\t"                                                            \
    "v_mov_b32_e32 %10, " HI "\n\t"                                                                     \
    "s_nop 0\n\t"                                                                                       \
    "v_addc_co_u32_e32 %2, vcc, v20, v19, vcc\n\t"                                                      \
    "s_nop 1\n\t"                                                                                       \
    "v_addc_co_u32_e32 %3, vcc, v21, v25, vcc\n\t"

// V3: both carries through VCC and the e32 forms of v_addc (junk carries of the other multiply-adds go to the scratch pair %12)
#define DRAW_V3(LO, HI)                                                                                 \
    "v_mul_lo_u32 v24, %3, %17\n\t"                                                                     \
    "v_xor_b32_e32 v25, %2, %3\n\t"                                                                     \
    "v_mad_u64_u32 v[22:23], %12, v25, %17, 0\n\t"                                                      \
    "v_mul_lo_u32 v25, v25, %18\n\t"                                                                    \
    "v_mad_u64_u32 v[18:19], %12, %1, %17, 0\n\t"                                                       \
    "v_mad_u64_u32 v[16:17], vcc, %0, %17, %15\n\t"       /* cA -> vcc */                                \
    "v_add3_u32 v23, v23, v25, v24\n\t"                                                                 \
    "v_or_b32_e32 v25, 1, %0\n\t"                                                                       \
    "v_addc_co_u32_e32 v19, vcc, 0, v19, vcc\n\t"                                                       \
    "v_xor_b32_sdwa v22, v23, v22 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" \
    "v_mul_lo_u32 v23, v23, v25\n\t"                                                                    \
    "v_mad_u64_u32 v[18:19], vcc, %0, %18, v[18:19]\n\t"  /* cB -> vcc */                                \
    "v_mul_lo_u32 v26, v22, %1\n\t"                                                                     \
    "v_mul_lo_u32 v27, %2, %18\n\t"                                                                     \
    "v_addc_co_u32_e32 v24, vcc, v27, v24, vcc\n\t"       /* T */                                        \
    "v_mad_u64_u32 v[20:21], %12, %2, %17, %16\n\t"                                                     \
    "v_mul_hi_u32 v27, v22, v25\n\t"                                                                    \
    "v_mad_u64_u32 v[20:21], %12, %1, %18, v[20:21]\n\t"                                                \
    "v_mul_lo_u32 " LO ", v22, v25\n\t"                                                                 \
    "v_add3_u32 " HI ", v27, v26, v23\n\t"                                                              \
    "v_mov_b32_e32 %0, v16\n\t"                                                                         \
    "v_add_co_u32_e32 %1, vcc, v17, v18\n\t"                                                            \
    "v_mov_b32_e32 %10, " HI "\n\t"                                                                     \
    "s_nop 0\n\t"                                                                                       \
    "v_addc_co_u32_e32 %2, vcc, v20, v19, vcc\n\t"                                                      \
    "s_nop 1\n\t"                                                                                       \
    "v_addc_co_u32_e32 %3, vcc, v21, v24, vcc\n\t"

#define OPERANDS                                                                                                                     \
    : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(lo0), "=&v"(hi0), "=&v"(lo1), "=&v"(hi1), "=&v"(lo2), "=&v"(hi2), "+v"(last), "=&s"(cA),   \
      "=&s"(cB), "=&s"(ex)                                                                                                           \
    : "v"(need), "v"(inc_lo), "v"(inc_hi), "s"(M0), "s"(M1)                                                                          \
    : SCRATCH

template <int V>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(6))) void k_draws(uint4 *state, unsigned long long *clk, int iters) {
    constexpr uint32_t M0 = (uint32_t)PCG_CHEAP_MULT, M1 = (uint32_t)(PCG_CHEAP_MULT >> 32);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s0 = gid * 2654435761u + 1u, s1 = gid ^ 0x9e3779b9u, s2 = gid * 40503u + 7u, s3 = ~gid;
    const uint64_t inc_lo = ((uint64_t)(gid * 77u + 5u) << 32) | (gid * 2u + 1u), inc_hi = ((uint64_t)(gid + 11u) << 32) | (gid * 13u);
    uint32_t acc = 0, need = 3, lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0, last = 0;
    uint64_t cA, cB, ex;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { // the compiler's lowering
            Rng r{(uint64_t)s2 | ((uint64_t)s3 << 32), (uint64_t)s0 | ((uint64_t)s1 << 32), inc_hi, inc_lo, 0, 0};
            const uint64_t o0 = pcg_next64_plain(r), o1 = pcg_next64_plain(r), o2 = pcg_next64_plain(r);
            lo0 = (uint32_t)o0, hi0 = (uint32_t)(o0 >> 32), lo1 = (uint32_t)o1, hi1 = (uint32_t)(o1 >> 32), lo2 = (uint32_t)o2, hi2 = (uint32_t)(o2 >> 32);
            s0 = (uint32_t)r.lo, s1 = (uint32_t)(r.lo >> 32), s2 = (uint32_t)r.hi, s3 = (uint32_t)(r.hi >> 32);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        else if (V == 1) {
            asm volatile(FK_PCG_DRAW("%4", "%5") FK_PCG_DRAW("%6", "%7") FK_PCG_DRAW("%8", "%9") "s_nop 0" OPERANDS);
        } else if (V == 2) {
            asm volatile(DRAW_V2("%4", "%5") DRAW_V2("%6", "%7") DRAW_V2("%8", "%9") "s_nop 0" OPERANDS);
        } else if (V == 3) {
            asm volatile(DRAW_V3("%4", "%5") DRAW_V3("%6", "%7") DRAW_V3("%8", "%9") "s_nop 0" OPERANDS);
        } else if (V == 4) { // the product form: exec narrowing + branches of pcg_draws (need = 3: every draw runs)
            pcg_draws(s0, s1, s2, s3, inc_lo, inc_hi, need, lo0, hi0, lo1, hi1, lo2, hi2, last);
        }
#endif
        acc ^= lo0 ^ hi0 ^ lo1 ^ hi1 ^ lo2 ^ hi2;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    state[gid] = make_uint4(s0 ^ acc, s1, s2, s3);
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

static std::vector<uint4> g_ref;

template <int V>
void run(const char *name, int iters) {
    const int grid = 256 * 2;
    uint4 *d;
    unsigned long long *clk;
    (void)hipMalloc(&d, (size_t)grid * 768 * 16);
    (void)hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_draws<V>, dim3(grid), dim3(768), 0, 0, d, clk, iters);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_draws<V>, dim3(grid), dim3(768), 0, 0, d, clk, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h((size_t)grid * 2);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < grid; ++i)
        if (h[2 * i + 1]) mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    std::sort(mhz.begin(), mhz.end());
    const double f = mhz[mhz.size() / 2];
    std::vector<uint4> out((size_t)grid * 768);
    (void)hipMemcpy(out.data(), d, out.size() * 16, hipMemcpyDeviceToHost);
    size_t bad = 0;
    if (V == 0) g_ref = out;
    else
        for (size_t i = 0; i < out.size(); ++i) bad += (out[i].x != g_ref[i].x || out[i].y != g_ref[i].y || out[i].z != g_ref[i].z || out[i].w != g_ref[i].w);
    const double draws_per_simd = (double)iters * 3.0 * 6.0;
    printf("%-44s %8.2f ms  clock %5.0f MHz  %7.1f cycles per draw per SIMD   mismatching lanes vs plain: %zu\n", name, ms, f, ms * 1e-3 * f * 1e6 / draws_per_simd, bad);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(clk);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    run<0>("compiler lowering (round 4)", iters);
    run<1>("hand-laid (FK_PCG_DRAW)", iters);
    run<2>("hand-laid, output product by v_mad_u64", iters);
    run<3>("hand-laid, carries in VCC (e32 v_addc)", iters);
    run<4>("pcg_draws (exec narrowing, need = 3)", iters);
    return 0;
}
