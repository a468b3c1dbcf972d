// Issue rate of single VALU instructions on gfx950, one opcode per kernel (inline asm, so nothing is fused or substituted), WITH the shader
// clock each kernel really ran at: every workgroup reads s_memtime (shader-clock ticks) and s_memrealtime (100 MHz reference) at both ends,
// the table prices a wave-instruction in cycles of the MEASURED clock.  Round 4's harness (tools/valu_classes.hip) assumed 2.4 GHz for
// sub-millisecond kernels and had no f32 control; this one runs every kernel for tens of milliseconds after a warm-up launch of the same
// length, has v_fma_f32 / v_add_f32 / v_mul_f32 as controls, and takes the waves per SIMD as an argument.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates [waves per SIMD = 8] [iterations = 150000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define DEFINE_KERNEL(NAME, ASM)                                                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters) {                  \
        uint32_t a[8], b = threadIdx.x * 2654435761u + seed, c = (b ^ 0x9e3779b9u) | 1u;                                             \
        for (int i = 0; i < 8; ++i) a[i] = b + i * 77u;                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                                 \
        for (int it = 0; it < iters; ++it) {                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                          \
                asm volatile(ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
                             : "v"(b), "v"(c) : "vcc");                                                                              \
            }                                                                                                                        \
        }                                                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                                 \
        uint32_t s = 0;                                                                                                              \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                              \
        if (threadIdx.x == 0) {                                                                                                      \
            clk[blockIdx.x * 2] = t1 - t0;                                                                                           \
            clk[blockIdx.x * 2 + 1] = r1 - r0;                                                                                       \
        }                                                                                                                            \
    }

#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %9\n" op " %2, %2, %8\n" op " %3, %3, %9\n" op " %4, %4, %8\n" op " %5, %5, %9\n" op " %6, %6, %8\n" op " %7, %7, %9\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %9, %8\n" op " %2, %2, %8, %9\n" op " %3, %3, %9, %8\n" op " %4, %4, %8, %9\n" op " %5, %5, %9, %8\n" op " %6, %6, %8, %9\n" op " %7, %7, %9, %8\n"
#define OPI(op, imm) op " %0, " imm ", %0\n" op " %1, " imm ", %1\n" op " %2, " imm ", %2\n" op " %3, " imm ", %3\n" op " %4, " imm ", %4\n" op " %5, " imm ", %5\n" op " %6, " imm ", %6\n" op " %7, " imm ", %7\n"
#define OPT(op, tail) op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n" op " %3, %3, " tail "\n" op " %4, %4, " tail "\n" op " %5, %5, " tail "\n" op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"

DEFINE_KERNEL(k_fma_f32, OP3("v_fma_f32"))
DEFINE_KERNEL(k_add_f32, OP2("v_add_f32"))
DEFINE_KERNEL(k_mul_f32, OP2("v_mul_f32"))
DEFINE_KERNEL(k_add, OP2("v_add_u32"))
DEFINE_KERNEL(k_sub, OP2("v_sub_u32"))
DEFINE_KERNEL(k_xor, OP2("v_xor_b32"))
DEFINE_KERNEL(k_and, OP2("v_and_b32"))
DEFINE_KERNEL(k_or, OP2("v_or_b32"))
DEFINE_KERNEL(k_min, OP2("v_min_u32"))
DEFINE_KERNEL(k_lshl, OPI("v_lshlrev_b32", "3"))
DEFINE_KERNEL(k_lshr, OPI("v_lshrrev_b32", "5"))
DEFINE_KERNEL(k_lshl_v, "v_lshlrev_b32 %0, %8, %0\nv_lshlrev_b32 %1, %9, %1\nv_lshlrev_b32 %2, %8, %2\nv_lshlrev_b32 %3, %9, %3\nv_lshlrev_b32 %4, %8, %4\nv_lshlrev_b32 %5, %9, %5\nv_lshlrev_b32 %6, %8, %6\nv_lshlrev_b32 %7, %9, %7\n")
DEFINE_KERNEL(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %8\nv_mov_b32 %3, %9\nv_mov_b32 %4, %8\nv_mov_b32 %5, %9\nv_mov_b32 %6, %8\nv_mov_b32 %7, %9\n")
DEFINE_KERNEL(k_add3, OP3("v_add3_u32"))
DEFINE_KERNEL(k_lshl_add, OPT("v_lshl_add_u32", "3, %8"))
DEFINE_KERNEL(k_bfe, OPT("v_bfe_u32", "3, 9"))
DEFINE_KERNEL(k_mad24, OP3("v_mad_u32_u24"))
DEFINE_KERNEL(k_mul24, OP2("v_mul_u32_u24"))
DEFINE_KERNEL(k_cndmask, OPT("v_cndmask_b32", "%8, vcc"))
DEFINE_KERNEL(k_cmp, "v_cmp_gt_u32 vcc, %0, %8\nv_cmp_gt_u32 vcc, %1, %9\nv_cmp_gt_u32 vcc, %2, %8\nv_cmp_gt_u32 vcc, %3, %9\nv_cmp_gt_u32 vcc, %4, %8\nv_cmp_gt_u32 vcc, %5, %9\nv_cmp_gt_u32 vcc, %6, %8\nv_cmp_gt_u32 vcc, %7, %9\n")
DEFINE_KERNEL(k_mul_lo, OP2("v_mul_lo_u32"))
DEFINE_KERNEL(k_mul_hi, OP2("v_mul_hi_u32"))
DEFINE_KERNEL(k_or3, OP3("v_or3_b32"))
DEFINE_KERNEL(k_and_or, OP3("v_and_or_b32"))
DEFINE_KERNEL(k_min3, OP3("v_min3_u32"))
DEFINE_KERNEL(k_xor_sdwa, OPT("v_xor_b32_sdwa", "%8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"))
// carry chain as the generator step has it: the writer of VCC and its reader two wait states apart (four chains interleaved)
DEFINE_KERNEL(k_addc, "v_add_co_u32 %0, vcc, %0, %8\nv_mov_b32 %4, %9\nv_mov_b32 %5, %8\nv_addc_co_u32 %1, vcc, %1, %9, vcc\nv_mov_b32 %6, %9\nv_mov_b32 %7, %8\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_mov_b32 %3, %9\n")

// v_mad_u64_u32 on even-aligned 64-bit pairs: its own kernel shape (four independent 64-bit chains)
__global__ __launch_bounds__(256) void k_mad64(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters) {
    unsigned long long a[4];
    uint32_t b = threadIdx.x * 2654435761u + seed, c = (b ^ 0x9e3779b9u) | 1u;
    for (int i = 0; i < 4; ++i) a[i] = b + i * 77u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %5, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %5, %4, %3\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a[0] + a[1] + a[2] + a[3]);
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

static int g_waves = 8, g_iters = 150000, g_cus = 256;

template <typename K>
void run(const char *name, K kern) {
    const int grid = g_cus * g_waves; // one 256-thread block = one wave per SIMD
    uint32_t *d;
    unsigned long long *clk;
    (void)hipMalloc(&d, (size_t)grid * 256 * 4);
    (void)hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, clk, 1u, g_iters); // warm-up of the same length
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, clk, 2u, g_iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h((size_t)grid * 2);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int i = 0; i < grid; ++i)
        if (h[2 * i + 1]) {
            mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
            cyc.push_back((double)h[2 * i]);
        }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    const double f = mhz[mhz.size() / 2], ticks = cyc[cyc.size() / 2];
    const double per_wave = (double)g_iters * 32.0;              // wave-instructions of one wave
    const double per_simd = per_wave * g_waves;                  // ... of the waves sharing a SIMD
    printf("%-16s %8.2f ms  clock %6.0f MHz (min %5.0f max %5.0f)  %5.2f cycles per wave-instruction per SIMD (in-kernel ticks)  %5.2f (event time x measured clock)  %5.2f (event time x 2400 MHz)\n",
           name, ms, f, mhz.front(), mhz.back(), ticks / per_simd, ms * 1e-3 * f * 1e6 / per_simd, ms * 1e-3 * 2.4e9 / per_simd);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(clk);
}

int main(int argc, char **argv) {
    if (argc > 1) g_waves = atoi(argv[1]);
    if (argc > 2) g_iters = atoi(argv[2]);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    printf("# %s, %d CUs, nominal %d MHz, %d waves per SIMD, %d iterations x 32 instructions per wave\n", p.name, g_cus, p.clockRate / 1000, g_waves, g_iters);
    run("v_fma_f32", k_fma_f32); run("v_add_f32", k_add_f32); run("v_mul_f32", k_mul_f32);
    run("v_add_u32", k_add); run("v_sub_u32", k_sub); run("v_xor_b32", k_xor); run("v_and_b32", k_and); run("v_or_b32", k_or);
    run("v_mov_b32", k_mov); run("v_min_u32", k_min); run("v_lshlrev_b32 imm", k_lshl); run("v_lshlrev_b32 vgpr", k_lshl_v); run("v_lshrrev_b32 imm", k_lshr);
    run("v_add3_u32", k_add3); run("v_lshl_add_u32", k_lshl_add); run("v_bfe_u32", k_bfe); run("v_or3_b32", k_or3); run("v_and_or_b32", k_and_or); run("v_min3_u32", k_min3);
    run("v_mad_u32_u24", k_mad24); run("v_mul_u32_u24", k_mul24); run("v_cndmask_b32", k_cndmask); run("v_cmp_gt_u32", k_cmp);
    run("v_xor_b32_sdwa", k_xor_sdwa); run("v_mul_lo_u32", k_mul_lo); run("v_mul_hi_u32", k_mul_hi); run("v_mad_u64_u32", k_mad64);
    run("add_co/addc+mov", k_addc);
    return 0;
}
