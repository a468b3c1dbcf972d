// Microbenchmark (round 4): issue rate of single integer VALU instructions, one opcode per kernel, written with inline asm so that
// the compiler cannot fuse or substitute them.  8 independent register chains x 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/valu_classes tools/valu_classes.hip && tools/valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITER 2048

#define DEFINE_KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                               \
        uint32_t a[8], b = threadIdx.x * 2654435761u + seed, c = b ^ 0x9e3779b9u;                            \
        for (int i = 0; i < 8; ++i) a[i] = b + i * 77u;                                                       \
        for (int it = 0; it < ITER; ++it) {                                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                   \
                asm volatile(ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                             : "v"(b), "v"(c));                                                               \
            }                                                                                                 \
        }                                                                                                     \
        uint32_t s = 0;                                                                                       \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
    }

#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %9\n" op " %2, %2, %8\n" op " %3, %3, %9\n" op " %4, %4, %8\n" op " %5, %5, %9\n" op " %6, %6, %8\n" op " %7, %7, %9\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %9, %8\n" op " %2, %2, %8, %9\n" op " %3, %3, %9, %8\n" op " %4, %4, %8, %9\n" op " %5, %5, %9, %8\n" op " %6, %6, %8, %9\n" op " %7, %7, %9, %8\n"
#define OPI(op, imm) op " %0, " imm ", %0\n" op " %1, " imm ", %1\n" op " %2, " imm ", %2\n" op " %3, " imm ", %3\n" op " %4, " imm ", %4\n" op " %5, " imm ", %5\n" op " %6, " imm ", %6\n" op " %7, " imm ", %7\n"

DEFINE_KERNEL(k_add, OP2("v_add_u32"))
DEFINE_KERNEL(k_sub, OP2("v_sub_u32"))
DEFINE_KERNEL(k_xor, OP2("v_xor_b32"))
DEFINE_KERNEL(k_and, OP2("v_and_b32"))
DEFINE_KERNEL(k_or, OP2("v_or_b32"))
DEFINE_KERNEL(k_min, OP2("v_min_u32"))
DEFINE_KERNEL(k_lshl, OPI("v_lshlrev_b32", "3"))
DEFINE_KERNEL(k_lshr, OPI("v_lshrrev_b32", "5"))
DEFINE_KERNEL(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %8\nv_mov_b32 %3, %9\nv_mov_b32 %4, %8\nv_mov_b32 %5, %9\nv_mov_b32 %6, %8\nv_mov_b32 %7, %9\n")
DEFINE_KERNEL(k_add3, OP3("v_add3_u32"))
DEFINE_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %8\nv_lshl_add_u32 %1, %1, 3, %9\nv_lshl_add_u32 %2, %2, 3, %8\nv_lshl_add_u32 %3, %3, 3, %9\nv_lshl_add_u32 %4, %4, 3, %8\nv_lshl_add_u32 %5, %5, 3, %9\nv_lshl_add_u32 %6, %6, 3, %8\nv_lshl_add_u32 %7, %7, 3, %9\n")
DEFINE_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 9\nv_bfe_u32 %1, %1, 3, 9\nv_bfe_u32 %2, %2, 3, 9\nv_bfe_u32 %3, %3, 3, 9\nv_bfe_u32 %4, %4, 3, 9\nv_bfe_u32 %5, %5, 3, 9\nv_bfe_u32 %6, %6, 3, 9\nv_bfe_u32 %7, %7, 3, 9\n")
DEFINE_KERNEL(k_mad24, OP3("v_mad_u32_u24"))
DEFINE_KERNEL(k_mul24, OP2("v_mul_u32_u24"))
DEFINE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %9, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %9, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %9, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %9, vcc\n")
DEFINE_KERNEL(k_cmp, "v_cmp_gt_u32 vcc, %0, %8\nv_cmp_gt_u32 vcc, %1, %9\nv_cmp_gt_u32 vcc, %2, %8\nv_cmp_gt_u32 vcc, %3, %9\nv_cmp_gt_u32 vcc, %4, %8\nv_cmp_gt_u32 vcc, %5, %9\nv_cmp_gt_u32 vcc, %6, %8\nv_cmp_gt_u32 vcc, %7, %9\n")
DEFINE_KERNEL(k_mul_lo, OP2("v_mul_lo_u32"))
DEFINE_KERNEL(k_or3, OP3("v_or3_b32"))
DEFINE_KERNEL(k_and_or, OP3("v_and_or_b32"))
DEFINE_KERNEL(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_xor_b32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")

template <typename K>
void run(const char *name, K kern) {
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 256 * 8; // 8 blocks of 256 per CU -> 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double waveinst = (double)grid * 4 * ITER * 32; // 4 waves per block, 32 instructions per iteration
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / waveinst);
    (void)hipFree(d);
}

int main() {
    run("v_add_u32", k_add); run("v_sub_u32", k_sub); run("v_xor_b32", k_xor); run("v_and_b32", k_and); run("v_or_b32", k_or);
    run("v_min_u32", k_min); run("v_lshlrev_b32", k_lshl); run("v_lshrrev_b32", k_lshr); run("v_mov_b32", k_mov);
    run("v_add3_u32", k_add3); run("v_lshl_add_u32", k_lshl_add); run("v_bfe_u32", k_bfe); run("v_mad_u32_u24", k_mad24);
    run("v_mul_u32_u24", k_mul24); run("v_cndmask_b32", k_cndmask); run("v_cmp_gt_u32", k_cmp); run("v_mul_lo_u32", k_mul_lo);
    run("v_or3_b32", k_or3); run("v_and_or_b32", k_and_or); run("v_xor_b32_sdwa", k_xor_sdwa);
    return 0;
}
