// Second issue-rate harness (round 5): the instruction FORMS the game kernels use around lane masks and carries — v_cndmask_b32 on VCC / on an SGPR
// pair, back to back and interleaved, v_addc_co_u32 in both encodings, v_mad_u64_u32 with its carry-out in VCC or in an SGPR pair, 64-bit
// adds, literal operands, s_nop between vector instructions.  Same method as tools/valu_rates.hip: event time x the clock measured inside.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates2 tools/valu_rates2.hip && tools/valu_rates2 [waves per SIMD = 6] [iterations]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

// NV = vector instructions per asm group (x4 groups per iteration)
#define DEFINE_KERNEL(NAME, ASM)                                                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters) {                  \
        uint32_t a[8], b = threadIdx.x * 2654435761u + seed, c = (b ^ 0x9e3779b9u) | 1u;                                             \
        unsigned long long m = 0x5555aaaa3333ccccull ^ seed, q[4];                                                                   \
        for (int i = 0; i < 8; ++i) a[i] = b + i * 77u;                                                                              \
        for (int i = 0; i < 4; ++i) q[i] = b + i * 77u;                                                                              \
        asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");                                                                          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                                 \
        for (int it = 0; it < iters; ++it) {                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                          \
                asm volatile(ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                             "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+s"(m)                                                 \
                             : "v"(b), "v"(c) : "vcc");                                                                              \
            }                                                                                                                        \
        }                                                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                                 \
        uint32_t s = (uint32_t)m;                                                                                                    \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                                       \
        for (int i = 0; i < 4; ++i) s += (uint32_t)q[i];                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                              \
        if (threadIdx.x == 0) {                                                                                                      \
            clk[blockIdx.x * 2] = t1 - t0;                                                                                           \
            clk[blockIdx.x * 2 + 1] = r1 - r0;                                                                                       \
        }                                                                                                                            \
    }
// operands: %0-%7 a[], %8-%11 q[] (64-bit pairs), %12 m (SGPR pair), %13 b, %14 c
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CND_VCC(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %13, vcc\n"
#define CND_SG(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %13, %12\n"
#define ADD(i) "v_add_u32_e32 %" #i ", %" #i ", %13\n"
#define CND_VCC_ADD(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %13, vcc\nv_add_u32_e32 %" #i ", %" #i ", %14\n"
#define ADDC_E64(i) "v_addc_co_u32_e64 %" #i ", vcc, %" #i ", %13, %12\n"
#define ADDC_E32(i) "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %13, vcc\n"
#define ADDCO_E32(i) "v_add_co_u32_e32 %" #i ", vcc, %" #i ", %13\n"
#define ADD_NOP0(i) "v_add_u32_e32 %" #i ", %" #i ", %13\ns_nop 0\n"
#define ADD_NOP1(i) "v_add_u32_e32 %" #i ", %" #i ", %13\ns_nop 1\n"
#define MOV_LIT(i) "v_mov_b32_e32 %" #i ", 0x12345678\n"
#define AND_LIT(i) "v_and_b32_e32 %" #i ", 0x7fff1234, %" #i "\n"
#define ADD_SG(i) "v_add_u32_e32 %" #i ", s12, %" #i "\n"
#define CMP_SG(i) "v_cmp_gt_u32_e64 %12, %" #i ", %13\n"
#define CMP_VCC(i) "v_cmp_gt_u32_e32 vcc, %" #i ", %13\n"
#define BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %13, %14 bitop3:0x6c\n"
#define XAD(i) "v_xad_u32 %" #i ", %" #i ", %13, %14\n"
#define MED3(i) "v_med3_i32 %" #i ", %" #i ", %13, %14\n"
#define SUBREV(i) "v_subrev_u32_e32 %" #i ", %13, %" #i "\n"
#define LSHL1(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
#define ASHR(i) "v_ashrrev_i32_e32 %" #i ", 3, %" #i "\n"
#define NOT(i) "v_not_b32_e32 %" #i ", %" #i "\n"
#define MAXU(i) "v_max_u32_e32 %" #i ", %" #i ", %13\n"
#define CND_E64_VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %13, vcc\n"

DEFINE_KERNEL(k_add, R8(ADD))
DEFINE_KERNEL(k_cnd_vcc, R8(CND_VCC))
DEFINE_KERNEL(k_cnd_e64_vcc, R8(CND_E64_VCC))
DEFINE_KERNEL(k_cnd_sg, R8(CND_SG))
DEFINE_KERNEL(k_cnd_vcc_add, CND_VCC_ADD(0) CND_VCC_ADD(1) CND_VCC_ADD(2) CND_VCC_ADD(3)) // 8 instructions: 4 cndmask + 4 add
// the increment select tree of fk_play_hc_kernel as the compiler emits it: four e32 selects on VCC back to back, then three e64 ones on SGPR pairs (8 per group)
DEFINE_KERNEL(k_tree_compiler, CND_VCC(0) CND_VCC(1) CND_VCC(2) CND_VCC(3) CND_SG(4) CND_SG(5) CND_SG(6) ADD(7))
DEFINE_KERNEL(k_tree_e64, CND_E64_VCC(0) CND_E64_VCC(1) CND_E64_VCC(2) CND_E64_VCC(3) CND_SG(4) CND_SG(5) CND_SG(6) ADD(7))
DEFINE_KERNEL(k_tree_interleaved, CND_VCC(0) CND_SG(4) CND_VCC(1) CND_SG(5) CND_VCC(2) CND_SG(6) CND_VCC(3) ADD(7))
DEFINE_KERNEL(k_cnd2_add2, CND_VCC(0) CND_VCC(1) ADD(2) ADD(3) CND_VCC(4) CND_VCC(5) ADD(6) ADD(7))
// round 6: at which run length / spacing does the e32 form on VCC stop being full rate?  Groups of eight: R selects in a row, then 8 - R adds;
// one select every second / fourth instruction; a run that starts with the s_mov / v_cmp that sets VCC (what a hand-laid select tree would do).
DEFINE_KERNEL(k_run1, CND_VCC(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_run2, CND_VCC(0) CND_VCC(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_run3, CND_VCC(0) CND_VCC(1) CND_VCC(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_run4, CND_VCC(0) CND_VCC(1) CND_VCC(2) CND_VCC(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_run6, CND_VCC(0) CND_VCC(1) CND_VCC(2) CND_VCC(3) CND_VCC(4) CND_VCC(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_every2, CND_VCC(0) ADD(1) CND_VCC(2) ADD(3) CND_VCC(4) ADD(5) CND_VCC(6) ADD(7))
DEFINE_KERNEL(k_every4, CND_VCC(0) ADD(1) ADD(2) ADD(3) CND_VCC(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_every2_e64, CND_SG(0) ADD(1) CND_SG(2) ADD(3) CND_SG(4) ADD(5) CND_SG(6) ADD(7))
DEFINE_KERNEL(k_run4_e64, CND_SG(0) CND_SG(1) CND_SG(2) CND_SG(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_smov_run4, "s_mov_b64 vcc, %12\n" CND_VCC(0) CND_VCC(1) CND_VCC(2) CND_VCC(3) ADD(4) ADD(5) ADD(6) ADD(7))
DEFINE_KERNEL(k_smov_every2, "s_mov_b64 vcc, %12\n" CND_VCC(0) ADD(1) CND_VCC(2) ADD(3) CND_VCC(4) ADD(5) CND_VCC(6) ADD(7))
DEFINE_KERNEL(k_every2_mad, "v_cndmask_b32_e32 %0, %0, %13, vcc\nv_mad_u64_u32 %8, %12, %13, %14, %8\nv_cndmask_b32_e32 %1, %1, %13, vcc\nv_mad_u64_u32 %9, %12, %14, %13, %9\n"
                            "v_cndmask_b32_e32 %2, %2, %13, vcc\nv_mad_u64_u32 %10, %12, %13, %14, %10\nv_cndmask_b32_e32 %3, %3, %13, vcc\nv_mad_u64_u32 %11, %12, %14, %13, %11\n")
DEFINE_KERNEL(k_cnd_vcc_nop, "v_cndmask_b32_e32 %0, %0, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %1, %1, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %2, %2, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %3, %3, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %4, %4, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %5, %5, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %6, %6, %13, vcc\ns_nop 0\nv_cndmask_b32_e32 %7, %7, %13, vcc\ns_nop 0\n")
DEFINE_KERNEL(k_cnd_vcc_src, "v_cndmask_b32_e32 %0, %13, %14, vcc\nv_cndmask_b32_e32 %1, %14, %13, vcc\nv_cndmask_b32_e32 %2, %13, %14, vcc\nv_cndmask_b32_e32 %3, %14, %13, vcc\nv_cndmask_b32_e32 %4, %13, %14, vcc\nv_cndmask_b32_e32 %5, %14, %13, vcc\nv_cndmask_b32_e32 %6, %13, %14, vcc\nv_cndmask_b32_e32 %7, %14, %13, vcc\n")
DEFINE_KERNEL(k_addc_e64, R8(ADDC_E64))
DEFINE_KERNEL(k_addc_e32, R8(ADDC_E32))
DEFINE_KERNEL(k_addco_e32, R8(ADDCO_E32))
DEFINE_KERNEL(k_add_nop0, R8(ADD_NOP0))
DEFINE_KERNEL(k_add_nop1, R8(ADD_NOP1))
DEFINE_KERNEL(k_mov_lit, R8(MOV_LIT))
DEFINE_KERNEL(k_and_lit, R8(AND_LIT))
DEFINE_KERNEL(k_cmp_sg, R8(CMP_SG))
DEFINE_KERNEL(k_cmp_vcc, R8(CMP_VCC))
DEFINE_KERNEL(k_bitop3, R8(BITOP))
DEFINE_KERNEL(k_xad, R8(XAD))
DEFINE_KERNEL(k_med3, R8(MED3))
DEFINE_KERNEL(k_subrev, R8(SUBREV))
DEFINE_KERNEL(k_lshl1, R8(LSHL1))
DEFINE_KERNEL(k_ashr, R8(ASHR))
DEFINE_KERNEL(k_not, R8(NOT))
DEFINE_KERNEL(k_max, R8(MAXU))
// 64-bit forms: 8 instructions per group on the four pairs (two rounds)
DEFINE_KERNEL(k_mad64_vcc, "v_mad_u64_u32 %8, vcc, %13, %14, %8\nv_mad_u64_u32 %9, vcc, %14, %13, %9\nv_mad_u64_u32 %10, vcc, %13, %14, %10\nv_mad_u64_u32 %11, vcc, %14, %13, %11\n"
                           "v_mad_u64_u32 %8, vcc, %13, %14, %8\nv_mad_u64_u32 %9, vcc, %14, %13, %9\nv_mad_u64_u32 %10, vcc, %13, %14, %10\nv_mad_u64_u32 %11, vcc, %14, %13, %11\n")
DEFINE_KERNEL(k_mad64_sg, "v_mad_u64_u32 %8, %12, %13, %14, %8\nv_mad_u64_u32 %9, %12, %14, %13, %9\nv_mad_u64_u32 %10, %12, %13, %14, %10\nv_mad_u64_u32 %11, %12, %14, %13, %11\n"
                          "v_mad_u64_u32 %8, %12, %13, %14, %8\nv_mad_u64_u32 %9, %12, %14, %13, %9\nv_mad_u64_u32 %10, %12, %13, %14, %10\nv_mad_u64_u32 %11, %12, %14, %13, %11\n")
DEFINE_KERNEL(k_mad64_zero, "v_mad_u64_u32 %8, vcc, %0, %14, 0\nv_mad_u64_u32 %9, vcc, %1, %13, 0\nv_mad_u64_u32 %10, vcc, %2, %14, 0\nv_mad_u64_u32 %11, vcc, %3, %13, 0\n"
                            "v_mad_u64_u32 %8, vcc, %4, %14, 0\nv_mad_u64_u32 %9, vcc, %5, %13, 0\nv_mad_u64_u32 %10, vcc, %6, %14, 0\nv_mad_u64_u32 %11, vcc, %7, %13, 0\n")
DEFINE_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %9\nv_lshl_add_u64 %9, %9, 0, %10\nv_lshl_add_u64 %10, %10, 0, %11\nv_lshl_add_u64 %11, %11, 0, %8\n"
                              "v_lshl_add_u64 %8, %8, 0, %9\nv_lshl_add_u64 %9, %9, 0, %10\nv_lshl_add_u64 %10, %10, 0, %11\nv_lshl_add_u64 %11, %11, 0, %8\n")
DEFINE_KERNEL(k_mov_b64, "v_mov_b64 %8, %9\nv_mov_b64 %9, %10\nv_mov_b64 %10, %11\nv_mov_b64 %11, %8\nv_mov_b64 %8, %9\nv_mov_b64 %9, %10\nv_mov_b64 %10, %11\nv_mov_b64 %11, %8\n")
DEFINE_KERNEL(k_lshlrev_b64, "v_lshlrev_b64 %8, 3, %8\nv_lshlrev_b64 %9, 3, %9\nv_lshlrev_b64 %10, 3, %10\nv_lshlrev_b64 %11, 3, %11\nv_lshlrev_b64 %8, 3, %8\nv_lshlrev_b64 %9, 3, %9\nv_lshlrev_b64 %10, 3, %10\nv_lshlrev_b64 %11, 3, %11\n")

static int g_waves = 6, g_iters = 100000, g_cus = 256;

template <typename K>
void run(const char *name, K kern, double vinst_per_group = 8.0) {
    const int grid = g_cus * g_waves;
    uint32_t *d;
    unsigned long long *clk;
    (void)hipMalloc(&d, (size_t)grid * 256 * 4);
    (void)hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, clk, 1u, g_iters);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, clk, 2u, g_iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h((size_t)grid * 2);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz, tk;
    for (int i = 0; i < grid; ++i)
        if (h[2 * i + 1]) {
            mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
            tk.push_back((double)h[2 * i]);
        }
    std::sort(mhz.begin(), mhz.end());
    std::sort(tk.begin(), tk.end());
    const double f = mhz[mhz.size() / 2];
    const double per_simd = (double)g_iters * 4.0 * vinst_per_group * g_waves;
    printf("%-28s %8.2f ms  clock %6.0f MHz  %6.2f cycles per vector instruction per SIMD (event time x measured clock)   block ticks / event ticks %.2f\n", name, ms, f,
           ms * 1e-3 * f * 1e6 / per_simd, tk[tk.size() / 2] / (ms * 1e-3 * f * 1e6));
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
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_add), 256, 0);
    printf("# %d CUs, %d waves per SIMD asked, occupancy API: %d blocks of 256 per CU, %d iterations x 4 groups\n", g_cus, g_waves, nb, g_iters);
    run("v_add_u32 (control)", k_add);
    run("v_cndmask e32 vcc x8", k_cnd_vcc);
    run("v_cndmask e64 vcc x8", k_cnd_e64_vcc);
    run("v_cndmask e64 sgpr-pair x8", k_cnd_sg);
    run("cndmask vcc + add (x4 each)", k_cnd_vcc_add);
    run("tree as compiled: 4 e32 + 3 e64 + add", k_tree_compiler);
    run("tree all e64 + add", k_tree_e64);
    run("tree e32 / e64 interleaved + add", k_tree_interleaved);
    run("cnd32 x2, add x2", k_cnd2_add2);
    if (getenv("FK_RUN_LENGTH_ONLY")) {
        run("add x8 (control)", k_add);
        run("cnd64 x8 (control)", k_cnd_sg);
        run("cnd32 run 1 + add x7", k_run1);
        run("cnd32 run 2 + add x6", k_run2);
        run("cnd32 run 3 + add x5", k_run3);
        run("cnd32 run 4 + add x4", k_run4);
        run("cnd32 run 6 + add x2", k_run6);
        run("cnd32 run 8", k_cnd_vcc);
        run("cnd32 every 2nd (4 + 4 add)", k_every2);
        run("cnd32 every 4th (2 + 6 add)", k_every4);
        run("cnd64 every 2nd (4 + 4 add)", k_every2_e64);
        run("cnd64 run 4 + add x4", k_run4_e64);
        run("s_mov vcc; cnd32 run 4 + add x4", k_smov_run4);
        run("s_mov vcc; cnd32 every 2nd", k_smov_every2);
        run("cnd32 / v_mad_u64_u32 alternating", k_every2_mad);
        return 0;
    }
    run("cnd32 + s_nop 0 (x8)", k_cnd_vcc_nop);
    run("cnd32 x8, independent sources", k_cnd_vcc_src);
    run("v_addc_co e64 sgpr carry", k_addc_e64);
    run("v_addc_co e32 vcc chain", k_addc_e32);
    run("v_add_co e32", k_addco_e32);
    run("v_add + s_nop 0", k_add_nop0);
    run("v_add + s_nop 1", k_add_nop1);
    run("v_mov literal", k_mov_lit);
    run("v_and literal", k_and_lit);
    run("v_cmp e64 -> sgpr pair", k_cmp_sg);
    run("v_cmp e32 -> vcc", k_cmp_vcc);
    run("v_bitop3_b32", k_bitop3);
    run("v_xad_u32", k_xad);
    run("v_med3_i32", k_med3);
    run("v_subrev_u32", k_subrev);
    run("v_lshlrev_b32 by 1", k_lshl1);
    run("v_ashrrev_i32", k_ashr);
    run("v_not_b32", k_not);
    run("v_max_u32", k_max);
    run("v_mad_u64_u32 carry->vcc", k_mad64_vcc);
    run("v_mad_u64_u32 carry->sgpr", k_mad64_sg);
    run("v_mad_u64_u32 addend 0", k_mad64_zero);
    run("v_lshl_add_u64", k_lshl_add_u64);
    run("v_mov_b64", k_mov_b64);
    run("v_lshlrev_b64", k_lshlrev_b64);
    return 0;
}
