// pcg_hand.h — the PCG64DXSM draw laid out by hand for gfx950 (round 5): an EXPERIMENT, measured and not shipped.
// Included by csrc/fk_device.h only under -DFK_PCG_HAND (tools/ab builds) and by tools/pcg_bench.hip.  Results (profiles/r05_pcg_draw_bench.txt,
// profiles/r05_ab_pcg_hand.log): 25 instead of 31 vector instructions per draw, v_mov_b32 per roll-loop trip 31 -> 15, 261 -> 245 vector
// instructions on the k = 2 kernel's hot path — and the kernel is 2.6 % SLOWER (9.57 against 9.33 ms per 10^7 games), the draw alone 2.5 %
// faster (101.6 against 104.2 cycles at six waves per SIMD; 108.8 with the exec narrowing of pcg_draws): the moves it removes are
// full-rate instructions (2.4 cycles per wave-instruction per SIMD), what replaces them (v_mul_hi_u32, v_addc_co_u32 on SGPR carries) is
// half-rate (4.3), as profiles/r05_valu_issue_rates.txt prices them.  Instruction COUNT is not the metric on this chip; issue cost is.
#pragma once

namespace fk {

// ---- one PCG64DXSM draw laid out by hand for gfx950 (round 5) ----
// state' = state * M + inc (mod 2^128), M = M1:M0 the 64-bit cheap multiplier, state = s3:s2:s1:s0, inc = i3:i2:i1:i0.
// The compiler's lowering of the 128-bit expression walks the 32-bit columns and feeds the HIGH word of one v_mad_u64_u32 as
// the zero-extended LOW word of the next one's addend; gfx950 wants 64-bit operands in even-aligned register pairs, so every
// such hand-over is two v_mov_b32 (six per draw in the ISA of round 4: 31 vector instructions per draw, and the predicated
// regions of a roll merged their 64-bit state through more copies).  Grouping the partial products by the PARITY of their
// 32-bit position keeps every addend a whole, already aligned pair:
//     position 0:   A  = s0 M0 + i1:i0                      carry cA (worth 2^64)
//     position 32:  X  = s0 M1 + (s1 M0 + cA << 32)         carry cB (worth 2^96); s1 M0 <= 2^64 - 2^33 + 1, so + 2^32 cannot carry
//     position 64:  Y  = s1 M1 + s2 M0 + i3:i2              (carries leave the 128 bits)
//     position 96:  T  = s2 M1 + s3 M0 + cB                 (low words; s3 M0 is also a term of the DXSM output)
//     s0' = A0   s1' = A1 + X0 (c1)   s2' = Y0 + X1 + c1 (c2)   s3' = Y1 + T + c2
// = 5 v_mad_u64_u32 + 1 v_mul_lo_u32 + 5 add-with-carry + 1 move; the DXSM output of the old state (2 xor, 1 or, 1 v_mad_u64_u32,
// 5 v_mul_lo_u32 / v_mul_hi_u32, 2 v_add3) is interleaved with it: 25 vector instructions per draw, the state words updated in
// place (the predicated regions of a roll merge nothing) and the draw's high word also left in `last_hi` (the next roll's
// buffered half word).  One asm block, because the halves of the 64-bit temporaries must be named: they live in v[16:27],
// declared as clobbers.  The carries are SGPR lane masks; on gfx950 a VALU read of an SGPR / VCC written by a VALU instruction
// needs two wait states in between, which inline asm must provide itself (the compiler's hazard recogniser does not look
// inside): independent instructions where there are any, s_nop otherwise.
// One draw as asm text: state words %0..%3 (updated in place), output words LO / HI, last_hi %10, carries %11 / %12, inc_lo %15, inc_hi %16,
// M0 %17, M1 %18.  Temporaries: v[16:17] = A, v[18:19] = X, v[20:21] = Y, v[22:23] = P (then h0, p1 l0), v24 = s3 M0, v25 = t, t M1, l0, s2 M1, T,
// v26 = h0 s1, v27 = high word of h0 l0.
#define FK_PCG_DRAW(LO, HI)                                                                                                        \
    "v_mul_lo_u32 v24, %3, %17\n\t"                      /* g = s3 M0 */                                                           \
    "v_xor_b32_e32 v25, %2, %3\n\t"                      /* t = low word of h ^ h >> 32 */                                         \
    "v_mad_u64_u32 v[22:23], vcc, v25, %17, 0\n\t"       /* P = t M0 */                                                            \
    "v_mul_lo_u32 v25, v25, %18\n\t"                     /* t M1 */                                                                \
    "v_mad_u64_u32 v[16:17], %11, %0, %17, %15\n\t"      /* A = s0 M0 + inc_lo -> cA */                                            \
    "v_add3_u32 v23, v23, v25, v24\n\t"                  /* p1 = high word of h M */                                               \
    "v_mad_u64_u32 v[18:19], vcc, %1, %17, 0\n\t"        /* s1 M0 */                                                               \
    "v_xor_b32_sdwa v22, v23, v22 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" /* h0 = P0 ^ p1 >> 16 */ \
    "v_or_b32_e32 v25, 1, %0\n\t"                        /* l0 = s0 | 1 */                                                         \
    "v_addc_co_u32_e64 v19, vcc, v19, 0, %11\n\t"        /* + cA << 32 (cA was written four instructions ago) */                   \
    "v_mul_lo_u32 v23, v23, v25\n\t"                     /* p1 l0 */                                                               \
    "v_mad_u64_u32 v[18:19], %12, %0, %18, v[18:19]\n\t" /* X = s0 M1 + ... -> cB */                                               \
    "v_mul_lo_u32 v26, v22, %1\n\t"                      /* h0 s1 */                                                               \
    "v_mad_u64_u32 v[20:21], vcc, %2, %17, %16\n\t"      /* s2 M0 + inc_hi */                                                      \
    "v_mul_hi_u32 v27, v22, v25\n\t"                     /* high word of h0 l0 */                                                  \
    "v_mad_u64_u32 v[20:21], vcc, %1, %18, v[20:21]\n\t" /* Y */                                                                   \
    "v_mul_lo_u32 " LO ", v22, v25\n\t"                  /* low output word */                                                     \
    "v_mul_lo_u32 v25, %2, %18\n\t"                      /* s2 M1 */                                                               \
    "v_add3_u32 " HI ", v27, v26, v23\n\t"               /* high output word */                                                    \
    "v_addc_co_u32_e64 v25, vcc, v25, v24, %12\n\t"      /* T = s2 M1 + s3 M0 + cB */                                              \
    "v_mov_b32_e32 %0, v16\n\t"                          /* s0' = A0 */                                                            \
    "v_add_co_u32_e32 %1, vcc, v17, v18\n\t"             /* s1' = A1 + X0 */                                                       \
    "v_mov_b32_e32 %10, " HI "\n\t"                      /* last_hi (one of the two wait states behind the carry) */               \
    "s_nop 0\n\t"                                                                                                                  \
    "v_addc_co_u32_e32 %2, vcc, v20, v19, vcc\n\t"       /* s2' = Y0 + X1 + c1 */                                                  \
    "s_nop 1\n\t"                                                                                                                  \
    "v_addc_co_u32_e32 %3, vcc, v21, v25, vcc\n\t"       /* s3' = Y1 + T + c2 */

// The `need` (0 .. 3) draws of one roll: draw i runs for the lanes with need > i (the exec mask is narrowed step by step and restored at
// the end; a step that no lane needs is branched over).  Words of draws that did not run: 1 (never a Lemire rejection) for the
// second and third draw, undefined for the first (it is skipped only by a one-die roll that has a buffered word, which reads none
// of them; what the rejection test may see in them is some older word — a detour at most).
__device__ inline void pcg_draws(uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3, uint64_t inc_lo, uint64_t inc_hi, uint32_t need,
                                 uint32_t &lo0, uint32_t &hi0, uint32_t &lo1, uint32_t &hi1, uint32_t &lo2, uint32_t &hi2, uint32_t &last_hi) {
    constexpr uint32_t M0 = (uint32_t)PCG_CHEAP_MULT, M1 = (uint32_t)(PCG_CHEAP_MULT >> 32);
    uint64_t cA, cB, ex;
    asm("s_mov_b64 %13, exec\n\t"
        "v_mov_b32_e32 %6, 1\n\t"
        "v_mov_b32_e32 %7, 1\n\t"
        "v_mov_b32_e32 %8, 1\n\t"
        "v_mov_b32_e32 %9, 1\n\t"
        "v_cmp_lt_u32_e32 vcc, 0, %14\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lfk_pcg_done_%=\n\t"
        FK_PCG_DRAW("%4", "%5")
        "v_cmp_lt_u32_e32 vcc, 1, %14\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lfk_pcg_done_%=\n\t"
        FK_PCG_DRAW("%6", "%7")
        "v_cmp_lt_u32_e32 vcc, 2, %14\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lfk_pcg_done_%=\n\t"
        FK_PCG_DRAW("%8", "%9")
        ".Lfk_pcg_done_%=:\n\t"
        "s_mov_b64 exec, %13"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(lo0), "=&v"(hi0), "=&v"(lo1), "=&v"(hi1), "=&v"(lo2), "=&v"(hi2), "+v"(last_hi),
          "=&s"(cA), "=&s"(cB), "=&s"(ex)
        : "v"(need), "v"(inc_lo), "v"(inc_hi), "s"(M0), "s"(M1)
        : "vcc", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
}

__device__ inline uint64_t pcg_next64(Rng &r) { // a single draw (seeding-time and replay paths)
    constexpr uint32_t M0 = (uint32_t)PCG_CHEAP_MULT, M1 = (uint32_t)(PCG_CHEAP_MULT >> 32);
    uint32_t s0 = (uint32_t)r.lo, s1 = (uint32_t)(r.lo >> 32), s2 = (uint32_t)r.hi, s3 = (uint32_t)(r.hi >> 32), lo, hi, d6, d7, d8, d9, last;
    uint64_t cA, cB, d13;
    uint32_t d14 = 0;
    asm(FK_PCG_DRAW("%4", "%5") "s_nop 0"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(lo), "=&v"(hi), "=&v"(d6), "=&v"(d7), "=&v"(d8), "=&v"(d9), "=&v"(last), "=&s"(cA),
          "=&s"(cB), "=&s"(d13)
        : "v"(d14), "v"(r.inc_lo), "v"(r.inc_hi), "s"(M0), "s"(M1)
        : "vcc", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    r.lo = (uint64_t)s0 | ((uint64_t)s1 << 32);
    r.hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

} // namespace fk
