// fk_perm_wave.h — the accepted Fisher-Yates draws of a shuffle produced by a whole WAVE (round 6).
//
// fk_perm_draw_kernel (fk_kernels.h) gives a shuffle to one thread: S - 1 accepted draws cost ~3 500 dependent generator steps, 1.03 ms at
// S = 5 160 whatever the launch — a quarter of the game kernels' time in the 256-MB launch groups of rows mode (900 shuffles: 15 waves on
// a 1 024-SIMD chip).  Throughput-optimal for tens of thousands of shuffles, latency-bound below.  Here lane L of a wave holds the
// generator state L steps ahead and every round jumps all lanes by 64 steps at once:
//
//     state_{n+64} = M^64 * state_n + inc * (M^63 + ... + M + 1)        (mod 2^128; M = the cheap multiplier of PCG64DXSM)
//
// so a round yields 64 consecutive 64-bit outputs = 128 consecutive 32-bit words of the buffered stream (low half first:
// Generator._shuffle_raw -> random_interval -> masked rejection over next_uint32).  What stays sequential is the rejection scan: word t is
// masked with the bit length of the CURRENT bound i and accepted iff the result is <= i, and i falls by one per accepted word.  A lane
// needs the number of words accepted BEFORE its two in the round; the wave finds it by iterating from "none accepted": pass p is right for at
// least the first p words (a word's decision depends on earlier words only), and since a decision only flips when the masked word falls in
// the few values between two estimates of i, two or three passes settle all 128 (the loop runs until nothing changes, so the result is
// the sequential one whatever the data).  Output format = fk_perm_draw_kernel's, bit for bit: groups of eight 16-bit draws, the last
// partial group right-aligned behind the previous draws (what its shifting queue leaves there).
#pragma once

#include "fk_device.h"

constexpr uint64_t PCG_JUMP64_MULT_HI = 0xd57169fe14f36c3bULL, PCG_JUMP64_MULT_LO = 0xfa678f9d12df3b01ULL; // M^64 mod 2^128
constexpr uint64_t PCG_JUMP64_GEO_HI = 0x5bdc97fab0939c54ULL, PCG_JUMP64_GEO_LO = 0x11de56ce690501c0ULL;   // M^63 + ... + M + 1 mod 2^128
constexpr int WAVE_DRAW_BLOCK = 256;        // four shuffles per workgroup
constexpr uint32_t WAVE_DRAW_MAX_SH = 32768; // beyond this many shuffles the thread-per-shuffle kernel fills the chip and wins on throughput

__global__ __launch_bounds__(WAVE_DRAW_BLOCK) void fk_perm_draw_wave_kernel(SeedPool prefix, uint64_t shuffle0, uint32_t n_sh, uint32_t S,
                                                                           uint32_t g_stride, uint32_t sh_stride, uint4 *draws) {
    typedef unsigned __int128 u128;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sh = blockIdx.x * (WAVE_DRAW_BLOCK / 64) + (threadIdx.x >> 6);
    if (sh >= n_sh || S < 2u) return; // (the whole wave: sh is wave-uniform)
    Rng r{};
    {
        SeedPool p = prefix;
        p.hc = HC_AFTER_6_WORDS;
        ss_absorb64(p, shuffle0 + sh); // shuffle_index
#pragma unroll
        for (int w = 0; w < 5; ++w) ss_absorb64(p, 0); // pair_id, order, game_index, seat_index, replicate_index
        uint32_t g8[8];
        ss_generate<8>(p, g8);
        pcg_seed(r, g8);
    }
    const u128 inc = ((u128)r.inc_hi << 64) | r.inc_lo;
    u128 st = ((u128)r.hi << 64) | r.lo;
    for (uint32_t s = 0; s < 63u; ++s) // lane L: L steps ahead
        if (s < lane) st = st * PCG_CHEAP_MULT + inc;
    const u128 jump_mult = ((u128)PCG_JUMP64_MULT_HI << 64) | PCG_JUMP64_MULT_LO;
    const u128 jump_add = inc * (((u128)PCG_JUMP64_GEO_HI << 64) | PCG_JUMP64_GEO_LO);

    const uint32_t N = S - 1u, full = N & ~7u, tail = N & 7u;
    uint16_t *const row = reinterpret_cast<uint16_t *>(draws + (size_t)sh * sh_stride);
    const size_t group_u16 = (size_t)g_stride * 8u; // u16 elements between two groups of the shuffle
    const uint64_t below = ((uint64_t)1 << lane) - 1u;
    uint32_t i = N; // the bound of the next draw (wave-uniform); N - i draws are accepted
    while (i >= 1u) {
        uint64_t h = (uint64_t)(st >> 64), l = (uint64_t)st | 1u; // DXSM output of the current state (fk_device.h pcg_next64_plain)
        h ^= h >> 32;
        h *= PCG_CHEAP_MULT;
        h ^= h >> 48;
        h *= l;
        st = st * jump_mult + jump_add;
        const uint32_t w_lo = (uint32_t)h, w_hi = (uint32_t)(h >> 32);
        // words 2 lane (low half) and 2 lane + 1 (high half) of the round
        uint64_t acc_lo = 0, acc_hi = 0; // ballots of the current estimate
        uint32_t j_lo = 0, j_hi = 0, c_lo = 0;
        bool a_lo = false, a_hi = false;
        for (;;) {
            c_lo = (uint32_t)__popcll(acc_lo & below) + (uint32_t)__popcll(acc_hi & below); // accepted before this lane's low word
            const uint32_t i_lo = i > c_lo ? i - c_lo : 0u;
            j_lo = i_lo ? (w_lo & (0xffffffffu >> __clz((int)i_lo))) : 0u;
            a_lo = i_lo >= 1u && j_lo <= i_lo;
            const uint32_t c_hi = c_lo + (a_lo ? 1u : 0u);
            const uint32_t i_hi = i > c_hi ? i - c_hi : 0u;
            j_hi = i_hi ? (w_hi & (0xffffffffu >> __clz((int)i_hi))) : 0u;
            a_hi = i_hi >= 1u && j_hi <= i_hi;
            const uint64_t n_lo = __ballot(a_lo), n_hi = __ballot(a_hi);
            if (n_lo == acc_lo && n_hi == acc_hi) break; // a fixed point of a causal recurrence: the sequential result
            acc_lo = n_lo;
            acc_hi = n_hi;
        }
        const uint32_t done = N - i; // draws accepted before this round
        auto put = [&](uint32_t p, uint32_t j) { // draw number p of the shuffle
            if (p < full) row[(size_t)(p >> 3) * group_u16 + (p & 7u)] = (uint16_t)j;
            if (tail && p + 8u >= N) { // among the last eight: also (or only) in the right-aligned last group
                const uint32_t slot = p + 8u - N;
                row[(size_t)(full >> 3) * group_u16 + slot] = (uint16_t)j;
            }
        };
        if (a_lo) put(done + c_lo, j_lo);
        if (a_hi) put(done + c_lo + (a_lo ? 1u : 0u), j_hi);
        const uint32_t accepted = (uint32_t)__popcll(acc_lo) + (uint32_t)__popcll(acc_hi);
        i -= accepted; // (accepted <= i: a word is only accepted while its own bound is >= 1)
    }
}
