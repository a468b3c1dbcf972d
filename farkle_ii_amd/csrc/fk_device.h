// fk_device.h — device-side building blocks of the gfx950 Farkle engine.
//
// Everything here is integer VALU work written for 64-wide wavefronts: one lane = one game,
// branch-light (predicated) code so that the lanes of a wave stay converged on the roll loop.
// Reference semantics are cited as path:line under the reference's root; NumPy's generators
// (third party, numpy>=1.26) are restated from their published algorithms.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fk {

// ----------------------------------------------------------------------------------------
// NumPy SeedSequence (seed_seq_fe128) — called by src/farkle/utils/random.py:156
// ----------------------------------------------------------------------------------------
constexpr uint32_t SS_INIT_A = 0x43b0d7e5u, SS_MULT_A = 0x931e8875u;
constexpr uint32_t SS_INIT_B = 0x8b51f9ddu, SS_MULT_B = 0x58f38dedu;
constexpr uint32_t SS_MIX_L = 0xca01f9ddu, SS_MIX_R = 0x4973f715u;

// hash constant in front of the c-th hashmix call (0-based): INIT_A * MULT_A^c
__host__ __device__ constexpr uint32_t ss_hc(int c) {
    uint32_t h = SS_INIT_A;
    for (int i = 0; i < c; ++i) h *= SS_MULT_A;
    return h;
}

__host__ __device__ inline uint32_t ss_hashmix(uint32_t v, uint32_t &hc) {
    v ^= hc;
    hc *= SS_MULT_A;
    v *= hc;
    v ^= v >> 16;
    return v;
}

__host__ __device__ inline uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> 16;
    return r;
}

// Pool state while entropy words are being absorbed.  `hc` is the running hash constant.
struct SeedPool {
    uint32_t p[4];
    uint32_t hc;
};

// Absorb the first four entropy words and run the all-pairs mixing round.
__host__ __device__ inline void ss_begin(SeedPool &s, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    s.hc = SS_INIT_A;
    s.p[0] = ss_hashmix(w0, s.hc);
    s.p[1] = ss_hashmix(w1, s.hc);
    s.p[2] = ss_hashmix(w2, s.hc);
    s.p[3] = ss_hashmix(w3, s.hc);
#pragma unroll
    for (int src = 0; src < 4; ++src) {
#pragma unroll
        for (int dst = 0; dst < 4; ++dst) {
            if (src != dst) s.p[dst] = ss_mix(s.p[dst], ss_hashmix(s.p[src], s.hc));
        }
    }
}

// Absorb one further entropy word (index >= 4) into all four pool slots.
__host__ __device__ inline void ss_absorb(SeedPool &s, uint32_t w) {
#pragma unroll
    for (int dst = 0; dst < 4; ++dst) s.p[dst] = ss_mix(s.p[dst], ss_hashmix(w, s.hc));
}

__host__ __device__ inline void ss_absorb64(SeedPool &s, uint64_t v) {
    ss_absorb(s, (uint32_t)v);         // low word first (random.py:62)
    ss_absorb(s, (uint32_t)(v >> 32));
}

// generate_state: n 32-bit words cycling over the pool
template <int N>
__host__ __device__ inline void ss_generate(const SeedPool &s, uint32_t (&out)[N]) {
    uint32_t hc = SS_INIT_B;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t v = s.p[i & 3];
        v ^= hc;
        hc *= SS_MULT_B;
        v *= hc;
        v ^= v >> 16;
        out[i] = v;
    }
}

// ----------------------------------------------------------------------------------------
// NumPy PCG64DXSM — constructed at src/farkle/utils/random.py:188
// ----------------------------------------------------------------------------------------
constexpr uint64_t PCG_CHEAP_MULT = 0xda942042e4dd58b5ULL;
constexpr uint64_t PCG_DEF_MULT_HI = 2549297995355413924ULL, PCG_DEF_MULT_LO = 4865540595714422341ULL;

struct Rng {
    uint64_t hi, lo;         // 128-bit LCG state
    uint64_t inc_hi, inc_lo; // odd increment
    uint32_t buf;            // buffered high half of the last 64-bit output
    uint32_t has_buf;        // 0/1
};

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// state = state * M(128 bit) + inc   (seeding uses the DEFAULT multiplier twice)
__host__ __device__ inline void pcg_step_default(uint64_t &hi, uint64_t &lo, uint64_t inc_hi, uint64_t inc_lo) {
    uint64_t nlo = lo * PCG_DEF_MULT_LO;
    uint64_t nhi = mulhi64(lo, PCG_DEF_MULT_LO) + lo * PCG_DEF_MULT_HI + hi * PCG_DEF_MULT_LO;
    nlo += inc_lo;
    nhi += inc_hi + (nlo < inc_lo ? 1u : 0u);
    hi = nhi;
    lo = nlo;
}

__host__ __device__ inline void pcg_seed(Rng &r, const uint32_t (&g)[8]) {
    uint64_t s0 = (uint64_t)g[0] | ((uint64_t)g[1] << 32), s1 = (uint64_t)g[2] | ((uint64_t)g[3] << 32);
    uint64_t s2 = (uint64_t)g[4] | ((uint64_t)g[5] << 32), s3 = (uint64_t)g[6] | ((uint64_t)g[7] << 32);
    // initstate = (s0 << 64) | s1 ; initseq = (s2 << 64) | s3 ; inc = (initseq << 1) | 1
    r.inc_hi = (s2 << 1) | (s3 >> 63);
    r.inc_lo = (s3 << 1) | 1u;
    r.hi = 0;
    r.lo = 0;
    pcg_step_default(r.hi, r.lo, r.inc_hi, r.inc_lo);
    uint64_t nlo = r.lo + s1;
    r.hi = r.hi + s0 + (nlo < s1 ? 1u : 0u);
    r.lo = nlo;
    pcg_step_default(r.hi, r.lo, r.inc_hi, r.inc_lo);
    r.buf = 0;
    r.has_buf = 0;
}

// DXSM output of the current state, then the cheap-multiplier step.
__host__ __device__ inline uint64_t pcg_next64_plain(Rng &r) {
    uint64_t h = r.hi, l = r.lo | 1u;
    h ^= h >> 32;
    h *= PCG_CHEAP_MULT;
    h ^= h >> 48;
    h *= l;
    // one 128-bit multiply-add: the compiler's lowering (5 v_mad_u64_u32 + a 4-word carry chain) is 6 VALU
    // instructions shorter than the lo/mulhi/hi split with its explicit carry compare
    typedef unsigned __int128 u128;
    const u128 st = (((u128)r.hi << 64) | r.lo) * PCG_CHEAP_MULT + (((u128)r.inc_hi << 64) | r.inc_lo);
    r.hi = (uint64_t)(st >> 64);
    r.lo = (uint64_t)st;
    return h;
}

__host__ __device__ inline uint64_t pcg_next64(Rng &r) { return pcg_next64_plain(r); }

// buffered 32-bit draw: low half first, high half on the next call (persists across rolls)
__host__ __device__ inline uint32_t pcg_next32(Rng &r) {
    if (r.has_buf) {
        r.has_buf = 0;
        return r.buf;
    }
    uint64_t o = pcg_next64(r);
    r.has_buf = 1;
    r.buf = (uint32_t)(o >> 32);
    return (uint32_t)o;
}

// ----------------------------------------------------------------------------------------
// Dice: FarklePlayer._roll (src/farkle/game/engine.py:85-101) -> Generator.integers(1, 7, size=n)
// = per die Lemire's bounded draw on the buffered 32-bit stream; rejection iff low32(r*6) < 4.
// Returns the roll as nibble-packed face counts: count(face f) at bits [4(f-1), 4(f-1)+3).
// ----------------------------------------------------------------------------------------
template <uint32_t STRIDE = 4>
__host__ __device__ inline uint32_t roll_counts_sequential(Rng &r, uint32_t n, uint32_t *faces_out) {
    uint32_t counts = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t m = (uint64_t)pcg_next32(r) * 6u;
        uint32_t left = (uint32_t)m;
        if (left < 6u) {
            while (left < 4u) { // threshold = (2^32 - 6) % 6 = 4
                m = (uint64_t)pcg_next32(r) * 6u;
                left = (uint32_t)m;
            }
        }
        uint32_t f = (uint32_t)(m >> 32); // face - 1
        counts += 1u << (STRIDE * f);
        if (faces_out) *faces_out |= (f + 1u) << (4u * i);
    }
    return counts;
}

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Converged fast path: all 64-bit outputs a roll can need (<= 3) are generated under per-lane
// predicates, the six candidate words are selected with v_cndmask, and the (once in ~2^30 dice)
// Lemire rejection falls back to the sequential form from the saved generator state.
//   face index:   4*f = mulhi(w, 24) & 28          (mulhi(w,24) = floor(4 * 6w / 2^32) in [4f, 4f+3])
//   count update: counts += on_i << 4f             (one v_lshl_add_u32; on_i = bit i of (1<<n)-1)
//   rejection:    6w mod 2^32 < 4 implies 24w mod 2^32 < 16, so the test is min over the six low words of
//                 24w < 16 (the low half of the same 64-bit product).  False positives (6w mod 2^30 < 4) and
//                 words that are generated but not consumed only cause a harmless, exact detour through the
//                 sequential path; words that are not generated are the constant 1 (24 >= 16, never detours).
// STRIDE = width of one face's count field in the result: 4 (nibbles, the SWAR scorer's input) or 3 (the 18-bit key
// of the score table, SCORE_LUT below; a count is at most 6).
// roll_counts_fast: the converged part.  `detour` = some low product word fell below the rejection bound: the caller restores the
// generator it started from and replays the roll with roll_counts_sequential (roll_counts below does both; the game kernels
// restore from the seat record in LDS instead of keeping a copy of the state in registers).
template <uint32_t STRIDE = 4>
__device__ inline uint32_t roll_counts_fast(Rng &r, uint32_t n, bool &detour, uint32_t *faces_out = nullptr) {
    const uint32_t hb = r.has_buf, buf_in = r.buf;
    const uint32_t need = (n - hb + 1u) >> 1; // new 64-bit outputs: ceil((n - has_buf) / 2), 0..3
    uint32_t lo0 = 1, hi0 = 1, lo1 = 1, hi1 = 1, lo2 = 1, hi2 = 1, last_hi = r.buf;
    if (need > 0u) {
        uint64_t o = pcg_next64_plain(r);
        lo0 = (uint32_t)o;
        hi0 = (uint32_t)(o >> 32);
        last_hi = hi0;
    }
    if (need > 1u) {
        uint64_t o = pcg_next64_plain(r);
        lo1 = (uint32_t)o;
        hi1 = (uint32_t)(o >> 32);
        last_hi = hi1;
    }
    if (need > 2u) {
        uint64_t o = pcg_next64_plain(r);
        lo2 = (uint32_t)o;
        hi2 = (uint32_t)(o >> 32);
        last_hi = hi2;
    }
    uint32_t w[6];
    w[0] = hb ? buf_in : lo0;
    w[1] = hb ? lo0 : hi0;
    w[2] = hb ? hi0 : lo1;
    w[3] = hb ? lo1 : hi1;
    w[4] = hb ? hi1 : lo2;
    w[5] = hb ? lo2 : hi2;
    const uint32_t onbits = (1u << n) - 1u;
    uint32_t counts = 0, faces = 0, minleft = 0xffffffffu;
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) {
        // one v_mad_u64_u32 per die: field shift from the high word, rejection test on the low word.
        //   STRIDE 4: 24w -> 4*face = hi & 28, low word 4 * (6w mod 2^30);   STRIDE 3: 6w -> face = hi, 3*face = hi + 2*hi
        const uint64_t pw = (uint64_t)w[i] * (STRIDE == 4u ? 24u : 6u);
        const uint32_t hiw = (uint32_t)(pw >> 32);
        const uint32_t fsh = (STRIDE == 4u) ? (hiw & 28u) : (hiw + 2u * hiw);
        const uint32_t face = (STRIDE == 4u) ? (fsh >> 2) : hiw;
        const uint32_t on = (onbits >> i) & 1u;
        counts += on << fsh;
        const uint32_t left = (uint32_t)pw;
        minleft = left < minleft ? left : minleft;
        if (faces_out) faces |= on ? ((face + 1u) << (4u * i)) : 0u;
    }
    r.has_buf = (n + hb) & 1u;
    r.buf = last_hi;
    detour = minleft < (STRIDE == 4u ? 16u : 4u); // rare (STRIDE 4: a superset of the rejections)
#ifdef FK_FORCE_DETOUR
    // Test builds only (tests/test_detour_gpu.py): a real rejection happens once in ~2^30 dice, so the kernels' detour — generator re-read
    // from the seat record / hot planes, the roll replayed sequentially — would never run in a parity test.  Here every
    // FK_FORCE_DETOUR-th roll (a power of two; picked by bits of the advanced state, so lanes of a wave disagree) takes it; the detour is
    // exact whether or not a word was rejected, so every result must stay bit-identical.
    detour = detour || ((((uint32_t)(r.lo >> 17)) & (uint32_t)(FK_FORCE_DETOUR - 1)) == 0u);
#endif
    if (faces_out) *faces_out = faces;
    return counts;
}

template <uint32_t STRIDE = 4>
__device__ inline uint32_t roll_counts(Rng &r, uint32_t n, uint32_t *faces_out = nullptr) {
    const Rng saved = r;
    bool detour;
    uint32_t faces = 0;
    uint32_t counts = roll_counts_fast<STRIDE>(r, n, detour, faces_out ? &faces : nullptr);
    if (detour) { // redo this roll exactly as NumPy would
        r = saved;
        faces = 0;
        counts = roll_counts_sequential<STRIDE>(r, n, &faces);
    }
    if (faces_out) *faces_out = faces;
    return counts;
}

// ----------------------------------------------------------------------------------------
// Scoring: _evaluate_nb (src/farkle/game/scoring_lookup.py:123-172) on nibble-packed counts (the generator of the
// score table the kernels read, and the host-side statement of the rule).
// With <= 6 dice at most ONE face can form a set outside the four 6-dice patterns, so the
// set is found with one SWAR compare + ffs instead of a per-face loop.
// ----------------------------------------------------------------------------------------
struct RawScore {
    int32_t score, used, sf, so; // points, dice used, lone fives, lone ones
};

__host__ __device__ inline uint32_t nibble_eq(uint32_t c, uint32_t v) {
    uint32_t x = c ^ (v * 0x111111u);
    return ~(x | (x >> 1) | (x >> 2)) & 0x111111u; // exact per-nibble flags (counts < 8)
}

__host__ __device__ inline int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

__host__ __device__ inline int ctz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}

__host__ __device__ inline RawScore score_counts(uint32_t c) {
    // straight-line: every conditional below is a select, not a branch
    const uint32_t e2 = nibble_eq(c, 2u), e3 = nibble_eq(c, 3u), e4 = nibble_eq(c, 4u);
    const uint32_t straight = (c == 0x111111u) ? 1u : 0u;                  // _straight :28-38
    const uint32_t three_pairs = (popc32(e2) == 3) ? 1u : 0u;              // _three_pairs :42-53
    const uint32_t two_triplets = (popc32(e3) == 2) ? 1u : 0u;             // _two_triplets :57-68
    const uint32_t four_two = ((e4 != 0u) ? 1u : 0u) & ((e2 != 0u) ? 1u : 0u); // _four_kind_plus_pair :72-82
    const uint32_t special = straight | three_pairs | two_triplets | four_two; // mutually exclusive (6 dice)
    // n-of-a-kind (_apply_sets :86-115): nibble >= 3  <=>  bit 3 of (nibble + 5); at most one such face here
    const uint32_t ge3 = (c + 0x555555u) & 0x888888u;
    const bool has_set = ge3 != 0u;
    const uint32_t sh = has_set ? (uint32_t)(ctz32(ge3 | 0x80000000u) - 3) : 0u; // 4 * face_index
    const int32_t n = (int32_t)((c >> sh) & 7u);
    const int32_t face = (int32_t)(sh >> 2) + 1;
    const int32_t trip = (face == 1) ? 300 : face * 100;
    const int32_t set_pts = has_set ? ((n == 3) ? trip : (n - 3) * 1000) : 0;
    const int32_t set_n = has_set ? n : 0;
    const uint32_t rest = has_set ? (c & ~(0xFu << sh)) : c;
    const int32_t ones = (int32_t)(rest & 7u), fives = (int32_t)((rest >> 16) & 7u);
    RawScore r;
    r.score = special ? (two_triplets ? 2500 : 1500) : (set_pts + 100 * ones + 50 * fives); // :168-172
    r.used = special ? 6 : (set_n + ones + fives);
    r.sf = special ? 0 : fives;
    r.so = special ? 0 : ones;
    return r;
}

// ----------------------------------------------------------------------------------------
// Score table.  _evaluate_nb depends on the count multiset only (923 multisets of <= 6 dice), so the game kernel
// reads it from a table built once per context by running score_counts() above on every key:
//   key   = six 3-bit face counts (what roll_counts<3> accumulates), 18 bits
//   entry = u16: [5:0] score / 50, [8:6] dice used, [11:9] lone fives, [14:12] lone ones, [15] all dice used
// 512 KiB in HBM, L2/L1-resident; ~10 VALU + one load per roll instead of the ~45 VALU of the SWAR scorer.
// ----------------------------------------------------------------------------------------
constexpr uint32_t SCORE_LUT_KEYS = 1u << 18;

__host__ __device__ inline uint32_t lut_key_to_nibbles(uint32_t key) {
    uint32_t c = 0;
    for (uint32_t f = 0; f < 6; ++f) c |= ((key >> (3u * f)) & 7u) << (4u * f);
    return c;
}

__host__ __device__ inline uint32_t nibbles_to_lut_key(uint32_t c) {
    uint32_t k = 0;
    for (uint32_t f = 0; f < 6; ++f) k |= ((c >> (4u * f)) & 7u) << (3u * f);
    return k;
}

// entry of one key; keys that are not a roll (a count of 7, or more than 6 dice) map to 0 and are never read
__host__ __device__ inline uint16_t score_lut_entry(uint32_t key) {
    uint32_t n = 0;
    for (uint32_t f = 0; f < 6; ++f) {
        const uint32_t cf = (key >> (3u * f)) & 7u;
        if (cf == 7u) return 0;
        n += cf;
    }
    if (n == 0u || n > 6u) return 0;
    const RawScore r = score_counts(lut_key_to_nibbles(key));
    return (uint16_t)((uint32_t)(r.score / 50) | ((uint32_t)r.used << 6) | ((uint32_t)r.sf << 9) | ((uint32_t)r.so << 12) |
                      (((uint32_t)r.used == n ? 1u : 0u) << 15));
}

// The kernels' score table holds 32-bit entries: the low half is the u16 entry above with bit 15 repurposed (no reader ever used "all dice
// used") as "the roll has lone 1s or 5s"; the high half is the ROLL's share of the discard table's key, already in place (DKEY_* below):
//   [17:16] min(lone fives, 2)   [19:18] min(lone ones, 2)   [28:26] min(score / 50, 7)
// so that the key of a roll is (entry >> 16) | (strategy bits & SF_DISCARD_KEY_BITS) | vmin << 4 | cmin << 7: nine full-rate and five
// half-rate vector instructions where round 4 assembled it from the decoded fields with 25 (profiles/r05_valu_issue_rates.txt prices them).
constexpr uint32_t SE_SINGLES = 1u << 15;

__host__ __device__ inline uint32_t score_lut_entry32(uint32_t key) {
    const uint32_t e = score_lut_entry(key);
    if (e == 0u) return 0u;
    const uint32_t raw50 = e & 63u, sf = (e >> 9) & 7u, so = (e >> 12) & 7u;
    const uint32_t keypart = (sf < 2u ? sf : 2u) | ((so < 2u ? so : 2u) << 2) | ((raw50 < 7u ? raw50 : 7u) << 10);
    return (e & 0x7fffu) | (((sf | so) != 0u) ? SE_SINGLES : 0u) | (keypart << 16);
}

__host__ __device__ inline RawScore raw_from_lut(uint32_t e) {
    RawScore r;
    r.score = (int32_t)((e & 63u) * 50u);
    r.used = (int32_t)((e >> 6) & 7u);
    r.sf = (int32_t)((e >> 9) & 7u);
    r.so = (int32_t)((e >> 12) & 7u);
    return r;
}

// ----------------------------------------------------------------------------------------
// Strategy record as the kernels see it: two dwords.
//   x = score_threshold (i32)
//   y = bits 0..7 dice_threshold (i8) | flag bits from 8
// ----------------------------------------------------------------------------------------
// (round 5: smart_one, require_both and favor_score sit at bits 13, 14, 15 — exactly where the discard table's key wants them, so the
// strategy's share of that key is `bits & SF_DISCARD_KEY_BITS`, one instruction per roll; see DKEY_* below)
enum : uint32_t {
    SF_SMART_FIVE = 1u << 8,
    SF_AUTO_HOT = 1u << 9,
    SF_CONSIDER_SCORE = 1u << 10,
    SF_CONSIDER_DICE = 1u << 11,
    SF_RUN_UP = 1u << 12,
    SF_SMART_ONE = 1u << 13,
    SF_REQUIRE_BOTH = 1u << 14,
    SF_FAVOR_SCORE = 1u << 15,
    SF_DISCARD_KEY_BITS = SF_SMART_ONE | SF_REQUIRE_BOTH | SF_FAVOR_SCORE
};

struct Strat {
    int32_t score_thr;
    uint32_t bits;
    __host__ __device__ int32_t dice_thr() const { return (int32_t)(int8_t)(bits & 0xffu); }
    __host__ __device__ bool has(uint32_t f) const { return (bits & f) != 0u; }
};

// _must_bank, src/farkle/game/scoring.py:283-300
__host__ __device__ inline bool must_bank(const Strat &s, int32_t score_after, int32_t dice_left_after) {
    const bool cs = s.has(SF_CONSIDER_SCORE), cd = s.has(SF_CONSIDER_DICE);
    const bool hit_score = cs && (score_after >= s.score_thr);
    const bool hit_dice = cd && (dice_left_after <= s.dice_thr());
    return (cs && cd && s.has(SF_REQUIRE_BOTH)) ? (hit_score && hit_dice) : (hit_score || hit_dice);
}

struct RollResult {
    int32_t score, used, d5, d1;
};

// default_score (src/farkle/game/scoring.py:618-693) = raw score + Smart-5/Smart-1 discard choice.
// The reference enumerates re-scored sub-rolls (generate_sequences/score_lister/_select_candidate,
// :197-366); every candidate that survives its `drop > singles` filter (:326-329) only removes lone
// 1s/5s, so its score is raw - 50*d5 - 100*d1 and its used dice raw_used - d5 - d1 (closed form).
//
// Reference form (loops), kept as the readable statement of the rule and used by host-side checks:
__host__ __device__ inline RollResult default_score_loops(uint32_t counts, int32_t n, int32_t turn_pre, const Strat &s) {
    const RawScore raw = score_counts(counts);
    RollResult out{raw.score, raw.used, 0, 0};
    if (!s.has(SF_SMART_FIVE) || raw.used == n || (raw.sf == 0 && raw.so == 0)) return out; // :433
    const int32_t max1 = s.has(SF_SMART_ONE) ? raw.so : 0;
    const bool favor_score = s.has(SF_FAVOR_SCORE);
    int32_t best_key = -1, b5 = 0, b1 = 0;
    for (int32_t d5 = 0; d5 <= raw.sf; ++d5) {       // outer loop over fives (:224)
        for (int32_t d1 = 0; d1 <= max1; ++d1) {     // inner loop over ones (:225)
            const int32_t cs = raw.score - 50 * d5 - 100 * d1;
            if (cs == 0) continue;                                    // score_lister :262
            const int32_t score_after = turn_pre + cs;                // :331
            const int32_t dice_left_after = n - (raw.used - d5 - d1); // :334
            if (must_bank(s, score_after, dice_left_after)) continue; // :337
            const int32_t key = favor_score ? (score_after * 8 + dice_left_after)
                                            : (dice_left_after * (1 << 20) + score_after);
            if (key > best_key) { // strict '>' keeps the first of equal keys (:354)
                best_key = key;
                b5 = d5;
                b1 = d1;
            }
        }
    }
    out.d5 = b5; // == single_fives - best_sf (:467)
    out.d1 = b1;
    out.score = raw.score - 50 * b5 - 100 * b1; // apply_discards :575-578
    out.used = raw.used - b5 - b1;
    return out;
}

// Branch-free form used by the kernels.  A candidate (d5, d1) != (0, 0) lives in one nibble of a 32-bit
// word (8 candidates, d5, d1 in 0..2); per-candidate predicates are SWAR compares whose result is bit 3
// of each nibble.  With v = d5 + 2*d1 (discarded points / 50) and cnt = d5 + d1 (dice returned):
//     score_after < score_thr      <=>  v   >= vmin = floor((pre + raw - thr) / 50) + 1   (0 if already below)
//     dice_left_after > dice_thr   <=>  cnt >= cmin = dice_thr - (n - used) + 1
//     not must_bank (:283-300)     <=>  OR rule:  v >= vmin AND cnt >= cmin;   AND rule (require_both): either
// The nibbles are laid out in preference order (best first), one layout per favor_dice_or_score value, so the
// argmax of the reference's tuple key (:346-359) is a find-first-set.  (0, 0) is handled on the side: it is the
// best candidate under SCORE preference and the worst under DICE preference.
constexpr uint32_t pack8(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
    return a0 | (a1 << 4) | (a2 << 8) | (a3 << 12) | (a4 << 16) | (a5 << 20) | (a6 << 24) | (a7 << 28);
}
// SCORE preference: ascending v, ties by descending cnt:  (1,0) (2,0) (0,1) (1,1) (2,1) (0,2) (1,2) (2,2)
constexpr uint32_t CS_D5 = pack8(1, 2, 0, 1, 2, 0, 1, 2), CS_D1 = pack8(0, 0, 1, 1, 1, 2, 2, 2);
// DICE preference: descending cnt, ties by ascending v:   (2,2) (2,1) (1,2) (2,0) (1,1) (0,2) (1,0) (0,1)
constexpr uint32_t CD_D5 = pack8(2, 2, 1, 2, 1, 0, 1, 0), CD_D1 = pack8(2, 1, 2, 0, 1, 2, 0, 1);
constexpr uint32_t NIB_H = 0x88888888u, NIB_1 = 0x11111111u;

// The inputs of the discard choice, reduced to the few values it really depends on.
struct DiscardQuery {
    uint32_t sf, m1;     // lone fives / lone ones that may be returned (0..2 each; ones only with smart_one)
    uint32_t so;         // lone ones of the roll whatever the strategy (the table key carries them and the smart_one flag apart)
    bool smart_one;
    uint32_t vmin, cmin; // thresholds in candidate units (0..7 / 0..5), see above
    uint32_t r15;        // min(raw score / 50, 15): the candidate with v == raw score / 50 would score 0 (score_lister :262)
    bool rb, fav;        // require_both, favor score
    bool eligible;       // smart_five, dice left over, something to return (:433)
};

__host__ __device__ inline DiscardQuery discard_query(const RawScore raw, int32_t n, int32_t turn_pre, const Strat &s) {
    DiscardQuery q;
    const uint32_t sf = (uint32_t)raw.sf, so = (uint32_t)raw.so;
    q.sf = sf;
    q.so = so;
    q.smart_one = s.has(SF_SMART_ONE);
    q.m1 = s.has(SF_SMART_ONE) ? so : 0u;
    q.eligible = s.has(SF_SMART_FIVE) & (raw.used != n) & ((sf | so) != 0u); // :433
    const bool cs = s.has(SF_CONSIDER_SCORE), cd = s.has(SF_CONSIDER_DICE);
    q.rb = s.has(SF_REQUIRE_BOTH);
    q.fav = s.has(SF_FAVOR_SCORE);
    const int32_t x = turn_pre + raw.score - s.score_thr;
    const uint32_t xq = ((uint32_t)(x < 1000 ? x : 1000) * 1311u) >> 16; // floor(x / 50) for 0 <= x <= 1000
    const uint32_t vmin = (x < 0) ? 0u : (xq + 1u < 7u ? xq + 1u : 7u);
    q.vmin = cs ? vmin : 0u;
    int32_t cm = s.dice_thr() - (n - raw.used) + 1;
    cm = cm < 0 ? 0 : (cm > 5 ? 5 : cm);
    q.cmin = cd ? (uint32_t)cm : 0u;
    const uint32_t r50 = ((uint32_t)raw.score * 1311u) >> 16; // raw.score / 50 (raw.score is a multiple of 50, < 2^16)
    q.r15 = r50 < 15u ? r50 : 15u;
    return q;
}

// SWAR search over the eight candidates; returns d5 | d1 << 2 of the best one, 0 if (0, 0) stays (eligibility aside).
// Relies on the ThresholdStrategy invariant require_both => consider_score && consider_dice (strategies.py:201-207,
// enforced by validate_strategies at the C-ABI): `rb` alone then selects the AND rule of _must_bank (:283-300).
__host__ __device__ inline uint32_t discard_choice(uint32_t sf, uint32_t m1, uint32_t vmin, uint32_t cmin, uint32_t r15, bool rb, bool fav) {
    // candidate layout for this strategy's preference
    const uint32_t D5 = fav ? CS_D5 : CD_D5, D1 = fav ? CS_D1 : CD_D1;
    const uint32_t V = D5 + 2u * D1, CNT = D5 + D1; // nibble-wise (max 6 / 4: no carries)
    const uint32_t ok5 = (sf * NIB_1 + (NIB_H - D5)) & NIB_H;  // d5 <= single fives (:326-329)
    const uint32_t ok1 = (m1 * NIB_1 + (NIB_H - D1)) & NIB_H;  // d1 <= single ones (smart_one only)
    const uint32_t mv = ((NIB_H + V) - vmin * NIB_1) & NIB_H;  // v >= vmin
    const uint32_t mc = ((NIB_H + CNT) - cmin * NIB_1) & NIB_H; // cnt >= cmin
    const uint32_t keep = rb ? (mv | mc) : (mv & mc);          // not must_bank
    const uint32_t xr = V ^ (r15 * NIB_1);
    const uint32_t nz = (xr | (xr << 1) | (xr << 2) | (xr << 3)) & NIB_H; // nibble != 0: candidate score != 0
    const uint32_t feas = ok5 & ok1 & keep & nz;
    const bool f0 = rb ? ((vmin == 0u) | (cmin == 0u)) : ((vmin == 0u) & (cmin == 0u)); // (0,0) not must_bank
    const bool any8 = feas != 0u;
    const uint32_t l4 = any8 ? (uint32_t)(ctz32(feas | 0u) - 3) : 0u;
    const bool take = any8 & !(fav & f0);
    return take ? (((D5 >> l4) & 3u) | (((D1 >> l4) & 3u) << 2)) : 0u;
}

// Discard table: the choice above for every (sf, so, vmin, cmin, r7, smart_one, rb, fav), 2^16 one-byte entries (64 KiB, built on
// the device once per context by running discard_choice on every key).  Key layout (round 5):
//   [1:0] min(lone fives, 2)  [3:2] min(lone ones, 2)  [6:4] vmin  [9:7] cmin  [12:10] r7 = min(raw score / 50, 7)
//   [13] smart_one  [14] require_both  [15] favor score                       (= the strategy's flag bits, SF_DISCARD_KEY_BITS)
// The lone ones count only under smart_one (m1 = smart_one ? so : 0): the TABLE applies that, so the roll's part of the key (bits 3:0
// and 12:10) depends on the roll alone and rides in the score table's entry (score_lut_entry32).  r7: a candidate scores 0 iff its
// discarded points v (<= 6) equal the raw score / 50, so every raw score of 350 and more behaves like 350 (the LDS image always had r7).
constexpr uint32_t DISCARD_LUT_KEYS = 1u << 16;
constexpr uint32_t DKEY_VMIN_SHIFT = 4, DKEY_CMIN_SHIFT = 7, DKEY_R7_SHIFT = 10;

__host__ __device__ inline uint32_t discard_key(const DiscardQuery &q) {
    const uint32_t r7 = q.r15 < 7u ? q.r15 : 7u, so = q.so;
    const bool smart_one = q.smart_one;
    return (q.sf < 2u ? q.sf : 2u) | ((so < 2u ? so : 2u) << 2) | (q.vmin << DKEY_VMIN_SHIFT) | (q.cmin << DKEY_CMIN_SHIFT) | (r7 << DKEY_R7_SHIFT) |
           (smart_one ? (uint32_t)SF_SMART_ONE : 0u) | (q.rb ? (uint32_t)SF_REQUIRE_BOTH : 0u) | (q.fav ? (uint32_t)SF_FAVOR_SCORE : 0u);
}

__host__ __device__ inline uint8_t discard_lut_entry(uint32_t key) {
    const uint32_t so = (key >> 2) & 3u, r7 = (key >> DKEY_R7_SHIFT) & 7u;
    return (uint8_t)discard_choice(key & 3u, (key & SF_SMART_ONE) ? so : 0u, (key >> DKEY_VMIN_SHIFT) & 7u, (key >> DKEY_CMIN_SHIFT) & 7u,
                                   r7 < 7u ? r7 : 15u, (key & SF_REQUIRE_BOTH) != 0u, (key & SF_FAVOR_SCORE) != 0u);
}

__host__ __device__ inline RollResult apply_discards(const RawScore raw, uint32_t choice) {
    const int32_t d5 = (int32_t)(choice & 3u), d1 = (int32_t)((choice >> 2) & 3u);
    RollResult out;
    out.d5 = d5; // == single_fives - best_sf (:467)
    out.d1 = d1;
    out.score = raw.score - 50 * d5 - 100 * d1; // apply_discards :575-578
    out.used = raw.used - d5 - d1;
    return out;
}

__host__ __device__ inline RollResult default_score_raw(const RawScore raw, int32_t n, int32_t turn_pre, const Strat &s) {
    const DiscardQuery q = discard_query(raw, n, turn_pre, s);
    return apply_discards(raw, q.eligible ? discard_choice(q.sf, q.m1, q.vmin, q.cmin, q.r15, q.rb, q.fav) : 0u);
}

// nibble-packed counts -> SWAR scorer (host checks, table construction)
__host__ __device__ inline RollResult default_score(uint32_t counts, int32_t n, int32_t turn_pre, const Strat &s) {
    return default_score_raw(score_counts(counts), n, turn_pre, s);
}

// 3-bit-packed counts -> score table (the kernels)
__device__ inline RollResult default_score_lut(const uint32_t *lut, const uint8_t *dlut, uint32_t key, int32_t n, int32_t turn_pre,
                                               const Strat &s) {
    const RawScore raw = raw_from_lut(lut[key] & 0xffffu);
    const DiscardQuery q = discard_query(raw, n, turn_pre, s);
    return apply_discards(raw, q.eligible ? (uint32_t)dlut[discard_key(q)] : 0u);
}

// ----------------------------------------------------------------------------------------
// The kernels' arithmetic in UNITS OF 50 POINTS.  Every Farkle score is a multiple of 50 (scoring_lookup.py:123-172: 50,
// 100, n * 100, 1000 ... 3000, 1500, 2500; discards take 50 / 100 away), so turn scores, banked scores and the score to
// beat are carried as score / 50: the table entry is used as it is (no x 50), the threshold terms need no division, and a
// seat's score fits 16 bits (the lean records of wide tables pack it beside the strategy index).  Thresholds and the
// target are arbitrary integers; they enter through exact integer bounds:
//     50 a <  thr   <=>  a <  ceil(thr / 50)         50 a >= target  <=>  a >= ceil(target / 50)
//     50 a >  x     <=>  a >  floor(x / 50)          (x = the initial score to beat, i.e. the target, engine.py:451)
// The functions above stay in points: they are the statement of the rule that the host-side exhaustive check
// (tests/native/device_header_host_check.hip) compares these against, and they build the two tables.
// ----------------------------------------------------------------------------------------
__host__ __device__ inline int32_t ceil_div50(int32_t a) { return a >= 0 ? (a + 49) / 50 : -((-a) / 50); }
__host__ __device__ inline int32_t floor_div50(int32_t a) { return a >= 0 ? a / 50 : -((-a + 49) / 50); }

struct Strat50 {
    int32_t thr50; // ceil(score_threshold / 50)
    uint32_t bits;
    int32_t dthr;  // dice threshold, sign-extended from bits[7:0] once
    __host__ __device__ Strat50(int32_t t, uint32_t b) : thr50(t), bits(b), dthr((int32_t)(int8_t)(b & 0xffu)) {}
    __host__ __device__ Strat50(int32_t t, uint32_t b, int32_t d) : thr50(t), bits(b), dthr(d) {}
    __host__ __device__ int32_t dice_thr() const { return dthr; }
    __host__ __device__ bool has(uint32_t f) const { return (bits & f) != 0u; }
};

__host__ __device__ inline Strat50 to_units50(const Strat &s) { return Strat50{ceil_div50(s.score_thr), s.bits}; }

struct Roll50 {
    int32_t score50, used, d5, d1;
};

// discard_query in units of 50: T = (turn score so far + raw score) / 50;  floor((50 T - thr) / 50) = T - ceil(thr / 50)
__host__ __device__ inline DiscardQuery discard_query50(uint32_t entry, int32_t n, int32_t pre50, const Strat50 &s) {
    DiscardQuery q;
    const uint32_t raw50 = entry & 63u, used = (entry >> 6) & 7u, sf = (entry >> 9) & 7u, so = (entry >> 12) & 7u;
    q.sf = sf;
    q.so = so;
    q.smart_one = s.has(SF_SMART_ONE);
    q.m1 = s.has(SF_SMART_ONE) ? so : 0u;
    q.eligible = s.has(SF_SMART_FIVE) & ((int32_t)used != n) & ((sf | so) != 0u); // :433
    const bool cs = s.has(SF_CONSIDER_SCORE), cd = s.has(SF_CONSIDER_DICE);
    q.rb = s.has(SF_REQUIRE_BOTH);
    q.fav = s.has(SF_FAVOR_SCORE);
    int32_t v = pre50 + (int32_t)raw50 - s.thr50 + 1;
    v = v < 0 ? 0 : (v > 7 ? 7 : v);
    q.vmin = cs ? (uint32_t)v : 0u;
    int32_t cm = s.dice_thr() - (n - (int32_t)used) + 1;
    cm = cm < 0 ? 0 : (cm > 5 ? 5 : cm);
    q.cmin = cd ? (uint32_t)cm : 0u;
    q.r15 = raw50 < 15u ? raw50 : 15u;
    return q;
}

__host__ __device__ inline Roll50 apply_discards50(uint32_t entry, uint32_t choice) {
    const int32_t d5 = (int32_t)(choice & 3u), d1 = (int32_t)((choice >> 2) & 3u);
    return Roll50{(int32_t)(entry & 63u) - d5 - 2 * d1, (int32_t)((entry >> 6) & 7u) - d5 - d1, d5, d1};
}

// 3-bit-packed counts -> score table -> discard table, units of 50: the readable form (every field decoded, the key assembled from the
// query) ...
__host__ __device__ inline Roll50 default_score_lut50_decoded(const uint32_t *lut, const uint8_t *dlut, uint32_t key, int32_t n, int32_t pre50,
                                                              const Strat50 &s) {
    const uint32_t e = lut[key] & 0xffffu;
    const DiscardQuery q = discard_query50(e, n, pre50, s);
    return apply_discards50(e, q.eligible ? (uint32_t)dlut[discard_key(q)] : 0u);
}

// ... and the game kernels' path (round 5): the same key from parts that are already in place.  The roll's share comes with the
// entry (high half), the strategy's share is three of its flag bits, the two thresholds are clamped sums shifted into their fields:
//     vmin = consider_score ? clamp(pre50 + raw50 - thr50 + 1, 0, 7) : 0        cmin = consider_dice ? clamp(dthr - (n - used) + 1, 0, 5) : 0
// (tests/native/device_header_host_check.hip runs it against the loop form of the rule on every multiset x flag set x threshold).
__host__ __device__ inline Roll50 default_score_lut50(const uint32_t *lut, const uint8_t *dlut, uint32_t key, int32_t n, int32_t pre50,
                                                      const Strat50 &s) {
    const uint32_t e = lut[key];
    const int32_t raw50 = (int32_t)(e & 63u), used = (int32_t)((e >> 6) & 7u);
    const bool eligible = s.has(SF_SMART_FIVE) & ((e & SE_SINGLES) != 0u) & (used != n); // scoring.py:433
    uint32_t choice = 0u;
    if (eligible) {
        int32_t v = pre50 + raw50 - s.thr50;             // vmin - 1 before clamping
        v = v < -1 ? -1 : (v > 6 ? 6 : v);
        int32_t c = s.dice_thr() - n + used;              // cmin - 1 before clamping
        c = c < -1 ? -1 : (c > 4 ? 4 : c);
        const uint32_t vpart = s.has(SF_CONSIDER_SCORE) ? (uint32_t)(v + 1) << DKEY_VMIN_SHIFT : 0u;
        const uint32_t cpart = s.has(SF_CONSIDER_DICE) ? (uint32_t)(c + 1) << DKEY_CMIN_SHIFT : 0u;
        choice = dlut[(e >> 16) | (s.bits & SF_DISCARD_KEY_BITS) | vpart | cpart];
    }
    const int32_t d5 = (int32_t)(choice & 3u), d1 = (int32_t)((choice >> 2) & 3u);
    return Roll50{raw50 - d5 - 2 * d1, used - d5 - d1, d5, d1};
}

// ----------------------------------------------------------------------------------------
// The two per-roll tables as an LDS IMAGE (LT instances of the game kernels; built once per context on the host from the
// functions above, copied in by every block at kernel start).  A gather from the 512 KiB / 64 KiB global tables costs the
// CU's texture addresser one cycle per lane (PMC, round 3: TA busy 49 % of the game kernel, two gathers per roll); the same
// look-ups from LDS cost three ds_read + ~8 VALU and no vector-memory instruction:
//   pair table  u32[512]: key half h (three 3-bit face counts) -> base(h) [15:0] | rank(h) << 16.  A roll's 18-bit key is two
//               halves (faces 1-3, faces 4-6); with rank() ordering the halves by dice count first, the roll's dense index is
//               base(low half) + rank(high half): 924 multisets of <= 6 dice
//   score table u16[924]: the entry of fk_device.h's score table for that multiset
//   discard table, 4 bits per entry: the choice of fk_device.h's discard_choice() for
//               index = ((((fav * 2 + rb) * 8 + min(r15, 7)) * 6 + cmin) * 8 + vmin) * 9 + m1 * 3 + sf   (13 824 entries)
// (round 5) score entries are 32 bits wide: the low half as in the global table's entry (bit 15 = the roll has lone 1s or 5s), the high
// half the roll's share of the DISCARD INDEX, already multiplied out:   [27:16] min(score / 50, 7) * 432 + min(lone fives, 2)
//                                                                        [31:28] 3 * min(lone ones, 2)   (counts under smart_one only)
// so that the index is that sum + (fav * 2 + rb) * 3456 + cmin * 72 + vmin * 9: two multiply-adds instead of the five of the nested form.
constexpr uint32_t LT_PAIR_OFF = 0, LT_SCORE_OFF = 2048, LT_SCORE_N = 924, LT_DISC_OFF = LT_SCORE_OFF + 4 * LT_SCORE_N /* 5744 */,
                   LT_DISC_N = 13824, LT_BYTES = LT_DISC_OFF + LT_DISC_N / 2 /* 12656 */;
static_assert(LT_DISC_OFF % 16 == 0 && LT_BYTES % 16 == 0, "table image is copied in 16-byte pieces");

__host__ __device__ inline uint32_t lt_score_entry32(uint32_t key) {
    const uint32_t e = score_lut_entry(key);
    if (e == 0u) return 0u;
    const uint32_t raw50 = e & 63u, sf = (e >> 9) & 7u, so = (e >> 12) & 7u;
    const uint32_t part = (raw50 < 7u ? raw50 : 7u) * 432u + (sf < 2u ? sf : 2u), so3 = 3u * (so < 2u ? so : 2u);
    return (e & 0x7fffu) | (((sf | so) != 0u) ? SE_SINGLES : 0u) | (part << 16) | (so3 << 28);
}

// score + discard choice of one roll from the LDS image (the kernels' path of default_score_lut50, same results)
__host__ __device__ inline Roll50 default_score_lds50(const uint8_t *img, uint32_t key, int32_t n, int32_t pre50, const Strat50 &s) {
    const uint32_t *pair = reinterpret_cast<const uint32_t *>(img + LT_PAIR_OFF);
    const uint32_t pa = pair[key & 511u], pb = pair[key >> 9];
    const uint32_t e = reinterpret_cast<const uint32_t *>(img + LT_SCORE_OFF)[(pa & 0xffffu) + (pb >> 16)];
    const int32_t raw50 = (int32_t)(e & 63u), used = (int32_t)((e >> 6) & 7u);
    const bool eligible = s.has(SF_SMART_FIVE) & ((e & SE_SINGLES) != 0u) & (used != n); // scoring.py:433
    uint32_t choice = 0u;
    if (eligible) {
        int32_t v = pre50 + raw50 - s.thr50; // vmin - 1 before clamping (see default_score_lut50)
        v = v < -1 ? -1 : (v > 6 ? 6 : v);
        int32_t c = s.dice_thr() - n + used;  // cmin - 1
        c = c < -1 ? -1 : (c > 4 ? 4 : c);
        const uint32_t vmin = s.has(SF_CONSIDER_SCORE) ? (uint32_t)(v + 1) : 0u, cmin = s.has(SF_CONSIDER_DICE) ? (uint32_t)(c + 1) : 0u;
        const uint32_t so3 = s.has(SF_SMART_ONE) ? (e >> 28) : 0u;
        const uint32_t strat = ((s.bits >> 14) & 3u) * 3456u; // (favor score * 2 + require_both) * 8 * 6 * 8 * 9: flags at bits 15, 14
        const uint32_t di = ((e >> 16) & 0xfffu) + so3 + strat + cmin * 72u + vmin * 9u;
        choice = ((uint32_t)(img + LT_DISC_OFF)[di >> 1] >> ((di & 1u) * 4u)) & 15u;
    }
    const int32_t d5 = (int32_t)(choice & 3u), d1 = (int32_t)((choice >> 2) & 3u);
    return Roll50{raw50 - d5 - 2 * d1, used - d5 - d1, d5, d1};
}
// The image itself (LT_BYTES, zero-filled by the caller): pair table, 32-bit score entries in dense multiset order, 4-bit discard choices.
inline void lt_build_image(uint8_t *img) {
    uint32_t *pair = reinterpret_cast<uint32_t *>(img + LT_PAIR_OFF);
    uint32_t *score = reinterpret_cast<uint32_t *>(img + LT_SCORE_OFF);
    auto hsum = [](uint32_t h) { return (h & 7u) + ((h >> 3) & 7u) + ((h >> 6) & 7u); };
    auto valid = [&](uint32_t h) { return (h & 7u) <= 6u && ((h >> 3) & 7u) <= 6u && ((h >> 6) & 7u) <= 6u && hsum(h) <= 6u; };
    uint32_t order[512], n_order = 0;
    for (uint32_t t = 0; t <= 6; ++t) // halves ordered by dice count first (stable in h)
        for (uint32_t h = 0; h < 512; ++h)
            if (valid(h) && hsum(h) == t) order[n_order++] = h;
    uint32_t rank[512] = {0}, upto[8] = {0};
    for (uint32_t i = 0; i < n_order; ++i) {
        rank[order[i]] = i;
        for (uint32_t t = hsum(order[i]); t <= 6; ++t) upto[t] += 1; // halves with at most t dice
    }
    uint32_t running = 0, base[512] = {0};
    for (uint32_t h = 0; h < 512; ++h)
        if (valid(h)) {
            base[h] = running;
            running += upto[6u - hsum(h)];
        }
    // running == 924: every multiset of at most six dice (the empty one included)
    for (uint32_t h = 0; h < 512; ++h) pair[h] = valid(h) ? (base[h] | (rank[h] << 16)) : 0u;
    for (uint32_t lo = 0; lo < 512; ++lo)
        for (uint32_t hi = 0; hi < 512; ++hi)
            if (valid(lo) && valid(hi) && hsum(lo) + hsum(hi) <= 6u) score[base[lo] + rank[hi]] = lt_score_entry32(lo | (hi << 9));
    uint8_t *disc = img + LT_DISC_OFF;
    for (uint32_t fav = 0; fav < 2; ++fav)
        for (uint32_t rb = 0; rb < 2; ++rb)
            for (uint32_t r7 = 0; r7 < 8; ++r7)
                for (uint32_t cmin = 0; cmin < 6; ++cmin)
                    for (uint32_t vmin = 0; vmin < 8; ++vmin)
                        for (uint32_t m1 = 0; m1 < 3; ++m1)
                            for (uint32_t sf = 0; sf < 3; ++sf) {
                                const uint32_t idx = (((((fav * 2u + rb) * 8u + r7) * 6u + cmin) * 8u + vmin) * 9u) + m1 * 3u + sf;
                                const uint32_t ch = discard_choice(sf, m1, vmin, cmin, r7, rb != 0u, fav != 0u) & 15u;
                                disc[idx >> 1] |= (uint8_t)(ch << ((idx & 1u) * 4u));
                            }
}

// Cold seat record of the hot / cold game kernel (fk_play_hc.h), three dwords.  The top bit of every counter field is a
// guard bit: a count that reaches it raises FK_ERR_COUNTER_OVERFLOW (the host replays the call on fk_play_kernel, whose
// fields are 16 bits wide) before the field can carry into its neighbour.
//   x = rolls [11:0] | farkles [20:12] | smart_five_uses [30:21] | has_scored [31]
//   y = n_smart_five_dice [10:0] | n_smart_one_dice [21:11] | smart_one_uses [31:22]
//   z = highest_turn / 50 [10:0] | banked total / 50 [22:11] | hot_dice [31:23]
// (highest turn and banked total are bounded by the turn-score guard and the launch plan: no guard bits.  The widths come from
// the reference's default grid: four never-banking strategies seated together play 200 rounds with up to 940 rolls, 200 farkles,
// 228 smart-five uses and 142 hot-dice turns of one seat.)
constexpr uint32_t HC_ROLLS_MASK = 0x7ffu, HC_FARKLE_SHIFT = 12, HC_FARKLE_MASK = 0xffu, HC_S5U_SHIFT = 21, HC_S1U_SHIFT = 22, HC_USES_MASK = 0x1ffu;
constexpr uint32_t HC_D1_SHIFT = 11, HC_DICE_MASK = 0x3ffu;
constexpr uint32_t HC_HI_MASK = 0x7ffu, HC_SCORE_SHIFT = 11, HC_SCORE_MASK = 0xfffu, HC_HOT_SHIFT = 23, HC_HOT_MASK = 0xffu, HC_HAS_SCORED = 1u << 31 /* in x */;
constexpr uint32_t HC_X_GUARD = (1u << 11) | (1u << 20) | (1u << 30), HC_Y_GUARD = (1u << 10) | (1u << 21) | (1u << 31), HC_Z_GUARD = 1u << 31;
constexpr int32_t HC_MAX_TARGET50 = 2700; // 2700 + 1310 < 4096

// should_continue in units of 50; stb50 = floor(score_to_beat / 50)
__host__ __device__ inline bool should_continue50(const Strat50 &s, int32_t turn50, int32_t dice_left, bool has_scored, bool final_round,
                                                  int32_t stb50, int32_t score50) {
    const bool above = score50 + turn50 > stb50;
    const bool stop = final_round & above & !s.has(SF_RUN_UP);
    const bool force = final_round & !above;
    const bool entry = !has_scored & (turn50 < 10); // 500 points (strategies.py:249)
    const bool cs = s.has(SF_CONSIDER_SCORE), cd = s.has(SF_CONSIDER_DICE);
    const bool want_s = cs & (turn50 < s.thr50);
    const bool want_d = cd & (dice_left > s.dice_thr());
    const bool thr = (cs & cd & !s.has(SF_REQUIRE_BOTH)) ? (want_s & want_d) : (want_s | want_d);
    return !stop & (force | entry | thr);
}

// FarklePlayer._should_continue (src/farkle/game/engine.py:156-205) with ThresholdStrategy.decide
// (src/farkle/simulation/strategies.py:212-275) and _decide_continue (:125-162) folded into boolean algebra:
//   stop  = final & running > to_beat & !run_up                      (engine.py:189)
//   force = final & running <= to_beat                               (engine.py:202, strategies.py:255)
//   entry = !has_scored & turn < 500                                 (strategies.py:249)
//   thr   = both considered ? (require_both ? want_s | want_d : want_s & want_d) : want_s | want_d   (:154-162)
//   continue = !stop & (force | entry | thr)
__host__ __device__ inline bool should_continue(const Strat &s, int32_t turn_score, int32_t dice_left, bool has_scored,
                                                bool final_round, int32_t score_to_beat, int32_t player_score) {
    const int32_t running_total = player_score + turn_score;
    const bool above = running_total > score_to_beat;
    const bool stop = final_round & above & !s.has(SF_RUN_UP);
    const bool force = final_round & !above;
    const bool entry = !has_scored & (turn_score < 500);
    const bool cs = s.has(SF_CONSIDER_SCORE), cd = s.has(SF_CONSIDER_DICE);
    const bool want_s = cs & (turn_score < s.score_thr);
    const bool want_d = cd & (dice_left > s.dice_thr());
    const bool thr = (cs & cd & !s.has(SF_REQUIRE_BOTH)) ? (want_s & want_d) : (want_s | want_d);
    return !stop & (force | entry | thr);
}

} // namespace fk
