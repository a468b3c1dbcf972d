// fk_kernels.h — the gfx950 kernels of the Farkle simulation engine (included by farkle_hip.hip, one translation unit).
//
//   fk_perm_kernel         one lane per shuffle: SeedSequence(ns=101) -> PCG64DXSM -> Fisher-Yates
//                          (Generator.permutation, run_tournament.py:312-318), arrays in LDS, written shuffle-minor.
//   fk_perm_draw_kernel +  large tables: the accepted draws of every shuffle at full occupancy, then the permutation WITHOUT
//   fk_perm_parallel_kernel  the serial swap chain (bucket sort + pointer jumping, one workgroup per shuffle), tiled into the
//   (+ fk_perm_block_kernel, blocked shuffle-minor layout; fk_perm_apply_kernel = the serial chains for tables beyond the
//    fk_perm_apply_kernel)  chain-free kernel's LDS reach.
//   fk_pool_kernel         SeedSequence pool after the words a shuffle / an H2H block shares (16 B per shuffle / block).
//   fk_class_count_kernel  sizes of the schedule classes (how patient the seats of a game are).
//   fk_seed_kernel         coordinate -> SeedSequence -> PCG64DXSM (state, increment) of every seat (random.py:80-188):
//                          one lane per game for the longest-first schedule, then one lane per (game, seat) pair so that the
//                          records of a wave — stored at the game's ticket — cover whole lines.
//   fk_play_kernel         persistent lanes, one lane = one game at a time, one roll per loop trip.  The seat records of the
//                          lane's game live in LDS (ten dwords per seat; loaded and stored inside every roll step).  State-
//                          store (GS) instances — on request, or when k records do not fit LDS — keep only the turn owner's
//                          record in LDS and the others in a per-game state store in HBM.  Finished lanes are handed new
//                          games in wave-level batches; per-strategy tallies privatised in LDS when they fit, otherwise one
//                          4-byte + one 32-byte result record per game.
//   fk_tally_reduce_kernel result records -> per-(batch, strategy) tallies, privatised in LDS slices (no HBM atomics
//                          from the game kernel).
//   fk_rows_kernel         state store + result records -> per-game rows (4 + 28k bytes), a streaming post-pass.
//   fk_seat_stats_kernel   state store + result records -> integer sufficient statistics of ALL seats per strategy.
//   fk_block_map_kernel,   batched H2H: game -> block, and result records of many blocks -> per-block completed / safety /
//   fk_h2h_reduce_kernel   wins.
//   fk_finalize_tally, fk_score_lut_kernel, fk_discard_lut_kernel, fk_dbg_* probes.
//
// The per-roll arithmetic (SeedSequence, PCG64DXSM, dice, scoring, discards, decisions) lives in fk_device.h.
#pragma once

#include "../../include/farkle_hip.h"
#include "fk_device.h"

#include <hip/hip_runtime.h>

using namespace fk;

namespace {

constexpr uint32_t HC_AFTER_6_WORDS = ss_hc(24);  // 4 + 12 (all pairs) + 2*4 hashmix calls
constexpr uint32_t HC_AFTER_12_WORDS = ss_hc(48); // + 6*4: shuffle, pair, order absorbed
constexpr uint32_t HC_AFTER_14_WORDS = ss_hc(56); // + 2*4: game_index absorbed

enum : uint32_t { MODE_PERM = 0, MODE_LIST = 1, MODE_BLOCKS = 2 };

// LDS seat-record fields (dwords), record layout lds[(seat * BLOCK + tid) * NFIELDS + field]
enum : uint32_t {
    F_LO0 = 0, F_LO1, F_HI0, F_HI1, F_INC_LO0, F_INC_LO1, F_INC_HI0, F_INC_HI1,
    F_BUF, F_SCORE, F_CA, F_CB, F_CC, F_CD, F_CE, F_SPX, F_SPY, NF
};
// packed u16 counter pairs, in the state store (R_*) and in full LDS records (scores are carried in units of 50 points)
//   cA = rolls | farkles << 16        cB = highest_turn / 50 | n_turns << 16 (the n_turns half: state store only)
//   cC = sf_uses | sf_dice << 16      cD = so_uses | so_dice << 16
//   cE = hot_dice | flags << 16       flags: bit0 has_scored, bit1 has_buf
constexpr uint32_t CE_HAS_SCORED = 1u << 16, CE_HAS_BUF = 1u << 17;
constexpr uint32_t CE_IDX_SHIFT = 18; // LEAN records in LDS: strategy index in cE[31:18] (S <= 16384)
// LEAN LDS record, ten dwords: LO0 LO1 | HI0 HI1 | BUF cA | cB cC | cD cE with
//   cB = highest_turn / 50 | hot_dice << 16        cE = score / 50 | has_scored << 16 | has_buf << 17 | strategy << 18
// (n_turns is a function of rounds, trigger seat and final round: restored when the game ends).  The banked total must fit
// 16 bits in units of 50: it stays below target / 50 + one turn (<= 1310), so LEAN needs target / 50 <= LEAN_MAX_TARGET50.
constexpr uint32_t LEAN_DW = 10;
constexpr int32_t LEAN_MAX_TARGET50 = 64000;

// State record of one seat in HBM (the seed kernel writes it, GS game kernels keep it current, the post-passes read the
// final one): the eleven dwords a turn mutates with score, n_turns and hot dice spelled out, + the seat's strategy index.
// 48 bytes = three 16-byte accesses; index = slot * k + seat (slot = ticket position of the game's schedule).
// R_SCORE and the highest_turn half of R_CB are in units of 50 points (fk_device.h); the post-passes multiply them out.
enum : uint32_t { R_LO0 = 0, R_LO1, R_HI0, R_HI1, R_BUF, R_SCORE, R_CA, R_CB, R_CC, R_CD, R_CE, R_IDX, STATE_DW = 12 };

// Result record of one finished game (index = game id), written by the game kernel when the tallies are not privatised
// in LDS or when rows / all-seat statistics are wanted:
//   d0 = winner's strategy index [23:0] | winner seat [30:24] | safety-limit flag [31]
//   d1 = winning score   d2 = n_rounds | farkles << 16   d3 = rolls | highest_turn << 16
//   d4 = sf_uses | sf_dice << 16   d5 = so_uses | so_dice << 16   d6 = hot_dice   d7 = 0      (winner's counters)
// d0 is also stored in a dense array of its own (rec0[id]): the reducing post-passes filter on it at 4 bytes per game and
// touch the 32-byte record of the games they keep only; H2H block launches store rec0 alone.
constexpr uint32_t REC_DW = 8, REC_SAFETY = 0x80000000u;

constexpr uint32_t LT_COLS = 24; // LDS tally columns: wins, completed, safety, 10 sums, 10 square sums, pad
constexpr uint32_t TICKET_CHUNK = 64;

struct DevOverride {
    uint32_t game; // chunk-local game id (the list is sorted by it)
    uint32_t max_rounds;
};

// One H2H block of a batched launch (pass): games [start, next block's start) are its attempts attempt0, attempt0 + 1, ...;
// `row` = the block's index in the CALL's block list: its seats are rows 2 row, 2 row + 1 of the strategy table, and its
// counts go to out[row] of the reduce pass — the table and the counts are per call, so that the passes of a call share them
// (and the next pass can be prepared while this one plays).
struct DevBlock {
    uint64_t pair, attempt0;
    uint32_t order, start, row, pad;
};

struct SeedArgs {
    SeedPool prefix;         // pool after entropy words 0..5 (version, namespace, root, k)
    const fk_coord *coords;  // LIST mode: explicit coordinates (full SeedSequence per game)
    uint64_t shuffle0, pair, order, game0;
    uint32_t gps;            // games per shuffle (affine id -> (shuffle, game)); 0 = no split
    uint32_t k;
    uint32_t n_games;
    uint32_t *state;         // [n_games][k][state_dw]: state_dw = 4 (PCG state {lo, hi} only, read once per game) or
                             // STATE_DW (the full initial state record, strategy index included)
    uint32_t state_dw;
    uint16_t *seat_idx;      // nullable, [n_games][k]: strategy index of every seat at the game's slot (tournament launches with
                             // state_dw == 4): the game kernel's hand-over reads it instead of redoing three integer
                             // divisions and a permutation gather per seat
    uint4 *inc;              // [n_games][k] PCG increment {lo, hi}: its own plane, re-read at every turn start by
                             // lean-record kernels (a compact plane keeps the increments of all resident games in L2)
    const int32_t *seat_strategy; // LIST mode, state_dw == STATE_DW: [n_games][k] strategy indices
    // longest-first scheduling (tournament mode): games whose seats ALL never bank run to the round
    // limit (~13x the mean length); they are dealt first so that they do not form the tail of a wave.
    const uint16_t *perm_T;  // nullable; blocked layout, see perm_at()
    uint32_t perm_slots, S;
    const uint8_t *patience;   // per strategy: 3 = never banks voluntarily ... 0 = banks readily (scheduling only)
    uint32_t n_sh;
    uint32_t *sched;         // [n_games] ticket -> game id, in dealing order (see the kernel)
    const uint32_t *class_ctr; // [SCHED_CLASSES] class sizes (fk_class_count_kernel)
    uint32_t *sched_ctr;     // [SCHED_CLASSES] per-class cursors
    // batched H2H blocks (MODE_BLOCKS): game -> (block, attempt)
    const DevBlock *blocks;
    uint32_t n_blocks;
    const uint32_t *game_block; // [n_games] pass-local block index of every game (fk_block_map_kernel)
    const uint32_t *game_row;   // [n_games] the block's row in the call's block list (strategy rows 2 row, 2 row + 1)
    // SeedSequence pool after entropy words 0..11 (…, shuffle, pair, order), shared by every game of a shuffle (tournament) or
    // of a block (batched H2H): [n_sh] / [n_blocks], written by fk_pool_kernel
    const uint4 *pools;
};

struct PlayArgs {
    const uint2 *strat;          // [S] packed strategies
    const uint32_t *score_lut;   // [SCORE_LUT_KEYS] score table, 32-bit entries (fk_device.h: score_lut_entry32)
    const uint8_t *discard_lut;  // [DISCARD_LUT_KEYS] discard table (fk_device.h)
    const uint16_t *perm_T;      // blocked permutations (MODE_PERM), see perm_at()
    uint32_t perm_slots;
    const int32_t *seat_strategy; // [n_games][k] (MODE_LIST)
    const uint32_t *game_block;  // [n_games] (MODE_BLOCKS): SeedArgs::game_row — strategy index of seat s = 2 * row + s
    uint32_t *state;             // seat state records, see SeedArgs (GS instances update them in place)
    uint32_t state_dw;
    const uint4 *inc;
    const uint16_t *seat_idx;    // nullable, see SeedArgs
    const uint32_t *sched;       // nullable: ticket -> game id (longest-first schedule; state records are stored by ticket)
    unsigned long long *tally;   // [S][26] (LDS-tally launches only: one batch)
    uint32_t *rec0;              // nullable: [n_games] d0 of the result records
    uint32_t *recs;              // nullable: [n_games][REC_DW] result records (tournament / list launches)
    uint32_t gs_out;             // LDS-record instances: flush every seat's final record to `state` at game end
    uint32_t *ticket;
    int32_t *err;                // [0] code, [1] game id
    const DevOverride *ov;
    uint32_t n_ov;
    uint32_t mode;
    uint32_t n_games, gps, n_sh, k, S;
    int32_t target50;            // ceil(target_score / 50): a banked total of s / 50 reaches the target iff s / 50 >= target50
    int32_t beat50;              // floor(target_score / 50): the initial score to beat (engine.py:451) in the same units
    uint32_t max_rounds;
    uint32_t batch_threshold;
    uint32_t use_lds_tally;
    uint32_t uflags;             // the flag bits (8..15) every strategy of the table shares, see MIXED below
    uint4 *cold;                 // fk_play_hc_kernel: [resident lanes][k] cold seat records (fk_play_hc.h)
    const uint8_t *lds_tables;   // fk_play_hc_kernel, LT instances: the LDS image of the score / discard tables (LT_BYTES, fk_device.h)
    unsigned long long *clk;     // nullable (option "clock_stamps"): [grid][4] = s_memtime, s_memrealtime at the block's first and last
                                 // instruction (zeroed before the launch) — the shader clock the launch really ran at (MI355X_MICROARCH.md, DVFS give-back item 6)
};

// One pair of stamps per block and end of the kernel: shader-clock ticks and the constant 100 MHz reference counter.  The first
// is the block's first wave's, the last is the LAST wave's to leave (the waves of a persistent block drain at different times).
__device__ inline void clock_stamp(unsigned long long *clk, uint32_t which) {
    if (!clk) return;
    unsigned long long *slot = clk + (size_t)blockIdx.x * 4u + 2u * which;
    if (which == 0u) {
        if (threadIdx.x == 0) {
            slot[0] = __builtin_amdgcn_s_memtime();
            slot[1] = __builtin_amdgcn_s_memrealtime();
        }
    } else if ((threadIdx.x & 63u) == 0u) {
        atomicMax(slot, (unsigned long long)__builtin_amdgcn_s_memtime());
        atomicMax(slot + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ inline uint32_t mbcnt(uint64_t mask) { // lanes of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ inline Strat unpack_strat(uint2 v) { return Strat{(int32_t)v.x, v.y}; }

// Permutations are stored blocked: [n_sh / slots][S][slots] (u16), `slots` = shuffles one fk_perm_kernel block holds
// in LDS.  A block writes one contiguous region; entry e of consecutive shuffles is contiguous inside a block.
__device__ inline uint32_t perm_at(const uint16_t *perm, uint32_t S, uint32_t slots, uint32_t sh, uint32_t e) {
    const uint32_t b = sh / slots, l = sh - b * slots;
    return perm[((size_t)b * S + e) * slots + l];
}

// ---------------------------------------------------------------------------------------
// Each lane shuffles its own u16[S] array held in LDS (lane-private, contiguous), then the block writes the
// arrays out shuffle-minor so that the writes — and every later read of entry i across shuffles — are coalesced.
// Fisher-Yates is a dependent chain of S swaps per shuffle; in LDS a step costs two ds_read + two ds_write instead
// of four scattered HBM/L2 transactions.  LDS capacity fixes the shuffles per CU (`slots` = min(512, 160 KiB / 2S):
// 512 at S <= 160, 15 at the 5 160-strategy grid); because each chain is latency-bound the slots are spread over
// the block's 8 waves (2 per SIMD) rather than packed into one.
constexpr int PERM_BLOCK = 512, PERM_WAVES = PERM_BLOCK / 64;

__global__ __launch_bounds__(PERM_BLOCK) void fk_perm_kernel(SeedPool prefix, uint64_t shuffle0, uint32_t n_sh, uint32_t S,
                                                             uint32_t slots, uint16_t *perm_T) {
    extern __shared__ uint16_t perm_lds[];
    const uint32_t per_wave = (slots + PERM_WAVES - 1u) / PERM_WAVES;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = wave * per_wave + lane;
    const uint32_t sh = blockIdx.x * slots + slot;
    const bool valid = lane < per_wave && slot < slots && sh < n_sh;
    uint16_t *a = perm_lds + (size_t)(valid ? slot : 0u) * S;
    if (valid) {
        for (uint32_t e = 0; e < S; ++e) a[e] = (uint16_t)e;
    }
    // Fisher-Yates: for i = S-1 .. 1: j = random_interval(i) (masked rejection on the buffered 32-bit stream);
    // swap(a[i], a[j]).  Every lane consumes exactly one 32-bit word per trip and only advances its own `i` when
    // the word is accepted, so a lane never waits for another lane's rejections and the low/high half-word phase
    // is uniform across the wave: one PCG64DXSM output per two trips, no divergence.
    Rng r{};
    if (valid) {
        SeedPool p = prefix;
        p.hc = HC_AFTER_6_WORDS;
        ss_absorb64(p, shuffle0 + sh); // shuffle_index
#pragma unroll
        for (int w = 0; w < 5; ++w) ss_absorb64(p, 0); // pair_id, order, game_index, seat_index, replicate_index
        uint32_t g8[8];
        ss_generate<8>(p, g8);
        pcg_seed(r, g8);
    }
    uint32_t i = valid ? S - 1u : 0u;
    auto consume = [&](uint32_t w) {
        if (i >= 1u) {
            const uint32_t j = w & (0xffffffffu >> __clz((int)i));
            if (j <= i) {
                const uint16_t ai = a[i], aj = a[j];
                a[i] = aj;
                a[j] = ai;
                i -= 1u;
            }
        }
    };
    while (__ballot(i >= 1u)) {
        const uint64_t o = pcg_next64(r);
        consume((uint32_t)o);         // low half first ...
        consume((uint32_t)(o >> 32)); // ... then the buffered high half
    }
    __syncthreads();
    // blocked store [block][e][slot]: one contiguous, fully coalesced region per block
    const uint32_t first = blockIdx.x * slots;
    const uint32_t count = min(slots, n_sh > first ? n_sh - first : 0u);
    uint16_t *out = perm_T + (size_t)blockIdx.x * S * slots;
    for (uint32_t idx = threadIdx.x; idx < S * slots; idx += PERM_BLOCK) {
        const uint32_t e = idx / slots, l = idx - e * slots;
        out[idx] = l < count ? perm_lds[(size_t)l * S + e] : (uint16_t)0;
    }
}

// ---------------------------------------------------------------------------------------
// Fisher-Yates in two kernels (large tables).  The swap chain of one shuffle is serial and its arrays fill LDS (15
// chains per CU at S = 5 160), so every instruction on the chain is paid at the latency of a nearly empty CU.  The
// draws do not depend on the array: fk_perm_draw_kernel produces the accepted j of every step — SeedSequence,
// PCG64DXSM and the masked rejection, the expensive part — with one lane per shuffle at full occupancy, packed eight
// 16-bit draws per 16-byte store, group-major ([group][shuffle]) so that lanes in step write neighbouring words;
// fk_perm_apply_kernel then runs the bare chain (two LDS reads, two LDS writes per step) over the LDS arrays.
constexpr int DRAW_BLOCK = 256;

__global__ __launch_bounds__(DRAW_BLOCK) void fk_perm_draw_kernel(SeedPool prefix, uint64_t shuffle0, uint32_t n_sh, uint32_t S,
                                                                 uint32_t g_stride, uint32_t sh_stride, uint4 *draws) {
    // group g of shuffle sh sits at draws[g * g_stride + sh * sh_stride]: group-major (n_sh_pad, 1) for the serial swap
    // chains, one row per shuffle (1, row length) for fk_perm_parallel_kernel
    const uint32_t sh = blockIdx.x * DRAW_BLOCK + threadIdx.x;
    const bool valid = sh < n_sh;
    Rng r{};
    if (valid) {
        SeedPool p = prefix;
        p.hc = HC_AFTER_6_WORDS;
        ss_absorb64(p, shuffle0 + sh); // shuffle_index
#pragma unroll
        for (int w = 0; w < 5; ++w) ss_absorb64(p, 0); // pair_id, order, game_index, seat_index, replicate_index
        uint32_t g8[8];
        ss_generate<8>(p, g8);
        pcg_seed(r, g8);
    }
    uint32_t i = valid ? S - 1u : 0u, n = 0;
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0; // the eight most recent draws, oldest in the low half of q0
    auto consume = [&](uint32_t w) {
        if (i >= 1u) {
            const uint32_t j = w & (0xffffffffu >> __clz((int)i));
            if (j <= i) {
                q0 = (q0 >> 16) | (q1 << 16);
                q1 = (q1 >> 16) | (q2 << 16);
                q2 = (q2 >> 16) | (q3 << 16);
                q3 = (q3 >> 16) | (j << 16);
                i -= 1u;
                n += 1u;
                if ((n & 7u) == 0u) draws[(size_t)((n >> 3) - 1u) * g_stride + (size_t)sh * sh_stride] = make_uint4(q0, q1, q2, q3);
            }
        }
    };
    while (__ballot(i >= 1u)) {
        const uint64_t o = pcg_next64(r);
        consume((uint32_t)o);         // low half first ...
        consume((uint32_t)(o >> 32)); // ... then the buffered high half
    }
    if (valid && (n & 7u)) draws[(size_t)(n >> 3) * g_stride + (size_t)sh * sh_stride] = make_uint4(q0, q1, q2, q3); // last group: draws in the TOP halves
}

__global__ __launch_bounds__(PERM_BLOCK) void fk_perm_apply_kernel(const uint4 *draws, uint32_t n_sh_pad, uint32_t n_sh, uint32_t S,
                                                                  uint32_t slots, uint16_t *perm_T) {
    extern __shared__ uint16_t perm_lds[];
    const uint32_t per_wave = (slots + PERM_WAVES - 1u) / PERM_WAVES;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = wave * per_wave + lane;
    const uint32_t sh = blockIdx.x * slots + slot;
    const bool valid = lane < per_wave && slot < slots && sh < n_sh;
    uint16_t *a = perm_lds + (size_t)(valid ? slot : 0u) * S;
    if (valid) {
        for (uint32_t e = 0; e < S; ++e) a[e] = (uint16_t)e;
        const uint32_t steps = S - 1u, groups = (steps + 7u) >> 3, tail = steps & 7u;
        uint32_t i = S - 1u;
        // The draws come from HBM/L2 (a couple of thousand cycles away); a group of eight steps takes about a thousand:
        // keep DEPTH groups in flight ahead of the chain.
        constexpr uint32_t DEPTH = 4;
        uint4 ring[DEPTH];
#pragma unroll
        for (uint32_t d = 0; d < DEPTH; ++d) ring[d] = (d < groups) ? draws[(size_t)d * n_sh_pad + sh] : make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t g0 = 0; g0 < groups; g0 += DEPTH) {
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; ++d) {
                const uint32_t g = g0 + d;
                if (g < groups) {
                    const uint4 q = ring[d];
                    ring[d] = (g + DEPTH < groups) ? draws[(size_t)(g + DEPTH) * n_sh_pad + sh] : make_uint4(0u, 0u, 0u, 0u);
                    uint32_t w[4] = {q.x, q.y, q.z, q.w};
                    uint32_t cnt = 8u;
                    if (g + 1u == groups && tail) { // the last group's draws sit in the top `tail` halves: bring them down
                        cnt = tail;
                        const uint32_t drop = 8u - tail; // halves to shift out
                        for (uint32_t x = 0; x < drop; ++x) {
                            w[0] = (w[0] >> 16) | (w[1] << 16);
                            w[1] = (w[1] >> 16) | (w[2] << 16);
                            w[2] = (w[2] >> 16) | (w[3] << 16);
                            w[3] = w[3] >> 16;
                        }
                    }
#pragma unroll
                    for (uint32_t t = 0; t < 8u; ++t) {
                        if (t < cnt) {
                            const uint32_t j = (w[t >> 1] >> (16u * (t & 1u))) & 0xffffu;
                            const uint16_t ai = a[i], aj = a[j];
                            a[i] = aj;
                            a[j] = ai;
                            i -= 1u;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // blocked store [block][e][slot]: one contiguous, fully coalesced region per block
    const uint32_t first = blockIdx.x * slots;
    const uint32_t count = min(slots, n_sh > first ? n_sh - first : 0u);
    uint16_t *out = perm_T + (size_t)blockIdx.x * S * slots;
    for (uint32_t idx = threadIdx.x; idx < S * slots; idx += PERM_BLOCK) {
        const uint32_t e = idx / slots, l = idx - e * slots;
        out[idx] = l < count ? perm_lds[(size_t)l * S + e] : (uint16_t)0;
    }
}

// ---------------------------------------------------------------------------------------
// Fisher-Yates WITHOUT the serial chain: one workgroup per shuffle, everything in LDS.
// numpy runs  for i = S-1 .. 1: swap(a[i], a[j_i])  on a = arange(S).  Position i is final after step i, and
//     final[i] = what position j_i held just before step i.
// A position p <= i changes before step i only as the TARGET of an earlier step i' > i with j_i' = p, which leaves
// there what position i' held just before step i'.  With f(i) = content of position i just before step i:
//     f(i)     = f(n(i)) if n(i) exists, else i,          n(i) = min{ i' > i : j_i' = i }     (a pointer chain, ascending)
//     final[i] = f(u(i)) if u(i) exists, else j_i,        u(i) = min{ i' > i : j_i' = j_i }
//     final[0] = f(0).
// n and u are "next step with the same target": the steps are bucketed by target with a counting sort (LDS atomics + one
// scan), each (tiny) bucket is sorted, and the chains are resolved by pointer jumping — O(S log S) fully parallel work
// instead of S dependent LDS round trips per shuffle.  14 bytes of LDS per strategy: two shuffles per CU at S = 5 160.
constexpr int PP_BLOCK = 1024;
constexpr uint32_t PP_NONE = 0xffffu;

__global__ __launch_bounds__(PP_BLOCK) void fk_perm_parallel_kernel(uint4 *draws_rows, uint32_t row_u4, uint32_t n_sh, uint32_t S) {
    extern __shared__ uint32_t pp_lds[];
    __shared__ uint32_t wave_sum[PP_BLOCK / 64];
    __shared__ uint32_t changed;
    uint32_t *cnt = pp_lds;                                              // [S] bucket sizes, then fill cursors
    uint16_t *J = reinterpret_cast<uint16_t *>(cnt + S);                 // [S] draw of step i (J[0] unused)
    uint16_t *order = J + S, *U = order + S, *F = U + S, *start = F + S; // [S] each
    const uint32_t sh = blockIdx.x, tid = threadIdx.x;
    if (sh >= n_sh) return;
    uint4 *row = draws_rows + (size_t)sh * row_u4;
    const uint32_t steps = S - 1u, groups = (steps + 7u) >> 3, tail = steps & 7u;
    for (uint32_t i = tid; i < S; i += PP_BLOCK) {
        cnt[i] = 0u;
        U[i] = (uint16_t)PP_NONE;
    }
    if (tid == 0) J[0] = 0;
    for (uint32_t g = tid; g < groups; g += PP_BLOCK) { // draw number n = 8g + e is the draw of step i = S - 1 - n
        const uint4 q = row[g];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        const bool last = (g + 1u == groups) && tail;
        const uint32_t n_in = last ? tail : 8u, shift = last ? 8u - tail : 0u; // the last group's draws sit in the TOP halves
        for (uint32_t e = 0; e < n_in; ++e) {
            const uint32_t h = e + shift;
            J[S - 1u - (8u * g + e)] = (uint16_t)((w[h >> 1] >> (16u * (h & 1u))) & 0xffffu);
        }
    }
    __syncthreads();
    for (uint32_t i = 1u + tid; i < S; i += PP_BLOCK) atomicAdd(&cnt[J[i]], 1u);
    __syncthreads();
    // exclusive scan of cnt: a contiguous chunk per thread, wave scan of the chunk sums, block scan of the wave sums
    const uint32_t chunk = (S + PP_BLOCK - 1u) / PP_BLOCK, c0 = min(tid * chunk, S), c1 = min(c0 + chunk, S);
    uint32_t mine = 0;
    for (uint32_t i = c0; i < c1; ++i) mine += cnt[i];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)(tid & 63u) >= d) incl += up;
    }
    if ((tid & 63u) == 63u) wave_sum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wave_sum[w];
    for (uint32_t i = c0; i < c1; ++i) {
        const uint32_t c = cnt[i];
        cnt[i] = base;
        start[i] = (uint16_t)base;
        base += c;
    }
    __syncthreads();
    for (uint32_t i = 1u + tid; i < S; i += PP_BLOCK) order[atomicAdd(&cnt[J[i]], 1u)] = (uint16_t)i;
    __syncthreads();
    for (uint32_t p = tid; p < S; p += PP_BLOCK) { // bucket p = the steps that target position p, ascending
        const uint32_t lo = start[p], hi = cnt[p];
        for (uint32_t a = lo + 1u; a < hi; ++a) { // insertion sort (buckets hold one or two steps on average)
            const uint16_t v = order[a];
            uint32_t b = a;
            while (b > lo && order[b - 1u] > v) {
                order[b] = order[b - 1u];
                --b;
            }
            order[b] = v;
        }
        uint32_t nxt = PP_NONE;
        for (uint32_t a = lo; a < hi; ++a) {
            const uint32_t i = order[a];
            if (a + 1u < hi) U[i] = order[a + 1u];
            if (nxt == PP_NONE && i > p) nxt = i; // n(p): the first step of the bucket beyond p itself (a self-swap is step p)
        }
        F[p] = (uint16_t)(nxt == PP_NONE ? p : nxt);
    }
    __syncthreads();
    for (int round = 0; round < 17; ++round) { // pointer jumping: chains ascend, terminals are fixed points
        if (tid == 0) changed = 0u;
        __syncthreads();
        uint32_t any = 0;
        for (uint32_t i = tid; i < S; i += PP_BLOCK) {
            const uint32_t f = F[i], ff = F[f];
            if (ff != f) {
                F[i] = (uint16_t)ff;
                any = 1u;
            }
        }
        if (any) changed = 1u;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    // the finished permutation goes back over the shuffle's own draws row (read completely in the first phase)
    uint16_t *out = reinterpret_cast<uint16_t *>(row);
    for (uint32_t i = tid; i < S; i += PP_BLOCK) {
        const uint32_t u = U[i];
        out[i] = (i == 0u) ? F[0] : (u != PP_NONE ? F[u] : J[i]);
    }
}

// rows [n_sh][row_u16] (one permutation per row) -> the blocked layout [n_sh / slots][S][slots] the game kernels read;
// a workgroup moves 256 entries of up to 64 shuffles of one block through an LDS tile (coalesced on both sides)
constexpr uint32_t PB_ROWS = 64;

__global__ __launch_bounds__(256) void fk_perm_block_kernel(const uint16_t *rows, uint32_t row_u16, uint32_t n_sh, uint32_t S, uint32_t slots,
                                                           uint16_t *perm_T) {
    __shared__ uint16_t tile[PB_ROWS * 256];
    const uint32_t b = blockIdx.y, e0 = blockIdx.x * 256u, ne = min(256u, S - e0);
    uint16_t *out = perm_T + ((size_t)b * S + e0) * slots;
    for (uint32_t l0 = 0; l0 < slots; l0 += PB_ROWS) {
        const uint32_t lc = min(PB_ROWS, slots - l0);
        for (uint32_t l = 0; l < lc; ++l) {
            const uint32_t sh = b * slots + l0 + l;
            if (threadIdx.x < ne) tile[l * 256u + threadIdx.x] = sh < n_sh ? rows[(size_t)sh * row_u16 + e0 + threadIdx.x] : (uint16_t)0;
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < ne * lc; j += 256u) {
            const uint32_t e = j / lc, l = j - e * lc;
            out[(size_t)e * slots + l0 + l] = tile[l * 256u + e];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
constexpr int SEED_BLOCK = 1024;

// Longest-first schedule classes (scheduling only: results do not depend on the order games are dealt in).
// A strategy's patience is 3 if it never banks voluntarily (its games against other such seats run to the round
// limit, ~13x the mean length), 2 / 1 if it rolls on until one / two dice are left unless BOTH of its conditions say
// bank (long turns, many farkles: 1.6x / 1.2x the mean game length on the reference grid), else 0.  A game's class is 0
// when every seat has patience 3, otherwise 15 - min(sum of patience, 14): class 0 is dealt first, class 15 last, so the
// launch drains on the games that are shortest in expectation.
constexpr uint32_t SCHED_CLASSES = 16;

__device__ inline uint32_t class_of(uint32_t sum, uint32_t n_never, uint32_t k) {
    return n_never == k ? 0u : (SCHED_CLASSES - 1u) - min(sum, SCHED_CLASSES - 2u);
}

__device__ inline uint32_t schedule_class(const uint16_t *perm_T, uint32_t perm_slots, uint32_t S, uint32_t k,
                                          const uint8_t *patience, uint32_t sh_local, uint32_t g_local) {
    uint32_t sum = 0, n_never = 0;
    for (uint32_t s = 0; s < k; ++s) {
        const uint32_t p = patience[perm_at(perm_T, S, perm_slots, sh_local, g_local * k + s)];
        sum += p;
        n_never += (p == 3u) ? 1u : 0u;
    }
    return class_of(sum, n_never, k);
}

// batched H2H: the two seats of block b are table rows 2b, 2b + 1
__device__ inline uint32_t schedule_class_block(const uint8_t *patience, uint32_t blk) {
    const uint32_t p0 = patience[2u * blk], p1 = patience[2u * blk + 1u];
    return class_of(p0 + p1, (p0 == 3u ? 1u : 0u) + (p1 == 3u ? 1u : 0u), 2u);
}

// batched H2H: game -> block (the block with the largest start <= game), one lane per game
__global__ void fk_block_map_kernel(const DevBlock *blocks, uint32_t n_blocks, uint32_t n_games, uint32_t *game_block, uint32_t *game_row) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_games) return;
    uint32_t lo = 0, hi = n_blocks;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blocks[mid].start <= t) lo = mid;
        else hi = mid;
    }
    game_block[t] = lo;
    game_row[t] = blocks[lo].row;
}

// Entropy words 6..11 (shuffle_index, pair_id, order: random.py:106-111) are the same for every game of a shuffle (tournament)
// or of a block (batched H2H): absorbed once here instead of once per game (24 of the 32 + 16 k pool updates a game costs).
__global__ void fk_pool_kernel(SeedPool prefix, uint64_t shuffle0, uint64_t pair, uint64_t order, const DevBlock *blocks, uint32_t n,
                               uint4 *pools) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SeedPool gp = prefix;
    gp.hc = HC_AFTER_6_WORDS;
    if (blocks) {
        ss_absorb64(gp, 0); // shuffle_index
        ss_absorb64(gp, blocks[i].pair);
        ss_absorb64(gp, blocks[i].order);
    } else {
        ss_absorb64(gp, shuffle0 + i);
        ss_absorb64(gp, pair);
        ss_absorb64(gp, order);
    }
    pools[i] = make_uint4(gp.p[0], gp.p[1], gp.p[2], gp.p[3]);
}

// An index word pair whose high half is zero (every index below 2^32): the hash of the zero word is a compile-time constant.
__device__ inline void ss_absorb_index(SeedPool &s, uint64_t v) {
    ss_absorb(s, (uint32_t)v);
    if ((uint32_t)(v >> 32) == 0u) ss_absorb(s, 0u);
    else ss_absorb(s, (uint32_t)(v >> 32));
}

// Class sizes: one lane per game in the seed kernel's walk order (coalesced permutation reads), grid-stride so that few
// blocks add to the same global words at the end.
__global__ __launch_bounds__(SEED_BLOCK) void fk_class_count_kernel(const uint16_t *perm_T, uint32_t perm_slots, uint32_t S,
                                                                    uint32_t k, uint32_t n_sh, uint32_t n_games,
                                                                    const uint8_t *patience, const uint32_t *game_block,
                                                                    uint32_t *class_ctr) {
    __shared__ uint32_t cnt[SCHED_CLASSES];
    if (threadIdx.x < SCHED_CLASSES) cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_games; base += gridDim.x * blockDim.x) { // wave-uniform trip count
        const uint32_t t = base + threadIdx.x;
        uint32_t cls = SCHED_CLASSES;
        if (t < n_games) {
            if (game_block) {
                cls = schedule_class_block(patience, game_block[t]);
            } else {
                const uint32_t g_local = t / n_sh, sh_local = t - g_local * n_sh;
                cls = schedule_class(perm_T, perm_slots, S, k, patience, sh_local, g_local);
            }
        }
        for (uint32_t cidx = 0; cidx < SCHED_CLASSES; ++cidx) {
            const uint64_t m = __ballot(cls == cidx);
            if (m && lane_id() == 0u) atomicAdd(&cnt[cidx], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < SCHED_CLASSES && cnt[threadIdx.x]) atomicAdd(&class_ctr[threadIdx.x], cnt[threadIdx.x]);
}

__global__ __launch_bounds__(SEED_BLOCK) void fk_seed_kernel(SeedArgs a) {
    __shared__ uint32_t wave_cnt[SCHED_CLASSES][SEED_BLOCK / 64];
    __shared__ uint32_t block_base[SCHED_CLASSES];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t < a.n_games;
    // Tournament mode walks the games shuffle-minor (consecutive lanes = consecutive shuffles of one game
    // slot) so that the shuffle-minor permutation is read coalesced; ids stay shuffle-major.
    uint32_t id = t, sh_local = 0, g_local = t;
    if (a.perm_T) {
        g_local = t / a.n_sh;
        sh_local = t - g_local * a.n_sh;
        id = sh_local * a.gps + g_local;
    } else if (a.gps) {
        sh_local = t / a.gps;
        g_local = t - sh_local * a.gps;
    }
    // Longest-first schedule (tournament mode; classes above).  fk_class_count_kernel has sized the classes, so a game's
    // ticket is class offset + its rank in the class; ranks come from one returning atomic per block and class (a single
    // word sustains only ~90 returning atomics/us).  The state records are stored at the TICKET position: a wave's 64
    // consecutive tickets then read 64 consecutive records whatever the class mix (stored in walk order, a sparse
    // class dragged a full 128-B line per game through L2: 3.7 GB of HBM fetches per 10^7 games instead of 0.5).
    uint32_t slot = t;
    if (a.sched) {
        uint32_t cls = SCHED_CLASSES;
        if (valid) cls = a.blocks ? schedule_class_block(a.patience, a.game_row[t])
                                  : schedule_class(a.perm_T, a.perm_slots, a.S, a.k, a.patience, sh_local, g_local);
        const uint32_t wave = threadIdx.x >> 6;
        uint64_t mine_m = 0;
        for (uint32_t cidx = 0; cidx < SCHED_CLASSES; ++cidx) {
            const uint64_t m = __ballot(cls == cidx);
            if (cls == cidx) mine_m = m;
            if (lane_id() == 0u) wave_cnt[cidx][wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < SCHED_CLASSES) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < SEED_BLOCK / 64; ++w) {
                const uint32_t c = wave_cnt[threadIdx.x][w];
                wave_cnt[threadIdx.x][w] = total; // exclusive prefix
                total += c;
            }
            uint32_t offset = 0;
            for (uint32_t cidx = 0; cidx < threadIdx.x; ++cidx) offset += a.class_ctr[cidx];
            block_base[threadIdx.x] = offset + (total ? atomicAdd(&a.sched_ctr[threadIdx.x], total) : 0u);
        }
        __syncthreads();
        if (valid) {
            slot = block_base[cls] + wave_cnt[cls][wave] + mbcnt(mine_m);
            a.sched[slot] = id;
        }
    }
    // Phase A, one lane per GAME: the pool after the game's own index word (entropy words 0..13).
    SeedPool gp{};
    uint64_t seat0 = 0, replicate = 0;
    uint32_t blk = 0;
    if (valid) {
        if (a.coords) {
            const fk_coord c = a.coords[id];
            seat0 = c.seat_index;
            replicate = c.replicate_index;
            ss_begin(gp, 2u, c.purpose, (uint32_t)c.root_seed, (uint32_t)(c.root_seed >> 32));
            ss_absorb64(gp, c.k);
            ss_absorb64(gp, c.shuffle_index);
            ss_absorb64(gp, c.pair_id);
            ss_absorb64(gp, c.order);
            ss_absorb64(gp, c.game_index);
        } else if (a.blocks) { // batched H2H: the block holding game t (fk_block_map_kernel), then its attempt index
            blk = a.game_block[t];
            const DevBlock b = a.blocks[blk];
            const uint4 q = a.pools[blk];
            gp.p[0] = q.x, gp.p[1] = q.y, gp.p[2] = q.z, gp.p[3] = q.w;
            gp.hc = HC_AFTER_12_WORDS;
            ss_absorb_index(gp, b.attempt0 + (t - b.start)); // attempt index in the game_index slot (random.py:106-111)
        } else {
            const uint4 q = a.pools[sh_local];
            gp.p[0] = q.x, gp.p[1] = q.y, gp.p[2] = q.z, gp.p[3] = q.w;
            gp.hc = HC_AFTER_12_WORDS;
            ss_absorb_index(gp, a.game0 + g_local);
        }
    }
    // Phase B, one lane per (game, seat) PAIR of the wave's 64 games, pair p = game * k + seat: lanes that are neighbours
    // hold seats that are neighbours in memory (records sit at slot * k + seat), so the 16-byte state / increment stores and
    // the 2-byte index stores of a wave cover whole lines.  (One lane per game and a loop over its seats wrote 16 bytes at
    // a stride of 16 k: partial lines that L2 evicted before their neighbours arrived, 2.2x the algorithmic HBM writes.)
    // The game's values travel from its phase-A lane by ds_bpermute (no LDS allocation), with every lane active.
    const uint32_t lane = lane_id();
    for (uint32_t j = 0; j < a.k; ++j) {
        const uint32_t p = j * 64u + lane;
        const uint32_t src = p / a.k, s = p - src * a.k;
        const int from = (int)src;
        SeedPool sp;
        sp.p[0] = (uint32_t)__shfl((int)gp.p[0], from);
        sp.p[1] = (uint32_t)__shfl((int)gp.p[1], from);
        sp.p[2] = (uint32_t)__shfl((int)gp.p[2], from);
        sp.p[3] = (uint32_t)__shfl((int)gp.p[3], from);
        sp.hc = HC_AFTER_14_WORDS;
        const uint32_t p_slot = (uint32_t)__shfl((int)slot, from), p_id = (uint32_t)__shfl((int)id, from);
        const uint32_t p_sh = (uint32_t)__shfl((int)sh_local, from), p_g = (uint32_t)__shfl((int)g_local, from);
        const uint32_t p_blk = (uint32_t)__shfl((int)blk, from);
        const bool p_valid = __shfl(valid ? 1 : 0, from) != 0;
        uint64_t p_seat0 = 0, p_rep = 0;
        if (a.coords) { // wave-uniform
            p_seat0 = (uint64_t)(uint32_t)__shfl((int)(uint32_t)seat0, from) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(seat0 >> 32), from) << 32);
            p_rep = (uint64_t)(uint32_t)__shfl((int)(uint32_t)replicate, from) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(replicate >> 32), from) << 32);
        }
        if (!p_valid) continue;
        if (a.coords) {
            ss_absorb64(sp, p_seat0 + s); // seat_index
            ss_absorb64(sp, p_rep);       // replicate_index
        } else { // seat s, replicate 0: three of the four words are literal zeros, their hashes constants
            ss_absorb(sp, s);
            ss_absorb(sp, 0u);
            ss_absorb(sp, 0u);
            ss_absorb(sp, 0u);
        }
        uint32_t g8[8];
        ss_generate<8>(sp, g8);
        Rng r;
        pcg_seed(r, g8);
        const size_t rec = (size_t)p_slot * a.k + s; // ticket position (walk order without a schedule)
        uint4 *dst = reinterpret_cast<uint4 *>(a.state + rec * a.state_dw);
        dst[0] = make_uint4((uint32_t)r.lo, (uint32_t)(r.lo >> 32), (uint32_t)r.hi, (uint32_t)(r.hi >> 32));
        if (a.state_dw == STATE_DW) { // full initial record: nothing buffered, score 0, counters 0, strategy index
            uint32_t idx = s;
            if (a.perm_T) idx = perm_at(a.perm_T, a.S, a.perm_slots, p_sh, p_g * a.k + s);
            else if (a.seat_strategy) idx = (uint32_t)a.seat_strategy[(size_t)p_id * a.k + s];
            else if (a.blocks) idx = 2u * a.blocks[p_blk].row + s;
            dst[1] = make_uint4(0u, 0u, 0u, 0u);
            dst[2] = make_uint4(0u, 0u, 0u, idx);
        } else if (a.seat_idx && a.perm_T) {
            a.seat_idx[rec] = perm_at(a.perm_T, a.S, a.perm_slots, p_sh, p_g * a.k + s);
        }
        a.inc[rec] = make_uint4((uint32_t)r.inc_lo, (uint32_t)(r.inc_lo >> 32), (uint32_t)r.inc_hi, (uint32_t)(r.inc_hi >> 32));
    }
}

// resident tally accumulator (fk_tally_resident_*): acc += tally, both int64 on the device
__global__ void fk_add_i64_kernel(unsigned long long *acc, const unsigned long long *src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += src[i];
}

// attempted = completed + safety for every (batch, strategy) row
// Tournament mode: every strategy is seated exactly once per shuffle (S % k == 0), so its attempted exposures in a
// batch equal the batch's shuffle count and only the (rare) safety-limit exposures are counted by the game kernel:
// completed = attempted - safety.  Other modes count completed explicitly: attempted = completed + safety.
__global__ void fk_finalize_tally(unsigned long long *tally, uint32_t n_rows, uint32_t S, uint32_t spb, uint64_t n_sh_total,
                                  uint32_t derive_completed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    unsigned long long *t = tally + (size_t)i * FK_TALLY_COLS;
    if (derive_completed) {
        const uint64_t batch = i / S, first = batch * spb;
        const uint64_t in_batch = first + spb <= n_sh_total ? spb : n_sh_total - first;
        t[1] = in_batch;
        t[2] = in_batch - t[3];
    } else {
        t[1] = t[2] + t[3];
    }
}

// ---------------------------------------------------------------------------------------
// The game kernel.  Every seat has a state record (PCG state, buffered half word, score, packed counters: the ten
// dwords a turn mutates); each roll step loads the turn owner's record from LDS, updates it and stores it back; nothing
// but the turn registers and the owner's read-only data (increment, strategy) is carried in VGPRs across rolls.
//
//   LDS-record instances (GS = false): all k records of the lane's game sit in LDS.  LEAN records keep only the ten
//   mutable dwords (40 bytes instead of 68): the read-only PCG increment and the packed strategy are re-read from the
//   increment plane / the strategy table (L2-resident) at the start of each turn, the strategy index riding in the
//   spare bits of cE.  Fewer LDS bytes per lane = more resident waves per SIMD (k = 2: 4 -> 6).
//
//   State-store instances (GS = true): LDS holds ONE record per lane, the turn owner's; the records of all seats live
//   in the per-game state store in HBM (written by the seed kernel).  A record is touched twice per turn, at the
//   hand-over to the next seat (already a divergent region): the owner's record is stored, the next seat's is loaded
//   — three 16-byte accesses each, L2 / Infinity-Cache resident for the games in flight.  LDS bytes per lane (40) and
//   with them the resident waves per SIMD (6) no longer depend on k, there is no limit on k or on S, and the final
//   records of every game stay in HBM for the streaming post-passes (rows, all-seat statistics).
//
// MIXED: the strategy flag bits that may differ between strategies of the table.  The other flags are the same for
// the whole table (threshold grids fix most of them): they arrive as a kernel argument, so their tests run on the
// scalar unit and the constants they select become s_cselects.  Instances: all flags mixed (generic), none, and
// require_both | favor_score (the pair the reference's grid always enumerates).
// BLK: batched-H2H instance (MODE_BLOCKS with lean LDS records): the strategy index comes from the lane's block index.
// KC: compile-time player count (0 = run-time a.k).  The two-player instances (BASELINE config 2, every H2H launch) fold
// the seat arithmetic of the table advance and the record addressing.
template <int BLOCK, bool LEAN, int WPE, uint32_t MIXED, bool GS, bool BLK = false, int KC = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPE))) void fk_play_kernel(PlayArgs a) {
    static_assert(!GS || LEAN, "state-store instances stage the lean record");
    static_assert(!BLK || (LEAN && !GS), "block-index instances use lean LDS records");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t K = KC ? (uint32_t)KC : a.k;
    constexpr uint32_t NFIELDS = LEAN ? LEAN_DW : (uint32_t)NF; // 10 or 17 dwords per seat record
    unsigned long long *tl = reinterpret_cast<unsigned long long *>(lds + NFIELDS * (GS ? 1u : K) * BLOCK);
    // batched H2H launches (no LDS tally): one dword per lane behind the records holds the lane's block index — the strategy
    // index of seat s is 2 * block + s, whatever the number of blocks (the 14-bit index field of cE would cap it at 8 192)
    uint32_t *lane_block = lds + NFIELDS * (GS ? 1u : K) * BLOCK + tid;
    clock_stamp(a.clk, 0u);

    if (a.use_lds_tally) {
        for (uint32_t i = tid; i < a.S * LT_COLS; i += BLOCK) tl[i] = 0ull;
        __syncthreads();
    }

    enum : uint32_t { ST_FRESH = 0, ST_ACTIVE = 1, ST_ENDED = 2, ST_DONE = 3 };
    uint32_t st = ST_FRESH;
    uint32_t pool_next = 0, pool_end = 0, exhausted = 0; // wave-uniform ticket pool

    // game registers
    uint32_t game_id = 0, seat = 0, rounds = 0, max_rounds = 0, trigger = 0, seed_slot = 0;
    uint32_t final_round = 0, safety = 0;
    int32_t score_to_beat = 0;
    // turn registers
    uint32_t dice = 6, rolls_this_turn = 0;
    int32_t turn_score = 0;
    uint32_t own_turns = 0; // state-store instances only: the owner's n_turns, kept out of the 10-dword LDS record
    // read-only data of the turn owner (PCG increment, packed strategy): the only per-seat values carried in registers
    // across roll iterations.  The mutable seat record (generator state, score, counters) is loaded from and stored to
    // LDS inside every roll step, so the hot loop carries no per-seat PHIs through its divergent turn hand-over.
    uint64_t own_inc_lo = 0, own_inc_hi = 0;
    int32_t own_thr = 0;
    uint32_t own_bits = 0;
    // two-player lean instances (KC = 2; not state-store): the packed strategies of both seats stay in registers for the game,
    // so a turn start is one increment load instead of LDS read -> index -> strategy load
    constexpr bool PK2 = (KC == 2) && LEAN && !GS;
    uint2 pk_seat0 = make_uint2(0u, 0u), pk_seat1 = make_uint2(0u, 0u);

    // LDS records are contiguous per (seat, lane): record base = (seat * BLOCK + tid) * NFIELDS, field = immediate
    // offset (one address VGPR per record).  Full records: odd stride (17 dwords), ds_read2/ds_write2 dword pairs, the
    // 32 lanes of an LDS lane group on 32 distinct banks whatever seat each lane is on (BLOCK % 32 == 0).  LEAN records:
    // ten dwords = five 8-byte-aligned pairs moved by ds_read_b64 / ds_write_b64; the stride of five bank PAIRS is odd,
    // so the 32 lanes of a group again cover all 64 banks once.
    // LEAN records have no increment / strategy / score slots: BUF moves up by four, the counters by five.
    // Address = loop-invariant lane base + seat * compile-time stride: one full-rate v_mad_u32_u24 per record instead
    // of the quarter-rate 32-bit multiplies the plain index expression costs.  GS: one record per lane, no seat term.
    const uint32_t lane_base = tid * NFIELDS;
    constexpr uint32_t SEAT_STRIDE = (uint32_t)BLOCK * NFIELDS; // < 2^24
    auto L = [&](uint32_t field, uint32_t s) __attribute__((always_inline)) -> uint32_t & {
        const uint32_t f = !LEAN ? field : (field > F_SCORE) ? field - 5u : (field > F_INC_HI1) ? field - 4u : field;
        if (GS) return lds[lane_base + f];
        return lds[__umul24(s, SEAT_STRIDE) + lane_base + f];
    };
    // state record of seat s of the lane's game in HBM
    auto G = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t * {
        return a.state + ((size_t)seed_slot * K + s) * a.state_dw;
    };

    auto strategy_index = [&](uint32_t id, uint32_t s) -> uint32_t {
        if (a.mode == MODE_PERM) {
            const uint32_t sh = id / a.gps, g = id - sh * a.gps;
            return perm_at(a.perm_T, a.S, a.perm_slots, sh, g * K + s);
        }
        if (a.mode == MODE_LIST) return (uint32_t)a.seat_strategy[(size_t)id * K + s];
        return 2u * a.game_block[id] + s; // MODE_BLOCKS
    };

    // per-seat views used by the end-of-game code (seat s may be the turn owner or not)
    auto seat_strategy = [&](uint32_t s) -> uint32_t { // strategy-table index of seat s of the lane's current game
        if (GS) return G(s)[R_IDX];
        if (BLK) return 2u * *lane_block + s;
        if (LEAN) return L(F_CE, s) >> CE_IDX_SHIFT;
        return strategy_index(game_id, s);
    };
    auto seat_score = [&](uint32_t s) -> int32_t {
        return (int32_t)(GS ? G(s)[R_SCORE] : LEAN ? (L(F_CE, s) & 0xffffu) : L(F_SCORE, s));
    };
    auto seat_counter = [&](uint32_t s, uint32_t field) -> uint32_t { // field in F_CA..F_CD (same packing in every layout)
        return GS ? G(s)[R_CA + (field - F_CA)] : L(field, s);
    };
    auto seat_hot_dice = [&](uint32_t s) -> uint32_t {
        return GS ? (G(s)[R_CE] & 0xffffu) : LEAN ? (L(F_CB, s) >> 16) : (L(F_CE, s) & 0xffffu);
    };
    // n_turns of seat s of a game that has ended (engine.py:236 counts them turn by turn): every seat began `rounds` turns,
    // minus the seats behind the trigger in its round, plus the final-round turn of every seat but the trigger's.
    auto seat_turns = [&](uint32_t s) -> uint32_t { return rounds + ((final_round != 0u && s < trigger) ? 1u : 0u); };

    // turn owner := seat s (engine.py:236-240): fresh turn registers, read-only data.  n_turns is not stored per turn in
    // the LDS instances (seat_turns above restores it when the game ends).
    auto begin_turn = [&](uint32_t s) __attribute__((always_inline)) {
        uint32_t idx = 0;
        if (GS) { // stage the seat's record: HBM state store -> the lane's LDS record
            const uint4 *g = reinterpret_cast<const uint4 *>(G(s));
            const uint4 q0 = g[0], q1 = g[1], q2 = g[2];
            uint2 *r = reinterpret_cast<uint2 *>(lds + lane_base);
            r[0] = make_uint2(q0.x, q0.y);
            r[1] = make_uint2(q0.z, q0.w);
            r[2] = make_uint2(q1.x, q1.z);                                           // buf, cA
            r[3] = make_uint2((q1.w & 0xffffu) | (q2.z << 16), q2.x);                 // highest_turn | hot_dice << 16, cC
            r[4] = make_uint2(q2.y, (q1.y & 0xffffu) | (q2.z & (CE_HAS_SCORED | CE_HAS_BUF))); // cD, score | flags
            own_turns = (q1.w >> 16) + 1u; // n_turns += 1 (engine.py:236)
            idx = q2.w;
        } else if (!PK2) {
            if (BLK) idx = 2u * *lane_block + s;
            else if (LEAN) idx = L(F_CE, s) >> CE_IDX_SHIFT;
        }
        if (LEAN) { // read-only per-seat data comes from HBM/L2; the loads overlap the first dice of the turn
            const uint4 inc = a.inc[(size_t)seed_slot * K + s];
            const uint2 pk = PK2 ? (s ? pk_seat1 : pk_seat0) : a.strat[idx];
            own_inc_lo = (uint64_t)inc.x | ((uint64_t)inc.y << 32);
            own_inc_hi = (uint64_t)inc.z | ((uint64_t)inc.w << 32);
            own_thr = (int32_t)pk.x;
            own_bits = pk.y;
        } else {
            own_inc_lo = (uint64_t)L(F_INC_LO0, s) | ((uint64_t)L(F_INC_LO1, s) << 32);
            own_inc_hi = (uint64_t)L(F_INC_HI0, s) | ((uint64_t)L(F_INC_HI1, s) << 32);
            own_thr = (int32_t)L(F_SPX, s);
            own_bits = L(F_SPY, s);
        }
        dice = 6;
        turn_score = 0;
        rolls_this_turn = 0;
    };

    auto raise = [&](int32_t code) {
        if (atomicCAS(&a.err[0], 0, code) == 0) a.err[1] = (int32_t)game_id;
        st = ST_DONE;
    };

    // ---- finished game -> LDS tallies or one result record (run_tournament.py:375-391) ----
    auto finish_game = [&]() {
        const bool completed = (safety == 0u);
        uint32_t w = 0;
        int32_t best = seat_score(0);
        for (uint32_t s = 1; s < K; ++s) { // stable sort on score desc: first maximum wins (engine.py:477)
            const int32_t sc = seat_score(s);
            if (sc > best) {
                best = sc;
                w = s;
            }
        }
        if (!GS && a.gs_out) { // final records of every seat -> state store (rows / all-seat statistics post-passes)
            for (uint32_t s = 0; s < K; ++s) { // the state store's format (R_*): score and n_turns spelled out
                uint4 *g = reinterpret_cast<uint4 *>(G(s));
                g[0] = make_uint4(L(F_LO0, s), L(F_LO1, s), L(F_HI0, s), L(F_HI1, s));
                g[1] = make_uint4(L(F_BUF, s), (uint32_t)seat_score(s), L(F_CA, s), (L(F_CB, s) & 0xffffu) | (seat_turns(s) << 16));
                g[2] = make_uint4(L(F_CC, s), L(F_CD, s), seat_hot_dice(s) | (L(F_CE, s) & (CE_HAS_SCORED | CE_HAS_BUF)),
                                  seat_strategy(s));
            }
        }
        if (a.use_lds_tally) {
            // exposures: tournament mode counts only safety-limit exposures (completed is derived in fk_finalize_tally)
            if (!completed || a.mode != MODE_PERM) {
                for (uint32_t s = 0; s < K; ++s) atomicAdd(&tl[seat_strategy(s) * LT_COLS + (completed ? 1u : 2u)], 1ull);
            }
        }
        if (!(a.use_lds_tally && completed) && !a.rec0) return;
        uint32_t widx = 0, wa = 0, wb = 0, wc = 0, wd = 0, we = 0;
        if (completed) {
            widx = seat_strategy(w);
            wa = seat_counter(w, F_CA), wb = seat_counter(w, F_CB), wc = seat_counter(w, F_CC), wd = seat_counter(w, F_CD),
            we = seat_hot_dice(w);
        } else if (a.mode == MODE_BLOCKS) {
            widx = seat_strategy(0); // names the block of a safety-limit attempt
        }
        if (a.use_lds_tally && completed) {
            const unsigned long long m[10] = {(unsigned long long)(uint32_t)best * 50u, rounds, wa >> 16, wa & 0xffffu,
                                              (wb & 0xffffu) * 50u, wc & 0xffffu, wc >> 16, wd & 0xffffu, wd >> 16, we};
            unsigned long long *t = tl + widx * LT_COLS;
            atomicAdd(&t[0], 1ull);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (m[j]) { // zero-valued metrics (e.g. smart-discard counters of non-smart winners) add nothing
                    atomicAdd(&t[3 + j], m[j]);
                    atomicAdd(&t[13 + j], m[j] * m[j]);
                }
            }
        }
        if (a.rec0) {
            const uint32_t d0 = widx | (completed ? (w << 24) : REC_SAFETY);
            a.rec0[game_id] = d0;
            if (a.recs) {
                uint4 *r = reinterpret_cast<uint4 *>(a.recs + (size_t)game_id * REC_DW);
                r[0] = make_uint4(d0, completed ? (uint32_t)best * 50u : 0u, rounds | (wa & 0xffff0000u),
                                  (wa & 0xffffu) | (((wb & 0xffffu) * 50u) << 16)); // points: highest_turn <= 65 500 by its guard band
                r[1] = make_uint4(wc, wd, we, 0u);
            }
        }
    };

    // ---- fresh game for this lane ----
    auto init_game = [&](uint32_t id, uint32_t ticket) {
        game_id = id;
        max_rounds = a.max_rounds;
        if (a.n_ov) { // sorted by game id: binary search
            uint32_t lo = 0, hi = a.n_ov;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (a.ov[mid].game < id) lo = mid + 1u;
                else hi = mid;
            }
            if (lo < a.n_ov && a.ov[lo].game == id) max_rounds = a.ov[lo].max_rounds;
        }
        // state records sit at the ticket position when there is a schedule, else in the seed kernel's walk order
        // (shuffle-minor in tournament mode)
        uint32_t slot = a.sched ? ticket : id;
        if (!a.sched && a.mode == MODE_PERM) {
            const uint32_t sh = id / a.gps, g = id - sh * a.gps;
            slot = g * a.n_sh + sh;
        }
        seed_slot = slot;
        if (!GS) {
            if (BLK) *lane_block = a.game_block[id];
            for (uint32_t s = 0; s < K; ++s) {
                const uint32_t *src = G(s);
                const uint4 stv = *reinterpret_cast<const uint4 *>(src);
                const uint32_t idx = (a.state_dw == STATE_DW) ? src[R_IDX]
                                     : a.seat_idx            ? (uint32_t)a.seat_idx[(size_t)slot * K + s]
                                                             : strategy_index(id, s);
                L(F_LO0, s) = stv.x;
                L(F_LO1, s) = stv.y;
                L(F_HI0, s) = stv.z;
                L(F_HI1, s) = stv.w;
                if (!LEAN) {
                    const uint4 inc = a.inc[(size_t)slot * K + s];
                    const uint2 pk = a.strat[idx];
                    L(F_INC_LO0, s) = inc.x;
                    L(F_INC_LO1, s) = inc.y;
                    L(F_INC_HI0, s) = inc.z;
                    L(F_INC_HI1, s) = inc.w;
                    L(F_SPX, s) = pk.x;
                    L(F_SPY, s) = pk.y;
                }
                L(F_BUF, s) = 0u;
                if (!LEAN) L(F_SCORE, s) = 0u;
                L(F_CA, s) = 0u;
                L(F_CB, s) = 0u;
                L(F_CC, s) = 0u;
                L(F_CD, s) = 0u;
                L(F_CE, s) = (LEAN && !BLK) ? (idx << CE_IDX_SHIFT) : 0u;
                if (PK2) {
                    const uint2 pk = a.strat[BLK ? 2u * a.game_block[id] + s : idx];
                    if (s == 0u) pk_seat0 = pk;
                    else pk_seat1 = pk;
                }
            }
        }
        seat = 0;
        trigger = 0;
        final_round = 0;
        safety = 0;
        score_to_beat = a.beat50; // engine.py:451 (units of 50)
        if (max_rounds == 0u) {   // `while rounds < max_rounds` never entered (engine.py:453)
            rounds = 0;
            safety = 1;
            st = ST_ENDED;
        } else {
            rounds = 1;
            begin_turn(0);
            st = ST_ACTIVE;
        }
    };

    // ---- after a turn: advance the table (engine.py:453-472, 523-550); `score` is the owner's banked total.
    // Written as selects: one predicated region per roll step instead of a tree of them. ----
    auto advance = [&](int32_t score) __attribute__((always_inline)) {
        const bool fr = final_round != 0u;
        const bool trig = !fr & (score >= a.target50);         // first trigger starts the final round (engine.py:462-468)
        const bool normal = !fr & !trig;
        const uint32_t n1 = seat + 1u;
        const bool wrap = n1 == K;
        const bool last = normal & wrap & (rounds >= max_rounds); // `while rounds < max_rounds` ends (engine.py:453, 472)
        uint32_t next_fr = n1 + ((n1 == trigger) ? 1u : 0u); // final round skips the trigger seat (engine.py:523-550)
        uint32_t next_tr = (seat == 0u) ? 1u : 0u, next_nm = wrap ? 0u : n1;
        // pin the three candidates in registers: the compiler otherwise sinks them into nested exec-mask regions (a dozen
        // scalar instructions per trip; measured -0.7 % kernel time with two plain selects)
        asm volatile("" : "+v"(next_fr), "+v"(next_tr), "+v"(next_nm));
        const uint32_t next = fr ? next_fr : trig ? next_tr : next_nm;
        rounds += (normal & wrap & !last) ? 1u : 0u;
        safety = last ? 1u : safety;
        score_to_beat = trig ? score : (fr & (score > score_to_beat)) ? score : score_to_beat; // engine.py:464, 547
        trigger = trig ? seat : trigger;
        final_round = (fr | trig) ? 1u : 0u;
        const bool ended = last | ((fr | trig) & (next >= K));
        if (ended) {
            st = ST_ENDED;
        } else {
            seat = next;
            begin_turn(next);
        }
    };

    // ---- two-player tables (PK2): the table advance as straight-line selects for EVERY lane of the trip ----
    // Some lane of a 64-lane wave ends its turn in nearly every trip, so `advance` above — an exec-masked region of ~47
    // instructions for the third of the lanes whose turn ended — is paid by the whole wave every trip.  With two seats the
    // rules collapse (engine.py:453-472, 523-550): the next seat is always the other one; a turn played in the final round
    // is the last of the game; the first score at or above the target starts the final round.  That is ~20 predicated
    // instructions with no region; only the next owner's increment load stays under the exec mask.
    auto advance2 = [&](bool over, int32_t score) __attribute__((always_inline)) {
        const bool fr = final_round != 0u;
        const bool trig = over & !fr & (score >= a.target50);                 // engine.py:462-468
        const bool close = over & !fr & !trig & (seat != 0u);                 // seat 1 closes a normal round
        const bool last = close & (rounds >= max_rounds);                     // `while rounds < max_rounds` ends (engine.py:453, 472)
        const bool ended = last | (over & fr);                                // the final round's one turn has been played
        rounds += (close & !last) ? 1u : 0u;
        safety = last ? 1u : safety;
        score_to_beat = trig ? score : score_to_beat;                         // engine.py:464
        trigger = trig ? seat : trigger;
        final_round = (fr | trig) ? 1u : 0u;
        const bool sw = over & !ended;
        st = ended ? (uint32_t)ST_ENDED : st;
        seat ^= sw ? 1u : 0u;
        if (sw) { // the new owner's increment: the one memory request of a turn start
            const uint4 inc = a.inc[(size_t)seed_slot * 2u + seat];
            own_inc_lo = (uint64_t)inc.x | ((uint64_t)inc.y << 32);
            own_inc_hi = (uint64_t)inc.z | ((uint64_t)inc.w << 32);
        }
        own_thr = (int32_t)(seat ? pk_seat1.x : pk_seat0.x);
        own_bits = seat ? pk_seat1.y : pk_seat0.y;
        dice = sw ? 6u : dice;
        turn_score = sw ? 0 : turn_score;
        rolls_this_turn = sw ? 0u : rolls_this_turn;
    };

    // ---- one roll of the current turn (engine.py:241-273): record in, roll, score, decide, record out ----
    auto roll_step = [&]() __attribute__((always_inline)) {
        const bool roll_limit = rolls_this_turn >= 1000u; // ROLL_LIMIT, engine.py:36,242 (raised below, before any store)
        const uint32_t s = seat;
        uint32_t cA, cB, cC, cD, cE, buf0;
        int32_t score;
        uint64_t lo0, hi0;
        uint2 *const rec = reinterpret_cast<uint2 *>(&L(F_LO0, s)); // LEAN: five aligned pairs
        if (LEAN) {
            const uint2 p0 = rec[0], p1 = rec[1], p2 = rec[2], p3 = rec[3], p4 = rec[4];
            lo0 = (uint64_t)p0.x | ((uint64_t)p0.y << 32);
            hi0 = (uint64_t)p1.x | ((uint64_t)p1.y << 32);
            buf0 = p2.x, cA = p2.y, cB = p3.x, cC = p3.y, cD = p4.x, cE = p4.y;
            score = (int32_t)(cE & 0xffffu); // score / 50 rides in cE[15:0] (plan_play keeps target / 50 + a turn below 2^16)
        } else {
            cA = L(F_CA, s), cB = L(F_CB, s), cC = L(F_CC, s), cD = L(F_CD, s), cE = L(F_CE, s);
            score = (int32_t)L(F_SCORE, s);
            lo0 = (uint64_t)L(F_LO0, s) | ((uint64_t)L(F_LO1, s) << 32);
            hi0 = (uint64_t)L(F_HI0, s) | ((uint64_t)L(F_HI1, s) << 32);
            buf0 = L(F_BUF, s);
        }
        Rng rng{hi0, lo0, own_inc_hi, own_inc_lo, buf0, (cE & CE_HAS_BUF) ? 1u : 0u};
        const uint32_t n = dice;
        bool detour;
        uint32_t key = roll_counts_fast<3>(rng, n, detour);
        if (detour) { // a Lemire rejection (once in ~2^30 dice): the generator comes back from the seat record, which is still the roll's input
            asm volatile("" ::: "memory"); // really re-read it: values forwarded from the loads above would stay live across the whole roll
            rng.lo = (uint64_t)L(F_LO0, s) | ((uint64_t)L(F_LO1, s) << 32);
            rng.hi = (uint64_t)L(F_HI0, s) | ((uint64_t)L(F_HI1, s) << 32);
            rng.buf = L(F_BUF, s);
            rng.has_buf = (cE & CE_HAS_BUF) ? 1u : 0u;
            key = roll_counts_sequential<3>(rng, n, nullptr);
        }
        rolls_this_turn += 1u;
        int32_t dthr = (int32_t)(int8_t)(own_bits & 0xffu);
        asm volatile("" : "+v"(dthr));
        const Strat50 sp{own_thr, (own_bits & (0xffu | MIXED)) | (a.uflags & (0xff00u & ~MIXED)), dthr};
        const Roll50 rr = default_score_lut50(a.score_lut, a.discard_lut, key, (int32_t)n, turn_score, sp); // turn_score, score: / 50
        const bool farkle = rr.score50 == 0;                            // engine.py:135-137, 247-249
        cA += 1u + (farkle ? 0x10000u : 0u);                            // n_rolls (engine.py:98), n_farkles
        cC += (rr.d5 > 0) ? (1u + ((uint32_t)rr.d5 << 16)) : 0u;        // engine.py:139-144
        cD += (rr.d1 > 0) ? (1u + ((uint32_t)rr.d1 << 16)) : 0u;
        dice = (rr.used == (int32_t)n) ? 6u : (n - (uint32_t)rr.used);  // engine.py:146
        turn_score = farkle ? 0 : (turn_score + rr.score50);
        const bool hot = !farkle & sp.has(SF_AUTO_HOT) & (dice == 6u);  // _apply_hot_dice, engine.py:149-154, 253
        if (LEAN) cB += hot ? 0x10000u : 0u; // hot-dice count: cB[31:16] (lean) or cE[15:0]
        else cE += hot ? 1u : 0u;
        const bool keep = should_continue50(sp, turn_score, (int32_t)dice, (cE & CE_HAS_SCORED) != 0u, final_round != 0u,
                                            score_to_beat, score);
        const bool over = farkle | (!hot & !keep);
        // bank (engine.py:265-273), branch-free: a farkled turn has turn_score 0 and changes nothing
        const uint32_t ts = over ? (uint32_t)turn_score : 0u;
        cE |= (ts >= 10u) ? CE_HAS_SCORED : 0u;                         // 500 points
        const uint32_t banked = (cE & CE_HAS_SCORED) ? ts : 0u;
        score += (int32_t)banked;
        if (LEAN) cE += banked;
        cB = (banked > (cB & 0xffffu)) ? ((cB & 0xffff0000u) | banked) : cB;
        // one rare exit for all error conditions: the roll limit, then the u16 guard bands (a turn adds <= 1000 rolls
        // and <= 2000 discarded dice; highest_turn must fit 16 bits IN POINTS: 1310 x 50 = 65 500)
        const bool overflow = (turn_score > 1310) | ((cA & 0xffffu) > 64000u) | ((cC >> 16) > 63000u) | ((cD >> 16) > 63000u);
        if (roll_limit | overflow) {
            raise(roll_limit ? FK_ERR_ROLL_LIMIT : FK_ERR_COUNTER_OVERFLOW);
            return;
        }
        cE = (cE & ~CE_HAS_BUF) | (rng.has_buf ? CE_HAS_BUF : 0u);
        if (GS) {
            if (over) { // the turn is over: the record goes back to the state store, the next seat's comes in
                uint4 *g = reinterpret_cast<uint4 *>(G(s));
                g[0] = make_uint4((uint32_t)rng.lo, (uint32_t)(rng.lo >> 32), (uint32_t)rng.hi, (uint32_t)(rng.hi >> 32));
                g[1] = make_uint4(rng.buf, (uint32_t)score, cA, (cB & 0xffffu) | (own_turns << 16));
                uint32_t *g2 = reinterpret_cast<uint32_t *>(g + 2); // R_IDX stays as the seed kernel wrote it
                g2[0] = cC;
                g2[1] = cD;
                g2[2] = (cB >> 16) | (cE & (CE_HAS_SCORED | CE_HAS_BUF));
                advance(score);
                return;
            }
        }
        if (LEAN) {
            rec[0] = make_uint2((uint32_t)rng.lo, (uint32_t)(rng.lo >> 32));
            rec[1] = make_uint2((uint32_t)rng.hi, (uint32_t)(rng.hi >> 32));
            rec[2] = make_uint2(rng.buf, cA);
            rec[3] = make_uint2(cB, cC);
            rec[4] = make_uint2(cD, cE);
        } else {
            L(F_LO0, s) = (uint32_t)rng.lo;
            L(F_LO1, s) = (uint32_t)(rng.lo >> 32);
            L(F_HI0, s) = (uint32_t)rng.hi;
            L(F_HI1, s) = (uint32_t)(rng.hi >> 32);
            L(F_BUF, s) = rng.buf;
            L(F_SCORE, s) = (uint32_t)score;
            L(F_CA, s) = cA;
            L(F_CB, s) = cB;
            L(F_CC, s) = cC;
            L(F_CD, s) = cD;
            L(F_CE, s) = cE;
        }
        if (PK2) advance2(over, score);
        else if (!GS && over) advance(score);
    };

    // ---- wave-level hand-over: finish ended games, deal new tickets ----
    auto handover = [&](uint64_t waiting) {
        const bool mine = (st == ST_FRESH || st == ST_ENDED);
        if (st == ST_ENDED) finish_game();
        const uint32_t n = (uint32_t)__popcll(waiting);
        const uint32_t avail = pool_end - pool_next;
        uint32_t new_base = 0, new_avail = 0;
        if (avail < n && !exhausted) {
            // v_readlane makes the pool registers provably wave-uniform, so the loops below branch on SGPRs
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(__ffsll((long long)waiting) - 1);
            uint32_t base = 0;
            if (mine && lane_id() == first) base = atomicAdd(a.ticket, TICKET_CHUNK);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
            if (base >= a.n_games) {
                exhausted = 1;
            } else {
                new_base = base;
                new_avail = min(TICKET_CHUNK, a.n_games - base);
                if (new_avail < TICKET_CHUNK) exhausted = 1;
            }
        }
        if (mine) {
            const uint32_t rank = mbcnt(waiting);
            uint32_t ticket = 0xffffffffu;
            if (rank < avail) ticket = pool_next + rank;
            else if (rank - avail < new_avail) ticket = new_base + (rank - avail);
            if (ticket != 0xffffffffu) init_game(a.sched ? a.sched[ticket] : ticket, ticket);
            else st = ST_DONE;
        }
        if (n <= avail) {
            pool_next += n;
        } else {
            const uint32_t used_new = min(n - avail, new_avail);
            pool_next = new_base + used_new;
            pool_end = new_base + new_avail;
        }
    };

    // Two nested loops.  The inner one is the hot roll loop: a bottom-tested loop with a single back edge whose exit
    // test is wave-uniform (ballots and the v_readlane'd ticket pool), so its loop-carried registers stay put (no PHI
    // copies, no full s_waitcnt at a merge point).  The rare hand-over sits on the outer back edge.
    auto handover_due = [&](uint64_t waiting, uint64_t active) -> bool {
        return waiting && (!active || (uint32_t)__popcll(waiting) >= a.batch_threshold || exhausted);
    };
    while (true) {
        uint64_t waiting = __ballot(st == ST_FRESH || st == ST_ENDED);
        uint64_t active = __ballot(st == ST_ACTIVE);
        if (!(waiting | active)) break; // no lane is active and none waits: the wave has drained
        if (handover_due(waiting, active)) {
            handover(waiting);
            continue;
        }
        do {
            if (st == ST_ACTIVE) roll_step();
            waiting = __ballot(st == ST_ENDED);
            active = __ballot(st == ST_ACTIVE);
        } while (active && !handover_due(waiting, active));
    }

    if (a.use_lds_tally) {
        __syncthreads();
        for (uint32_t i = tid; i < a.S * LT_COLS; i += BLOCK) {
            const unsigned long long v = tl[i];
            if (v == 0ull) continue;
            const uint32_t idx = i / LT_COLS, c = i - idx * LT_COLS;
            if (c == LT_COLS - 1u) continue;
            const uint32_t col = (c == 0u) ? 0u : (c == 1u) ? 2u : (c == 2u) ? 3u : (c < 13u) ? (c + 1u) : (c + 2u);
            atomicAdd(&a.tally[(size_t)idx * FK_TALLY_COLS + col], v);
        }
    }
    clock_stamp(a.clk, 1u);
}

#include "fk_play_hc.h" // the hot / cold variant of the game kernel (k >= 3 seats)

// ---------------------------------------------------------------------------------------
// Lag sufficient statistics of the RNG diagnostics, strategy family (analysis/rng_diagnostics.py:1870-1905 observation
// records, :2031-2076 _OnlineMetric): a strategy is seated exactly once per shuffle, so its series (win indicator, n_rounds)
// is indexed by the shuffle index; for every lag L the six sums (pairs, sum x, sum y, sum x^2, sum y^2, sum xy; x = the
// EARLIER value) are shifted element-wise products over a [shuffle][strategy] matrix.
//   fk_lag_values_kernel: one lane per (game, seat): V[row0 + shuffle][strategy] = n_rounds | won << 15 (u16; coalesced
//                         reads of the result records, scattered 2-byte stores inside one S-wide row)
//   fk_lag_sums_kernel:   thread = (strategy, segment of rows); consecutive threads read consecutive strategies of a row
//                         (coalesced); the earlier value of a pair is re-read `lag` rows up (L2); per-lag accumulators in
//                         registers, one set of int64 atomics per (thread, lag)
// Rows [0, carry) of V hold the last shuffles of the previous chunk of the call, so pairs across chunks are counted here;
// pairs across CALLS (ranks, launch groups) are the host's: every call returns the first and last max_lag rows of its range.
constexpr uint32_t LAG_WIN_BIT = 0x8000u;

__global__ __launch_bounds__(256) void fk_lag_values_kernel(const uint32_t *recs, const uint16_t *perm_T, uint32_t perm_slots, uint32_t S,
                                                            uint32_t k, uint32_t gps, uint32_t n_games, uint16_t *V) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_games * k) return;
    const uint32_t id = (uint32_t)(t / k), seat = (uint32_t)(t - (size_t)id * k);
    const uint32_t sh = id / gps, g = id - sh * gps;
    const uint4 q0 = *reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
    const bool won = !(q0.x & REC_SAFETY) && ((q0.x >> 24) & 0x7fu) == seat; // safety-limit rows have no winner (:1129-1147)
    const uint32_t idx = perm_at(perm_T, S, perm_slots, sh, g * k + seat);
    V[(size_t)sh * S + idx] = (uint16_t)((q0.z & 0x7fffu) | (won ? LAG_WIN_BIT : 0u));
}

// rows [first_row, n_rows) are this chunk's shuffles (later elements of the pairs), rows [valid_from, first_row) the carry
__global__ __launch_bounds__(256) void fk_lag_sums_kernel(const uint16_t *V, uint32_t S, uint32_t valid_from, uint32_t first_row, uint32_t n_rows,
                                                          uint32_t rows_per_seg, const int32_t *lags, uint32_t n_lags, long long *out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t r0 = first_row + blockIdx.y * rows_per_seg, r1 = min(r0 + rows_per_seg, n_rows);
    for (uint32_t li = 0; li < n_lags; ++li) {
        const uint32_t lag = (uint32_t)lags[li];
        long long n = 0, wx = 0, wy = 0, wxy = 0, rx = 0, ry = 0, rxx = 0, ryy = 0, rxy = 0;
        for (uint32_t r = max(r0, valid_from + lag); r < r1; ++r) {
            const uint32_t later = V[(size_t)r * S + s], earlier = V[(size_t)(r - lag) * S + s];
            const long long a = earlier & 0x7fffu, b = later & 0x7fffu, wa = earlier >> 15, wb = later >> 15;
            n += 1;
            wx += wa, wy += wb, wxy += wa & wb;
            rx += a, ry += b, rxx += a * a, ryy += b * b, rxy += a * b;
        }
        if (n == 0) continue;
        unsigned long long *o = reinterpret_cast<unsigned long long *>(out) + ((size_t)s * n_lags + li) * FK_LAG_COLS;
        const long long v[FK_LAG_COLS] = {n, wx, wy, wx, wy, wxy, rx, ry, rxx, ryy, rxy}; // (an indicator is its own square)
#pragma unroll
        for (int c = 0; c < FK_LAG_COLS; ++c)
            if (v[c]) atomicAdd(&o[c], (unsigned long long)v[c]);
    }
}

// ---------------------------------------------------------------------------------------
// Post-passes over the result records / the state store (streaming kernels, one thread per game).
// ---------------------------------------------------------------------------------------

// chunk-local game id -> schedule slot of its state records (inverse of the seed kernel's dealing order)
__device__ inline uint32_t walk_slot(uint32_t id, uint32_t gps, uint32_t n_sh, bool perm_mode) {
    if (!perm_mode) return id;
    const uint32_t sh = id / gps, g = id - sh * gps;
    return g * n_sh + sh;
}

// Result records -> tally [n_batches][S][26].  Games are in id order = shuffle order, so a deterministic batch is a
// contiguous record range; a workgroup takes one part of one batch and one SLICE of the strategy axis, accumulates the
// slice in LDS (ds_add_u64) and flushes what is non-zero with contiguous global atomics: the 4-byte rec0 word of every game
// is read ceil(S / slice) times (coalesced), its 32-byte record once — by the slice that holds the winner; nothing is
// atomically added to HBM per game.
//   grid = (parts_per_batch * n_batches, n_slices); completed games count wins + 10 sums + 10 square sums for the winner,
//   safety-limit games one safety exposure per seat (strategies re-derived from the permutation).
constexpr uint32_t REDUCE_BLOCK = 1024, RT_COLS = 22; // wins, safety, 10 sums, 10 square sums

__global__ __launch_bounds__(REDUCE_BLOCK) void fk_tally_reduce_kernel(const uint32_t *rec0, const uint32_t *recs, uint32_t n_games, uint32_t gps, uint32_t k,
                                                                       uint32_t S, const uint16_t *perm_T, uint32_t perm_slots,
                                                                       uint32_t sh_offset, uint32_t spb, uint32_t n_sh,
                                                                       uint32_t parts_per_batch, uint32_t slice, uint32_t first_batch,
                                                                       unsigned long long *tally) {
    extern __shared__ unsigned long long rt[];
    const uint32_t b_local = blockIdx.x / parts_per_batch, part = blockIdx.x - b_local * parts_per_batch;
    const uint32_t batch = first_batch + b_local;
    // shuffles of this batch inside the chunk: [sh_lo, sh_hi) chunk-local
    const uint64_t g_lo = (uint64_t)batch * spb, g_hi = g_lo + spb;
    const uint32_t sh_lo = g_lo > sh_offset ? (uint32_t)(g_lo - sh_offset) : 0u;
    const uint32_t sh_hi = (uint32_t)min<uint64_t>(n_sh, g_hi > sh_offset ? g_hi - sh_offset : 0u);
    if (sh_hi <= sh_lo) return;
    const uint32_t games = (sh_hi - sh_lo) * gps, per_part = (games + parts_per_batch - 1u) / parts_per_batch;
    const uint32_t first = sh_lo * gps + part * per_part, last = min(first + per_part, sh_lo * gps + games);
    const uint32_t s_lo = blockIdx.y * slice, s_n = min(slice, S - s_lo);
    for (uint32_t i = threadIdx.x; i < s_n * RT_COLS; i += REDUCE_BLOCK) rt[i] = 0ull;
    __syncthreads();
    // four games per thread and trip: the four rec0 words, then the (up to four) 32-byte records of this slice's winners, are
    // in flight together before the first LDS atomic (one game per trip left a single dependent load pair per thread)
    constexpr uint32_t U = 4;
    for (uint32_t base = first + threadIdx.x; base < last; base += U * REDUCE_BLOCK) {
        uint32_t d0[U];
        uint4 q0[U], q1[U];
        bool mine[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t id = base + u * REDUCE_BLOCK;
            d0[u] = id < last ? rec0[id] : REC_SAFETY - 1u; // out of range: a strategy index no slice holds
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t id = base + u * REDUCE_BLOCK;
            mine[u] = id < last && !(d0[u] & REC_SAFETY) && ((d0[u] & 0xffffffu) - s_lo) < s_n;
            if (mine[u]) {
                const uint4 *r = reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
                q0[u] = r[0];
                q1[u] = r[1];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t id = base + u * REDUCE_BLOCK;
            if (id < last && (d0[u] & REC_SAFETY)) {
                const uint32_t sh = id / gps, g = id - sh * gps;
                for (uint32_t s = 0; s < k; ++s) {
                    const uint32_t idx = perm_at(perm_T, S, perm_slots, sh, g * k + s) - s_lo;
                    if (idx < s_n) atomicAdd(&rt[idx * RT_COLS + 1u], 1ull);
                }
            }
            if (!mine[u]) continue;
            const uint32_t idx = (d0[u] & 0xffffffu) - s_lo;
            const unsigned long long m[10] = {q0[u].y, q0[u].z & 0xffffu, q0[u].z >> 16, q0[u].w & 0xffffu, q0[u].w >> 16,
                                              q1[u].x & 0xffffu, q1[u].x >> 16, q1[u].y & 0xffffu, q1[u].y >> 16, q1[u].z};
            unsigned long long *t = rt + idx * RT_COLS;
            atomicAdd(&t[0], 1ull);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (m[j]) {
                    atomicAdd(&t[2 + j], m[j]);
                    atomicAdd(&t[12 + j], m[j] * m[j]);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long *out = tally + ((size_t)batch * S + s_lo) * FK_TALLY_COLS;
    for (uint32_t i = threadIdx.x; i < s_n * RT_COLS; i += REDUCE_BLOCK) {
        const unsigned long long v = rt[i];
        if (v == 0ull) continue;
        const uint32_t idx = i / RT_COLS, c = i - idx * RT_COLS;
        // wins -> 0, safety -> 3, sums -> 4..13, square sums -> 15..24 (column 14 / 25 = winner_hit_max_rounds, always 0)
        const uint32_t col = (c == 0u) ? 0u : (c == 1u) ? 3u : (c < 12u) ? (c + 2u) : (c + 3u);
        atomicAdd(&out[(size_t)idx * FK_TALLY_COLS + col], v);
    }
}

// Small batches (per-shuffle tallies of a few dozen games each): one thread per game, atomics straight into the tally.
__global__ void fk_tally_direct_kernel(const uint32_t *recs, uint32_t n_games, uint32_t gps, uint32_t k, uint32_t S,
                                       const uint16_t *perm_T, uint32_t perm_slots, uint32_t sh_offset, uint32_t spb,
                                       unsigned long long *tally) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_games) return;
    const uint32_t sh = id / gps, g = id - sh * gps;
    const uint32_t batch = (sh_offset + sh) / spb;
    const uint4 *r = reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
    const uint4 q0 = r[0];
    unsigned long long *base = tally + (size_t)batch * S * FK_TALLY_COLS;
    if (q0.x & REC_SAFETY) {
        for (uint32_t s = 0; s < k; ++s)
            atomicAdd(&base[(size_t)perm_at(perm_T, S, perm_slots, sh, g * k + s) * FK_TALLY_COLS + 3u], 1ull);
        return;
    }
    if ((q0.x & 0xffffffu) >= S) return;
    const uint4 q1 = r[1];
    const unsigned long long m[10] = {q0.y, q0.z & 0xffffu, q0.z >> 16, q0.w & 0xffffu, q0.w >> 16,
                                      q1.x & 0xffffu, q1.x >> 16, q1.y & 0xffffu, q1.y >> 16, q1.z};
    unsigned long long *t = base + (size_t)(q0.x & 0xffffffu) * FK_TALLY_COLS;
    atomicAdd(&t[0], 1ull);
    for (int j = 0; j < 10; ++j) {
        if (m[j]) {
            atomicAdd(&t[4 + j], m[j]);
            atomicAdd(&t[15 + j], m[j] * m[j]);
        }
    }
}

// Result records of a batched H2H launch -> per-block {completed, safety, wins_seat1, wins_seat2}.  A block's attempts
// are contiguous games: a wave walks H2H_RUN consecutive 64-game groups and keeps the counts of the block it is in in
// wave-uniform registers; they go out with one atomic per counter when the block changes or the run ends (one atomic per
// counter and 64 games made 10^7 waves queue on the 4 x 132 words of a config-5 pass: 160 ms per 6 x 10^8 games).
constexpr uint32_t H2H_RUN = 32;

__global__ void fk_h2h_reduce_kernel(const uint32_t *recs, uint32_t n_games, uint32_t n_blocks,
                                     unsigned long long *out /* [n_blocks][4] */) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t lane = lane_id();
    uint32_t cur = 0xffffffffu, n_comp = 0, n_safe = 0, n_w2 = 0; // wave-uniform: the block being counted
    auto flush = [&]() {
        if (cur != 0xffffffffu && lane == 0u) {
            unsigned long long *o = out + (size_t)cur * 4;
            if (n_comp) atomicAdd(&o[0], (unsigned long long)n_comp);
            if (n_safe) atomicAdd(&o[1], (unsigned long long)n_safe);
            if (n_comp - n_w2) atomicAdd(&o[2], (unsigned long long)(n_comp - n_w2));
            if (n_w2) atomicAdd(&o[3], (unsigned long long)n_w2);
        }
        n_comp = n_safe = n_w2 = 0;
    };
    for (uint32_t g = 0; g < H2H_RUN; ++g) {
        const uint64_t base = ((uint64_t)wave * H2H_RUN + g) * 64u;
        if (base >= n_games) break; // wave-uniform
        const uint32_t id = (uint32_t)base + lane;
        const uint32_t d0 = id < n_games ? recs[id] : 0u;
        const uint32_t blk = (d0 & 0xffffffu) >> 1;
        const bool valid = id < n_games && blk < n_blocks; // (a record of a launch that raised an error may be garbage)
        const bool safety = (d0 & REC_SAFETY) != 0u;
        const uint32_t which = safety ? 1u : (2u + ((d0 >> 24) & 1u)); // completed games also count in column 0 below
        const uint64_t live = __ballot(valid);
        if (!live) continue;
        const uint32_t first_live = (uint32_t)(__ffsll((long long)live) - 1);
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)blk, (int)__builtin_amdgcn_readfirstlane((int)first_live));
        if (__ballot(valid && blk != lead) == 0ull) {
            if (lead != cur) {
                flush();
                cur = lead;
            }
            const uint32_t s_n = (uint32_t)__popcll(__ballot(valid && safety));
            n_safe += s_n;
            n_comp += (uint32_t)__popcll(live) - s_n;
            n_w2 += (uint32_t)__popcll(__ballot(valid && which == 3u));
        } else if (valid) { // a group that straddles blocks
            unsigned long long *o = out + (size_t)blk * 4;
            if (!safety) atomicAdd(&o[0], 1ull);
            atomicAdd(&o[which], 1ull);
        }
    }
    flush();
}

// State store + result records -> rows (fk_row_hdr + k x fk_seat = 4 + 28k bytes, simulation.py:628-655), in game-id
// order; ranks by stable sort on score desc (engine.py:477-483).  One thread builds one row; with STAGE the rows of a block go
// through an LDS tile (blockDim x row dwords, fk_rows_lds_bytes) and leave as whole lines: consecutive threads store
// consecutive dwords of the block's contiguous row range (a thread writing its own 60-byte row touched 15 scattered lines
// per wave store instruction: 1.7 ms per 10^7 k=2 rows, now HBM-write bound).
__device__ inline void build_row(const uint32_t *state, const uint32_t *recs, const uint32_t *inv_sched, uint32_t id, uint32_t gps, uint32_t n_sh,
                                 uint32_t k, uint32_t perm_mode, uint32_t *row) {
    const uint32_t slot = inv_sched ? inv_sched[id] : walk_slot(id, gps, n_sh, perm_mode != 0u);
    const uint4 q0 = *reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
    const bool completed = !(q0.x & REC_SAFETY);
    // header: n_rounds u16 | status u8 | winner_seat i8
    row[0] = (q0.z & 0xffffu) | ((completed ? (uint32_t)FK_COMPLETED : (uint32_t)FK_SAFETY_LIMIT) << 16) |
             ((completed ? ((q0.x >> 24) & 0x7fu) : 0xffu) << 24);
    const uint32_t *g = state + (size_t)slot * k * STATE_DW;
    for (uint32_t s = 0; s < k; ++s) {
        const uint32_t *x = g + (size_t)s * STATE_DW;
        const int32_t sc = (int32_t)x[R_SCORE]; // units of 50: ranks compare as they are, the row stores points
        uint32_t rank = 0;
        if (completed) {
            rank = 1;
            for (uint32_t j = 0; j < k; ++j) {
                const int32_t o = (int32_t)g[(size_t)j * STATE_DW + R_SCORE];
                rank += (o > sc || (o == sc && j < s)) ? 1u : 0u;
            }
        }
        const uint32_t xa = x[R_CA], xb = x[R_CB], xe = x[R_CE];
        uint32_t *d = row + 1u + 7u * s;
        d[0] = (uint32_t)sc * 50u;
        d[1] = x[R_IDX];
        d[2] = (xa >> 16) | (xa << 16);                    // farkles, rolls
        d[3] = (xb >> 16) | (((xb & 0xffffu) * 50u) << 16); // n_turns, highest_turn
        d[4] = x[R_CC];                 // sf_uses, sf_dice
        d[5] = x[R_CD];                 // so_uses, so_dice
        d[6] = (xe & 0xffffu) | (rank << 16) | ((completed ? 0u : 1u) << 24); // hot_dice, rank, hit_max_rounds
    }
}

template <bool STAGE>
__global__ void fk_rows_kernel(const uint32_t *state, const uint32_t *recs, const uint32_t *inv_sched, uint32_t n_games, uint32_t gps,
                               uint32_t n_sh, uint32_t k, uint32_t perm_mode, uint8_t *rows) {
    extern __shared__ uint32_t row_tile[];
    const uint32_t row_dw = 1u + 7u * k; // (4 + 28 k) / 4
    const uint32_t first = blockIdx.x * blockDim.x, id = first + threadIdx.x;
    uint32_t *out = reinterpret_cast<uint32_t *>(rows);
    if (!STAGE) {
        if (id < n_games) build_row(state, recs, inv_sched, id, gps, n_sh, k, perm_mode, out + (size_t)id * row_dw);
        return;
    }
    if (id < n_games) build_row(state, recs, inv_sched, id, gps, n_sh, k, perm_mode, row_tile + threadIdx.x * row_dw);
    __syncthreads();
    const uint32_t n_rows = min(blockDim.x, n_games - first), n_dw = n_rows * row_dw;
    uint32_t *dst = out + (size_t)first * row_dw;
    for (uint32_t i = threadIdx.x; i < n_dw; i += blockDim.x) dst[i] = row_tile[i];
}

// State store + result records -> the COLUMN IMAGE of every shuffle: what a row shard's Parquet pages hold, value for value, in
// their physical type (csrc/fk_shard_writer.h frames them on host threads; the reference's per-shuffle row shard,
// run_tournament.py:530-558, schema utils/schema_helpers.py:23-90).  Per shuffle, `stride` bytes:
//     int32 planes [4 + 13 k][gps]:  winner_strategy (id), winning_score, victory_margin, n_rounds, then per seat
//                                    score, farkles, rolls, highest_turn, strategy (id), rank, loss_margin, smart_five_uses,
//                                    n_smart_five_dice, smart_one_uses, n_smart_one_dice, hot_dice, n_turns   (schema order)
//     u8 status[gps] (1 = safety limit: every nullable field of the row is null), u8 winner_seat[gps], u8 rank_order[gps][k]
// One thread per game; a wave's stores to a plane are 64 consecutive int32 (whole lines), its loads one contiguous 48 k-byte
// state group per lane as in build_row.  Ranks by stable sort on score desc (engine.py:477-483); the margin is the winner's
// score minus the best of the others (ties: 0), as sorted(scores)[-2] in simulation.py:628-655.
__global__ __launch_bounds__(256) void fk_row_columns_kernel(const uint32_t *state, const uint32_t *recs, const uint32_t *inv_sched,
                                                             uint32_t n_games, uint32_t gps, uint32_t n_sh, uint32_t k, uint32_t perm_mode,
                                                             const int32_t *ids, uint8_t *out, size_t stride) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_games) return;
    const uint32_t sh = id / gps, g = id - sh * gps;
    const uint32_t slot = inv_sched ? inv_sched[id] : walk_slot(id, gps, n_sh, perm_mode != 0u);
    const uint4 q0 = *reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
    const bool completed = !(q0.x & REC_SAFETY);
    const uint32_t w = completed ? ((q0.x >> 24) & 0x7fu) : 0u;
    const uint32_t *gs = state + (size_t)slot * k * STATE_DW;
    uint8_t *image = out + (size_t)sh * stride;
    int32_t *plane = reinterpret_cast<int32_t *>(image) + g;
    uint8_t *bytes = image + (size_t)(4u + 13u * k) * 4u * gps;
    const int32_t win50 = (int32_t)gs[(size_t)w * STATE_DW + R_SCORE];
    int32_t second50 = 0;
    bool any = false;
    for (uint32_t j = 0; j < k; ++j) {
        const int32_t o = (int32_t)gs[(size_t)j * STATE_DW + R_SCORE];
        if (j != w && (!any || o > second50)) {
            second50 = o;
            any = true;
        }
    }
    plane[0] = completed ? ids[gs[(size_t)w * STATE_DW + R_IDX]] : 0;
    plane[(size_t)gps] = completed ? win50 * 50 : 0;
    plane[(size_t)2 * gps] = completed ? (win50 - (any ? second50 : 0)) * 50 : 0;
    plane[(size_t)3 * gps] = (int32_t)(q0.z & 0xffffu);
    bytes[g] = completed ? 0u : 1u;
    bytes[gps + g] = (uint8_t)w;
    uint8_t *order = bytes + (size_t)2 * gps + (size_t)g * k;
    for (uint32_t s = 0; s < k; ++s) {
        const uint32_t *x = gs + (size_t)s * STATE_DW;
        const int32_t sc = (int32_t)x[R_SCORE];
        uint32_t rank = 0;
        if (completed) {
            rank = 1;
            for (uint32_t j = 0; j < k; ++j) {
                const int32_t o = (int32_t)gs[(size_t)j * STATE_DW + R_SCORE];
                rank += (o > sc || (o == sc && j < s)) ? 1u : 0u;
            }
            order[rank - 1u] = (uint8_t)s;
        } else {
            order[s] = 0;
        }
        const uint32_t xa = x[R_CA], xb = x[R_CB], xc = x[R_CC], xd = x[R_CD], xe = x[R_CE];
        int32_t *p = plane + (size_t)(4u + 13u * s) * gps;
        p[0] = sc * 50;
        p[(size_t)1 * gps] = (int32_t)(xa >> 16);             // farkles
        p[(size_t)2 * gps] = (int32_t)(xa & 0xffffu);         // rolls
        p[(size_t)3 * gps] = (int32_t)((xb & 0xffffu) * 50u); // highest_turn
        p[(size_t)4 * gps] = ids[x[R_IDX]];                   // strategy id
        p[(size_t)5 * gps] = (int32_t)rank;
        p[(size_t)6 * gps] = completed ? (win50 - sc) * 50 : 0; // loss_margin
        p[(size_t)7 * gps] = (int32_t)(xc & 0xffffu);         // smart_five_uses
        p[(size_t)8 * gps] = (int32_t)(xc >> 16);             // n_smart_five_dice
        p[(size_t)9 * gps] = (int32_t)(xd & 0xffffu);         // smart_one_uses
        p[(size_t)10 * gps] = (int32_t)(xd >> 16);            // n_smart_one_dice
        p[(size_t)11 * gps] = (int32_t)(xe & 0xffffu);        // hot_dice
        p[(size_t)12 * gps] = (int32_t)(xb >> 16);            // n_turns
    }
}

// Integer sufficient statistics of ALL seats per (batch, strategy) — what the reference's unconditional all-player
// metrics are sums of (analysis/all_player_metrics.py:257-340): exposures, completed / safety / wins, sums and square
// sums of final score, n_turns, turns - rounds, rank, loss margin and the eight behaviour counters.  Exact in int64; the
// two ratio statistics of that module (score / n_turns, score / n_rounds) are float64 sums in row order and stay on the host.
// A strategy is seated exactly once per shuffle, so the statistics are GATHERED: thread = (strategy, part of a batch's
// shuffles); for every shuffle it finds the strategy's seat through the inverse permutation, reads that exposure's 32-byte
// digest (written game-major by fk_seat_digest_kernel) and accumulates in registers — no atomics per exposure.
//   columns: 0 exposures 1 completed 2 safety 3 wins 4 score 5 score^2 6 turns 7 turns^2 8 [turns != rounds]
//            9 (turns - rounds) 10 (turns - rounds)^2, then (sum, sum of squares) of rank, loss_margin [completed games
//            only], rolls, farkles, highest_turn, hot_dice, smart_five_uses, n_smart_five_dice, smart_one_uses, n_smart_one_dice
__global__ void fk_invert_perm_kernel(const uint16_t *perm_T, uint32_t S, uint32_t slots, uint32_t n_sh, uint16_t *inv_T) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // walks the blocked layout: coalesced reads
    const uint32_t per_block = S * slots, blocks = (n_sh + slots - 1u) / slots;
    if (t >= per_block * blocks) return;
    const uint32_t b = t / per_block, r = t - b * per_block, e = r / slots, l = r - e * slots;
    if (b * slots + l >= n_sh) return;
    inv_T[((size_t)b * S + perm_T[t]) * slots + l] = (uint16_t)e;
}

// Phase 1 of the all-seat statistics, game-major: one lane per (game, seat) digests what the statistics need of that exposure
// into 32 bytes at digest[game id * k + seat] (coalesced stores; the state records of a game are one contiguous 48 k-byte
// read, its result record one 32-byte read shared by the k lanes):
//   q0 = score / 50,   n_rounds | completed << 16 | won << 17,   rolls | farkles << 16,   highest_turn / 50 | hot_dice << 16
//   q1 = sf_uses | sf_dice << 16,   so_uses | so_dice << 16,   (winning score - own score) / 50,   n_turns | rank << 16
// The gather of phase 2 then touches one 32-byte record per exposure instead of the result record plus k state records.
__global__ __launch_bounds__(256) void fk_seat_digest_kernel(const uint32_t *state, const uint32_t *recs, const uint32_t *inv_sched,
                                                             uint32_t n_games, uint32_t gps, uint32_t n_sh, uint32_t k, uint4 *digest) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_games * k) return;
    const uint32_t id = (uint32_t)(t / k), seat = (uint32_t)(t - (size_t)id * k);
    const uint32_t slot = inv_sched ? inv_sched[id] : walk_slot(id, gps, n_sh, true);
    const uint4 q0 = *reinterpret_cast<const uint4 *>(recs + (size_t)id * REC_DW);
    const bool completed = !(q0.x & REC_SAFETY);
    const uint32_t *gs = state + (size_t)slot * k * STATE_DW, *x = gs + (size_t)seat * STATE_DW;
    const int32_t score = (int32_t)x[R_SCORE]; // units of 50
    uint32_t rank = 0, margin = 0, won = 0;
    if (completed) {
        rank = 1;
        for (uint32_t j = 0; j < k; ++j) { // stable sort on score desc (engine.py:477-483)
            const int32_t o = (int32_t)gs[(size_t)j * STATE_DW + R_SCORE];
            rank += (o > score || (o == score && j < seat)) ? 1u : 0u;
        }
        margin = q0.y / 50u - (uint32_t)score; // winning score - own score
        won = (((q0.x >> 24) & 0x7fu) == seat) ? 1u : 0u;
    }
    const uint32_t xb = x[R_CB], xe = x[R_CE];
    uint4 *d = digest + t * 2;
    d[0] = make_uint4((uint32_t)score, (q0.z & 0xffffu) | ((completed ? 1u : 0u) << 16) | (won << 17), x[R_CA],
                      (xb & 0xffffu) | (xe << 16));
    d[1] = make_uint4(x[R_CC], x[R_CD], margin, (xb >> 16) | (rank << 16)); // rank <= k <= 65 535
}

// Phase 2: thread = (strategy, part of a batch's shuffles); one digest record per exposure, accumulated in registers.
__global__ __launch_bounds__(256) void fk_seat_stats_kernel(const uint4 *digest, const uint16_t *inv_T, uint32_t perm_slots, uint32_t S, uint32_t k,
                                                            uint32_t gps, uint32_t n_sh, uint32_t sh_offset, uint32_t spb,
                                                            uint32_t parts_per_batch, uint32_t first_batch, long long *stats) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b_local = blockIdx.y / parts_per_batch, part = blockIdx.y - b_local * parts_per_batch;
    const uint32_t batch = first_batch + b_local;
    const uint64_t g_lo = (uint64_t)batch * spb, g_hi = g_lo + spb;
    const uint32_t sh_lo = g_lo > sh_offset ? (uint32_t)(g_lo - sh_offset) : 0u;
    const uint32_t sh_hi = (uint32_t)min<uint64_t>(n_sh, g_hi > sh_offset ? g_hi - sh_offset : 0u);
    if (s >= S || sh_hi <= sh_lo) return;
    const uint32_t per_part = (sh_hi - sh_lo + parts_per_batch - 1u) / parts_per_batch;
    const uint32_t first = sh_lo + part * per_part, last = min(first + per_part, sh_hi);
    long long acc[FK_SEAT_STAT_COLS];
#pragma unroll
    for (int c = 0; c < FK_SEAT_STAT_COLS; ++c) acc[c] = 0;
    for (uint32_t sh = first; sh < last; ++sh) {
        const uint32_t p = perm_at(inv_T, S, perm_slots, sh, s); // position = game * k + seat of the strategy in this shuffle
        const uint4 *d = digest + ((size_t)sh * gps * k + p) * 2;
        const uint4 q0 = d[0], q1 = d[1];
        const bool completed = (q0.y >> 16) & 1u;
        const long long score = (long long)(int32_t)q0.x * 50, rounds = q0.y & 0xffffu;
        const long long turns = q1.w & 0xffffu, tmr = turns - rounds;
        acc[0] += 1;
        acc[completed ? 1 : 2] += 1;
        acc[4] += score;
        acc[5] += score * score;
        acc[6] += turns;
        acc[7] += turns * turns;
        acc[8] += tmr != 0 ? 1 : 0;
        acc[9] += tmr;
        acc[10] += tmr * tmr;
        if (completed) {
            const long long rank = q1.w >> 16, margin = (long long)q1.z * 50;
            acc[3] += (q0.y >> 17) & 1u;
            acc[11] += rank;
            acc[12] += rank * rank;
            acc[13] += margin;
            acc[14] += margin * margin;
        }
        const long long v[8] = {q0.z & 0xffffu, q0.z >> 16, (long long)(q0.w & 0xffffu) * 50, q0.w >> 16,
                                q1.x & 0xffffu, q1.x >> 16, q1.y & 0xffffu, q1.y >> 16};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[15 + 2 * j] += v[j];
            acc[16 + 2 * j] += v[j] * v[j];
        }
    }
    long long *out = stats + ((size_t)batch * S + s) * FK_SEAT_STAT_COLS;
#pragma unroll
    for (int c = 0; c < FK_SEAT_STAT_COLS; ++c)
        if (acc[c]) atomicAdd(reinterpret_cast<unsigned long long *>(&out[c]), (unsigned long long)acc[c]);
}

// The four float64 sums of the all-player table that are NOT sums of integers (analysis/all_player_metrics.py:308-321): per exposure
// exact_return = score / n_turns and proxy_return = score / n_rounds (0 where the denominator is 0), their sums and the sums of
// their squares.  The reference accumulates them with np.add.at (:174-177: "deliberately unbuffered. It preserves source-row
// addition order"), i.e. for every strategy ONE sequential float64 sum over its exposures in source-row order.  A strategy is
// seated once per shuffle, so that order is the shuffle order: thread = (strategy, batch) walks the batch's shuffles of this chunk
// in ascending order and continues the running sums the previous chunk left in `ratios` — the same additions in the same order,
// IEEE-754 division / multiplication / addition each rounded once (no contraction into fused multiply-adds): the same bits.
__global__ __launch_bounds__(256) void fk_seat_ratio_kernel(const uint4 *digest, const uint16_t *inv_T, uint32_t perm_slots, uint32_t S, uint32_t k,
                                                            uint32_t gps, uint32_t n_sh, uint32_t sh_offset, uint32_t spb, uint32_t first_batch,
                                                            double *ratios) {
#pragma clang fp contract(off)
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t batch = first_batch + blockIdx.y;
    const uint64_t g_lo = (uint64_t)batch * spb, g_hi = g_lo + spb;
    const uint32_t sh_lo = g_lo > sh_offset ? (uint32_t)(g_lo - sh_offset) : 0u;
    const uint32_t sh_hi = (uint32_t)min<uint64_t>(n_sh, g_hi > sh_offset ? g_hi - sh_offset : 0u);
    if (s >= S || sh_hi <= sh_lo) return;
    double *out = ratios + ((size_t)batch * S + s) * FK_SEAT_RATIO_COLS;
    double exact_sum = out[0], exact_sq = out[1], proxy_sum = out[2], proxy_sq = out[3];
    auto add = [&](const uint4 &q0, const uint4 &q1) __attribute__((always_inline)) {
        const double score = (double)((long long)(int32_t)q0.x * 50), rounds = (double)(q0.y & 0xffffu), turns = (double)(q1.w & 0xffffu);
        const double exact = turns != 0.0 ? score / turns : 0.0, proxy = rounds != 0.0 ? score / rounds : 0.0;
        const double exact2 = exact * exact, proxy2 = proxy * proxy;
        exact_sum = exact_sum + exact;
        exact_sq = exact_sq + exact2;
        proxy_sum = proxy_sum + proxy;
        proxy_sq = proxy_sq + proxy2;
    };
    // The additions are one chain per sum, but the two dependent loads in front of them (inverse permutation -> digest record) are
    // not on it: sixteen shuffles' loads are issued together, then their quotients are added in shuffle order (a one-batch call is a
    // single thread per strategy over the whole shuffle range: 212 -> 27 ms for 312 500 shuffles of the 64-strategy grid).
    constexpr uint32_t U = 16;
    uint32_t sh = sh_lo;
    for (; sh + U <= sh_hi; sh += U) {
        uint32_t p[U];
        uint4 q0[U], q1[U];
#pragma unroll
        for (uint32_t j = 0; j < U; ++j) p[j] = perm_at(inv_T, S, perm_slots, sh + j, s);
#pragma unroll
        for (uint32_t j = 0; j < U; ++j) {
            const uint4 *d = digest + ((size_t)(sh + j) * gps * k + p[j]) * 2;
            q0[j] = d[0], q1[j] = d[1];
        }
#pragma unroll
        for (uint32_t j = 0; j < U; ++j) add(q0[j], q1[j]);
    }
    for (; sh < sh_hi; ++sh) {
        const uint4 *d = digest + ((size_t)sh * gps * k + perm_at(inv_T, S, perm_slots, sh, s)) * 2;
        add(d[0], d[1]);
    }
    out[0] = exact_sum, out[1] = exact_sq, out[2] = proxy_sum, out[3] = proxy_sq;
}

// ticket -> game id schedule inverted (rows are produced in game-id order)
__global__ void fk_invert_sched_kernel(const uint32_t *sched, uint32_t n_games, uint32_t *inv) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_games) inv[sched[t]] = t;
}

// ---------------------------------------------------------------------------------------
// single-op probes (parity tests of the device functions above)
// ---------------------------------------------------------------------------------------
__device__ inline uint32_t pack_faces(const uint8_t *f, int32_t n) {
    uint32_t c = 0;
    for (int32_t i = 0; i < n; ++i) c += 1u << (4u * (uint32_t)(f[i] - 1));
    return c;
}

__global__ void fk_score_lut_kernel(uint32_t *lut) { // the score table of fk_device.h, built on the device once per context
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < SCORE_LUT_KEYS) lut[key] = score_lut_entry32(key);
}

__global__ void fk_discard_lut_kernel(uint8_t *lut) { // the discard table of fk_device.h
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < DISCARD_LUT_KEYS) lut[key] = discard_lut_entry(key);
}

__global__ void fk_dbg_score_kernel(int64_t n, const uint8_t *faces, const int32_t *len, const int32_t *pre,
                                    const uint2 *strat, const uint32_t *lut, const uint8_t *dlut, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Strat50 s{(int32_t)strat[i].x, strat[i].y}; // packed strategies carry ceil(score_threshold / 50)
    // the game kernel's path: 3-bit count key -> score table -> discard choice, in units of 50 (pre[i] is a multiple of 50)
    const Roll50 r = default_score_lut50(lut, dlut, nibbles_to_lut_key(pack_faces(faces + i * 6, len[i])), len[i], pre[i] / 50, s);
    out[i * 5 + 0] = r.score50 * 50;
    out[i * 5 + 1] = r.used;
    out[i * 5 + 2] = len[i] - r.used;
    out[i * 5 + 3] = r.d5;
    out[i * 5 + 4] = r.d1;
}

// SeedSequence fingerprints of whole coordinates: generate_state(1, uint32)[0] and generate_state(1, uint64)[0]
// (utils/random.py:190-232; the ns-100 shuffle_seed and ns-102 game_seed columns of the row contract, the ns-1 seed of
// simulate_many_games).  All 18 entropy words are absorbed, including seat_index and replicate_index of the record.
__global__ void fk_coordinate_seed_kernel(int64_t n, const fk_coord *coords, uint32_t *out32, uint64_t *out64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fk_coord c = coords[i];
    SeedPool p;
    ss_begin(p, 2u, c.purpose, (uint32_t)c.root_seed, (uint32_t)(c.root_seed >> 32));
    ss_absorb64(p, c.k);
    ss_absorb64(p, c.shuffle_index);
    ss_absorb64(p, c.pair_id);
    ss_absorb64(p, c.order);
    ss_absorb64(p, c.game_index);
    ss_absorb64(p, c.seat_index);
    ss_absorb64(p, c.replicate_index);
    uint32_t w[2];
    ss_generate<2>(p, w);
    if (out32) out32[i] = w[0];
    if (out64) out64[i] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
}

// ns-102 game fingerprints of a shuffle range (the game_seed column of the row contract, run_tournament.py:340-350):
// out[(shuffle - shuffle0) * gps + g] = generate_state(1, uint32)[0] of coordinate (purpose, root, k, shuffle, game_index = g)
__global__ void fk_game_seed_kernel(SeedPool prefix, uint64_t shuffle0, uint32_t n_sh, uint32_t gps, uint32_t *out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_sh * gps) return;
    const uint32_t sh = (uint32_t)(t / gps), g = (uint32_t)(t - (size_t)sh * gps);
    SeedPool p = prefix; // entropy words 0..5: version, namespace, root, k
    p.hc = HC_AFTER_6_WORDS;
    ss_absorb64(p, shuffle0 + sh);
    ss_absorb64(p, 0); // pair_id
    ss_absorb64(p, 0); // order
    ss_absorb64(p, g); // game_index
    ss_absorb64(p, 0); // seat_index
    ss_absorb64(p, 0); // replicate_index
    uint32_t w[1];
    ss_generate<1>(p, w);
    out[t] = w[0];
}

__global__ void fk_dbg_continue_kernel(int64_t n, const int32_t *args, const uint2 *strat, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *g = args + i * 6;
    // turn_score g[0] and player_score g[5] are multiples of 50; the score to beat g[4] may be any integer (the target)
    out[i] = should_continue50(Strat50{(int32_t)strat[i].x, strat[i].y}, g[0] / 50, g[1], g[2] != 0, g[3] != 0, floor_div50(g[4]), g[5] / 50) ? 1 : 0;
}

__global__ void fk_dbg_dice_kernel(int64_t n, const uint4 *seeds, const uint4 *incs, const uint64_t *state_in, int32_t n_calls,
                                   const int32_t *sizes, int32_t total, uint8_t *faces, uint64_t *raw64,
                                   uint64_t *state_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r;
    if (seeds) {
        const uint4 s = seeds[i], c = incs[i]; // state plane (state_dw = 4), increment plane (k = 1)
        r.lo = (uint64_t)s.x | ((uint64_t)s.y << 32);
        r.hi = (uint64_t)s.z | ((uint64_t)s.w << 32);
        r.inc_lo = (uint64_t)c.x | ((uint64_t)c.y << 32);
        r.inc_hi = (uint64_t)c.z | ((uint64_t)c.w << 32);
        r.buf = 0;
        r.has_buf = 0;
    } else {
        const uint64_t *s = state_in + i * 6;
        r.hi = s[0];
        r.lo = s[1];
        r.inc_hi = s[2];
        r.inc_lo = s[3];
        r.has_buf = (uint32_t)s[4];
        r.buf = (uint32_t)s[5];
    }
    if (raw64) {
        Rng t = r;
        for (int j = 0; j < 4; ++j) raw64[i * 4 + j] = pcg_next64(t);
    }
    uint8_t *f = faces + i * total;
    for (int32_t c = 0; c < n_calls; ++c) {
        uint32_t packed = 0;
        roll_counts(r, (uint32_t)sizes[c], &packed);
        for (int32_t j = 0; j < sizes[c]; ++j) *f++ = (uint8_t)((packed >> (4 * j)) & 0xfu);
    }
    if (state_out) {
        uint64_t *s = state_out + i * 6;
        s[0] = r.hi;
        s[1] = r.lo;
        s[2] = r.inc_hi;
        s[3] = r.inc_lo;
        s[4] = r.has_buf;
        s[5] = r.buf;
    }
}

// The game kernels' own dice instantiation: roll_counts<3> (one 6w product per die, 18-bit count key, rejection test
// `min low word < 4`) from explicit generator states — keys[i * n_calls + c] = key of call c, state_out as fk_dbg_dice_kernel.
__global__ void fk_dbg_dice_key_kernel(int64_t n, const uint64_t *state_in, int32_t n_calls, const int32_t *sizes, uint32_t *keys,
                                       uint64_t *state_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *s = state_in + i * 6;
    Rng r{s[0], s[1], s[2], s[3], (uint32_t)s[5], (uint32_t)s[4]};
    for (int32_t c = 0; c < n_calls; ++c) keys[i * n_calls + c] = roll_counts<3>(r, (uint32_t)sizes[c]);
    uint64_t *o = state_out + i * 6;
    o[0] = r.hi, o[1] = r.lo, o[2] = r.inc_hi, o[3] = r.inc_lo, o[4] = r.has_buf, o[5] = r.buf;
}

} // namespace
