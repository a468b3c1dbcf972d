// fk_kernels.h — the gfx950 kernels of the Farkle simulation engine (included by farkle_hip.hip, one translation unit).
//
//   fk_perm_kernel         one lane per shuffle: SeedSequence(ns=101) -> PCG64DXSM -> Fisher-Yates
//                          (Generator.permutation, run_tournament.py:312-318), arrays in LDS, written shuffle-minor.
//   fk_class_count_kernel  sizes of the schedule classes (how many seats of a game never bank).
//   fk_seed_kernel         one lane per game: coordinate -> SeedSequence -> PCG64DXSM (state, increment) of every seat
//                          (random.py:80-188), longest-first schedule, seed records stored in dealing order.
//   fk_play_kernel         persistent lanes, one lane = one game at a time, one roll per loop trip; seat records in
//                          LDS, finished lanes are handed new games in wave-level batches; per-strategy tallies
//                          privatised in LDS when they fit.
//   fk_finalize_tally, fk_score_lut_kernel, fk_discard_lut_kernel, fk_coordinate_seed_kernel, fk_dbg_* probes.
//
// The per-roll arithmetic (SeedSequence, PCG64DXSM, dice, scoring, discards, decisions) lives in fk_device.h.
#pragma once

#include "../../include/farkle_hip.h"
#include "fk_device.h"

#include <hip/hip_runtime.h>

using namespace fk;

namespace {

constexpr uint32_t HC_AFTER_6_WORDS = ss_hc(24); // 4 + 12 (all pairs) + 2*4 hashmix calls

enum : uint32_t { MODE_PERM = 0, MODE_LIST = 1, MODE_FIXED = 2 };

// LDS seat-record fields (dwords), record layout lds[(seat * BLOCK + tid) * NFIELDS + field]
enum : uint32_t {
    F_LO0 = 0, F_LO1, F_HI0, F_HI1, F_INC_LO0, F_INC_LO1, F_INC_HI0, F_INC_HI1,
    F_BUF, F_SCORE, F_CA, F_CB, F_CC, F_CD, F_CE, F_SPX, F_SPY, NF
};
// packed u16 counter pairs
//   cA = rolls | farkles << 16        cB = highest_turn | n_turns << 16
//   cC = sf_uses | sf_dice << 16      cD = so_uses | so_dice << 16
//   cE = hot_dice | flags << 16       flags: bit0 has_scored, bit1 has_buf
constexpr uint32_t CE_HAS_SCORED = 1u << 16, CE_HAS_BUF = 1u << 17;
constexpr uint32_t CE_IDX_SHIFT = 18; // LEAN records: strategy index in cE[31:18] (S <= 16384)

constexpr uint32_t LT_COLS = 24; // LDS tally columns: wins, completed, safety, 10 sums, 10 square sums, pad
constexpr uint32_t TICKET_CHUNK = 64;

struct DevOverride {
    uint32_t game; // chunk-local game id
    uint32_t max_rounds;
};

struct SeedArgs {
    SeedPool prefix;         // pool after entropy words 0..5 (version, namespace, root, k)
    const fk_coord *coords;  // LIST mode: explicit coordinates (full SeedSequence per game)
    uint64_t shuffle0, pair, order, game0;
    uint32_t gps;            // games per shuffle (affine id -> (shuffle, game)); 0 = no split
    uint32_t k;
    uint32_t n_games;
    uint4 *seeds;            // [2][n_games][k]: plane 0 = PCG state {lo, hi} (read once per game), plane 1 = increment
                             // {lo, hi} (re-read at every turn start by lean-record kernels: a compact plane keeps
                             // the increments of all resident games in L2)
    // longest-first scheduling (tournament mode): games whose seats ALL never bank run to the round
    // limit (~13x the mean length); they are dealt first so that they do not form the tail of a wave.
    const uint16_t *perm_T;  // nullable; blocked layout, see perm_at()
    uint32_t perm_slots, S;
    const uint8_t *patience;   // per strategy: 3 = never banks voluntarily ... 0 = banks readily (scheduling only)
    uint32_t n_sh;
    uint32_t *sched;         // [n_games] ticket -> game id, in dealing order (see the kernel)
    const uint32_t *class_ctr; // [SCHED_CLASSES] class sizes (fk_class_count_kernel)
    uint32_t *sched_ctr;     // [SCHED_CLASSES] per-class cursors
};

struct PlayArgs {
    const uint2 *strat;          // [S] packed strategies
    const uint16_t *score_lut;   // [SCORE_LUT_KEYS] score table (fk_device.h)
    const uint8_t *discard_lut;  // [DISCARD_LUT_KEYS] discard table (fk_device.h)
    const uint16_t *perm_T;      // blocked permutations (MODE_PERM), see perm_at()
    uint32_t perm_slots;
    const int32_t *seat_strategy; // [n_games][k] (MODE_LIST)
    const uint4 *seeds;
    const uint32_t *sched;       // nullable: ticket -> game id (longest-first schedule; seeds are stored by ticket)
    unsigned long long *tally;   // [n_batches][S][26]
    uint8_t *rows;               // nullable, [n_games] * (4 + 28k)
    uint32_t *ticket;
    int32_t *err;                // [0] code, [1] game id
    const DevOverride *ov;
    uint32_t n_ov;
    uint32_t mode;
    uint32_t n_games, gps, n_sh, k, S;
    uint32_t sh_offset, spb;     // batch = (sh_offset + sh_local) / spb
    int32_t target;
    uint32_t max_rounds;
    uint32_t batch_threshold;
    uint32_t use_lds_tally;
    uint32_t uflags;             // the flag bits (8..15) every strategy of the table shares, see MIXED below
};

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
constexpr int SEED_BLOCK = 1024;

// Longest-first schedule classes (scheduling only: results do not depend on the order games are dealt in).
// A strategy's patience is 3 if it never banks voluntarily (its games against other such seats run to the round
// limit, ~13x the mean length), 2 / 1 if it rolls on until one / two dice are left unless BOTH of its conditions say
// bank (long turns, many farkles: 1.6x / 1.2x the mean game length on the reference grid), else 0.  A game's class is 0
// when every seat has patience 3, otherwise 15 - min(sum of patience, 14): class 0 is dealt first, class 15 last, so the
// launch drains on the games that are shortest in expectation.
constexpr uint32_t SCHED_CLASSES = 16;

__device__ inline uint32_t schedule_class(const uint16_t *perm_T, uint32_t perm_slots, uint32_t S, uint32_t k,
                                          const uint8_t *patience, uint32_t sh_local, uint32_t g_local) {
    uint32_t sum = 0, n_never = 0;
    for (uint32_t s = 0; s < k; ++s) {
        const uint32_t p = patience[perm_at(perm_T, S, perm_slots, sh_local, g_local * k + s)];
        sum += p;
        n_never += (p == 3u) ? 1u : 0u;
    }
    return n_never == k ? 0u : (SCHED_CLASSES - 1u) - min(sum, SCHED_CLASSES - 2u);
}

// Class sizes: one lane per game in the seed kernel's walk order (coalesced permutation reads), grid-stride so that few
// blocks add to the same global words at the end.
__global__ __launch_bounds__(SEED_BLOCK) void fk_class_count_kernel(const uint16_t *perm_T, uint32_t perm_slots, uint32_t S,
                                                                    uint32_t k, uint32_t n_sh, uint32_t n_games,
                                                                    const uint8_t *patience, uint32_t *class_ctr) {
    __shared__ uint32_t cnt[SCHED_CLASSES];
    if (threadIdx.x < SCHED_CLASSES) cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_games; base += gridDim.x * blockDim.x) { // wave-uniform trip count
        const uint32_t t = base + threadIdx.x;
        uint32_t cls = SCHED_CLASSES;
        if (t < n_games) {
            const uint32_t g_local = t / n_sh, sh_local = t - g_local * n_sh;
            cls = schedule_class(perm_T, perm_slots, S, k, patience, sh_local, g_local);
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
    // word sustains only ~90 returning atomics/us).  The seeds are stored at the TICKET position: a wave's 64
    // consecutive tickets then read 64 consecutive seed records whatever the class mix (stored in walk order, a sparse
    // class dragged a full 128-B line per game through L2: 3.7 GB of HBM fetches per 10^7 games instead of 0.5).
    uint32_t slot = t;
    if (a.sched) {
        const uint32_t cls = valid ? schedule_class(a.perm_T, a.perm_slots, a.S, a.k, a.patience, sh_local, g_local) : SCHED_CLASSES;
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
    if (valid) {
        SeedPool gp;
        uint64_t seat0 = 0, replicate = 0;
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
        } else {
            gp = a.prefix;
            gp.hc = HC_AFTER_6_WORDS;
            ss_absorb64(gp, a.shuffle0 + sh_local);
            ss_absorb64(gp, a.pair);
            ss_absorb64(gp, a.order);
            ss_absorb64(gp, a.game0 + g_local);
        }
        for (uint32_t s = 0; s < a.k; ++s) {
            SeedPool sp = gp;
            ss_absorb64(sp, seat0 + s); // seat_index
            ss_absorb64(sp, replicate); // replicate_index
            uint32_t g8[8];
            ss_generate<8>(sp, g8);
            Rng r;
            pcg_seed(r, g8);
            uint4 *dst = a.seeds + ((size_t)slot * a.k + s); // ticket position (walk order without a schedule)
            dst[0] = make_uint4((uint32_t)r.lo, (uint32_t)(r.lo >> 32), (uint32_t)r.hi, (uint32_t)(r.hi >> 32));
            dst[(size_t)a.n_games * a.k] =
                make_uint4((uint32_t)r.inc_lo, (uint32_t)(r.inc_lo >> 32), (uint32_t)r.inc_hi, (uint32_t)(r.inc_hi >> 32));
        }
    }
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
// One seat's context.  Every seat has a contiguous LDS record that each roll step loads, updates and stores; nothing
// but the turn registers and the owner's read-only data (increment, strategy) is carried in VGPRs across rolls.
// LEAN records keep only what a turn mutates (PCG state, buffered half word, score, counters: 11 dwords = 44 bytes
// instead of 68): the read-only PCG increment and the packed strategy are re-read from the seed buffer's increment
// plane / the strategy table (L2-resident) at the start of each turn, the strategy index riding in the spare bits of
// cE.  Fewer LDS bytes per lane = more resident waves per SIMD (k=2: 4 -> 6, k=4: 2 -> 3.5).  `Seat` is the in-register
// form used while a fresh game is set up.
struct Seat {
    uint64_t lo, hi, inc_lo, inc_hi; // PCG64DXSM state / increment
    uint32_t buf;                    // buffered half word (has_buf is bit 17 of cE)
    int32_t score;
    uint32_t cA, cB, cC, cD, cE;     // packed u16 counters + flags
    Strat sp;
};

// MIXED: the strategy flag bits that may differ between strategies of the table.  The other flags are the same for
// the whole table (threshold grids fix most of them): they arrive as a kernel argument, so their tests run on the
// scalar unit and the constants they select become s_cselects.  Instances: all flags mixed (generic), none, and
// require_both | favor_score (the pair the reference's grid always enumerates).
template <int BLOCK, bool LEAN, int WPE, uint32_t MIXED>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void fk_play_kernel(PlayArgs a) {
    extern __shared__ uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t K = a.k;
    constexpr uint32_t NFIELDS = LEAN ? (uint32_t)NF - 6u : (uint32_t)NF; // 11 or 17 dwords per seat record
    unsigned long long *tl = reinterpret_cast<unsigned long long *>(lds + NFIELDS * K * BLOCK);

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
    // read-only data of the turn owner (PCG increment, packed strategy): the only per-seat values carried in registers
    // across roll iterations.  The mutable seat record (generator state, score, counters) is loaded from and stored to
    // LDS inside every roll step, so the hot loop carries no per-seat PHIs through its divergent turn hand-over.
    uint64_t own_inc_lo = 0, own_inc_hi = 0;
    int32_t own_thr = 0;
    uint32_t own_bits = 0;

    // Seat records are contiguous per (seat, lane): record base = (seat * BLOCK + tid) * NFIELDS, field = immediate
    // offset (one address VGPR per record, ds_read2/ds_write2 pairs).  The odd record stride (11 / 17 dwords) maps the
    // 32 lanes of an LDS lane group to 32 distinct banks whatever seat each lane is on (BLOCK % 32 == 0).
    // LEAN records have no increment / strategy slots: fields after the increment move up by four.
    // Address = loop-invariant lane base + seat * compile-time stride: one full-rate v_mad_u32_u24 per record instead
    // of the quarter-rate 32-bit multiplies the plain index expression costs.
    const uint32_t lane_base = tid * NFIELDS;
    constexpr uint32_t SEAT_STRIDE = (uint32_t)BLOCK * NFIELDS; // < 2^24
    auto L = [&](uint32_t field, uint32_t s) __attribute__((always_inline)) -> uint32_t & {
        const uint32_t f = (LEAN && field > F_INC_HI1) ? field - 4u : field;
        return lds[__umul24(s, SEAT_STRIDE) + lane_base + f];
    };

    auto strategy_index = [&](uint32_t id, uint32_t s) -> uint32_t {
        if (a.mode == MODE_PERM) {
            const uint32_t sh = id / a.gps, g = id - sh * a.gps;
            return perm_at(a.perm_T, a.S, a.perm_slots, sh, g * K + s);
        }
        if (a.mode == MODE_LIST) return (uint32_t)a.seat_strategy[(size_t)id * K + s];
        return s;
    };

    // per-seat views used by the end-of-game code (seat s may be the turn owner or not)
    auto seat_strategy = [&](uint32_t s) -> uint32_t { // strategy-table index of seat s of the lane's current game
        if (LEAN) return L(F_CE, s) >> CE_IDX_SHIFT;
        return strategy_index(game_id, s);
    };
    auto seat_score = [&](uint32_t s) -> int32_t { return (int32_t)L(F_SCORE, s); };
    auto seat_counter = [&](uint32_t s, uint32_t field) -> uint32_t { return L(field, s); }; // field in F_CA..F_CE

    uint32_t x_idx = 0; // strategy index of the seat last loaded by load_seat_from_global
    auto load_seat_from_global = [&](Seat &x, uint32_t id, uint32_t slot, uint32_t s) {
        x_idx = strategy_index(id, s);
        const uint2 pk = a.strat[x_idx];
        const uint4 *src = a.seeds + ((size_t)slot * K + s);
        const uint4 stv = src[0], inc = src[(size_t)a.n_games * K];
        x.lo = (uint64_t)stv.x | ((uint64_t)stv.y << 32);
        x.hi = (uint64_t)stv.z | ((uint64_t)stv.w << 32);
        x.inc_lo = (uint64_t)inc.x | ((uint64_t)inc.y << 32);
        x.inc_hi = (uint64_t)inc.z | ((uint64_t)inc.w << 32);
        x.buf = 0;
        x.score = 0;
        x.cA = x.cB = x.cC = x.cD = x.cE = 0;
        x.sp = Strat{(int32_t)pk.x, pk.y};
    };

    // turn owner := seat s (engine.py:236-240): n_turns += 1 in its record, fresh turn registers, read-only data
    auto begin_turn = [&](uint32_t s) __attribute__((always_inline)) {
        L(F_CB, s) += 0x10000u; // n_turns += 1 (engine.py:236)
        if (LEAN) { // read-only per-seat data comes from HBM/L2; the loads overlap the first dice of the turn
            const uint4 inc = a.seeds[(size_t)a.n_games * K + (size_t)seed_slot * K + s];
            const uint2 pk = a.strat[L(F_CE, s) >> CE_IDX_SHIFT];
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

    // ---- finished game -> tallies / row (run_tournament.py:375-391, simulation.py:628-655) ----
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
        uint32_t batch = 0;
        if (a.mode == MODE_PERM) batch = (a.sh_offset + game_id / a.gps) / a.spb;
        uint32_t widx = 0;
        // exposures: tournament mode counts only safety-limit exposures (completed is derived in fk_finalize_tally)
        const bool count_exposures = !completed || a.mode != MODE_PERM;
        if (count_exposures) {
            for (uint32_t s = 0; s < K; ++s) {
                const uint32_t idx = seat_strategy(s);
                if (a.use_lds_tally) atomicAdd(&tl[idx * LT_COLS + (completed ? 1u : 2u)], 1ull);
                else atomicAdd(&a.tally[((size_t)batch * a.S + idx) * FK_TALLY_COLS + (completed ? 2u : 3u)], 1ull);
            }
        }
        if (completed) widx = seat_strategy(w);
        if (completed) {
            const uint32_t wa = seat_counter(w, F_CA), wb = seat_counter(w, F_CB), wc = seat_counter(w, F_CC),
                           wd = seat_counter(w, F_CD), we = seat_counter(w, F_CE);
            const unsigned long long m[10] = {(unsigned long long)(uint32_t)best, rounds, wa >> 16, wa & 0xffffu,
                                              wb & 0xffffu, wc & 0xffffu, wc >> 16, wd & 0xffffu, wd >> 16, we & 0xffffu};
            if (a.use_lds_tally) {
                unsigned long long *t = tl + widx * LT_COLS;
                atomicAdd(&t[0], 1ull);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    if (m[j]) { // zero-valued metrics (e.g. smart-discard counters of non-smart winners) add nothing
                        atomicAdd(&t[3 + j], m[j]);
                        atomicAdd(&t[13 + j], m[j] * m[j]);
                    }
                }
            } else {
                unsigned long long *t = a.tally + ((size_t)batch * a.S + widx) * FK_TALLY_COLS;
                atomicAdd(&t[0], 1ull);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    if (m[j]) {
                        atomicAdd(&t[4 + j], m[j]);
                        atomicAdd(&t[15 + j], m[j] * m[j]);
                    }
                }
            }
        }
        if (a.rows) {
            const size_t row_bytes = sizeof(fk_row_hdr) + sizeof(fk_seat) * (size_t)K;
            uint8_t *row = a.rows + (size_t)game_id * row_bytes;
            fk_row_hdr hdr;
            hdr.n_rounds = (uint16_t)rounds;
            hdr.status = completed ? FK_COMPLETED : FK_SAFETY_LIMIT;
            hdr.winner_seat = completed ? (int8_t)w : (int8_t)-1;
            *reinterpret_cast<fk_row_hdr *>(row) = hdr;
            for (uint32_t s = 0; s < K; ++s) {
                const int32_t sc = seat_score(s);
                uint32_t rank = 0;
                if (completed) {
                    rank = 1;
                    for (uint32_t j = 0; j < K; ++j) {
                        const int32_t o = seat_score(j);
                        rank += (o > sc || (o == sc && j < s)) ? 1u : 0u;
                    }
                }
                const uint32_t xa = seat_counter(s, F_CA), xb = seat_counter(s, F_CB), xc = seat_counter(s, F_CC),
                               xd = seat_counter(s, F_CD), xe = seat_counter(s, F_CE);
                uint32_t *d = reinterpret_cast<uint32_t *>(row + sizeof(fk_row_hdr) + sizeof(fk_seat) * s);
                d[0] = (uint32_t)sc;
                d[1] = seat_strategy(s);
                d[2] = (xa >> 16) | (xa << 16);                    // farkles, rolls
                d[3] = (xb >> 16) | (xb << 16);                    // n_turns, highest_turn
                d[4] = xc;                                         // sf_uses, sf_dice
                d[5] = xd;                                         // so_uses, so_dice
                d[6] = (xe & 0xffffu) | (rank << 16) | ((completed ? 0u : 1u) << 24); // hot_dice, rank, hit_max_rounds
            }
        }
    };

    // ---- fresh game for this lane ----
    auto init_game = [&](uint32_t id, uint32_t ticket) {
        game_id = id;
        max_rounds = a.max_rounds;
        for (uint32_t i = 0; i < a.n_ov; ++i)
            if (a.ov[i].game == id) max_rounds = a.ov[i].max_rounds;
        // seed records sit at the ticket position when there is a schedule, else in the seed kernel's walk order
        // (shuffle-minor in tournament mode)
        uint32_t slot = a.sched ? ticket : id;
        if (!a.sched && a.mode == MODE_PERM) {
            const uint32_t sh = id / a.gps, g = id - sh * a.gps;
            slot = g * a.n_sh + sh;
        }
        seed_slot = slot;
        for (uint32_t s = 0; s < K; ++s) {
            Seat x;
            load_seat_from_global(x, id, slot, s);
            L(F_LO0, s) = (uint32_t)x.lo;
            L(F_LO1, s) = (uint32_t)(x.lo >> 32);
            L(F_HI0, s) = (uint32_t)x.hi;
            L(F_HI1, s) = (uint32_t)(x.hi >> 32);
            if (!LEAN) {
                L(F_INC_LO0, s) = (uint32_t)x.inc_lo;
                L(F_INC_LO1, s) = (uint32_t)(x.inc_lo >> 32);
                L(F_INC_HI0, s) = (uint32_t)x.inc_hi;
                L(F_INC_HI1, s) = (uint32_t)(x.inc_hi >> 32);
                L(F_SPX, s) = (uint32_t)x.sp.score_thr;
                L(F_SPY, s) = x.sp.bits;
            }
            L(F_BUF, s) = 0u;
            L(F_SCORE, s) = 0u;
            L(F_CA, s) = 0u;
            L(F_CB, s) = 0u;
            L(F_CC, s) = 0u;
            L(F_CD, s) = 0u;
            L(F_CE, s) = LEAN ? (x_idx << CE_IDX_SHIFT) : 0u;
        }
        seat = 0;
        trigger = 0;
        final_round = 0;
        safety = 0;
        score_to_beat = a.target; // engine.py:451
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
        const bool trig = !fr & (score >= a.target);           // first trigger starts the final round (engine.py:462-468)
        const bool normal = !fr & !trig;
        const uint32_t n1 = seat + 1u;
        const bool wrap = n1 == K;
        const bool last = normal & wrap & (rounds >= max_rounds); // `while rounds < max_rounds` ends (engine.py:453, 472)
        const uint32_t next_fr = n1 + ((n1 == trigger) ? 1u : 0u); // final round skips the trigger seat (engine.py:523-550)
        const uint32_t next = fr ? next_fr : trig ? ((seat == 0u) ? 1u : 0u) : wrap ? 0u : n1;
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

    // ---- one roll of the current turn (engine.py:241-273): record in, roll, score, decide, record out ----
    auto roll_step = [&]() __attribute__((always_inline)) {
        const bool roll_limit = rolls_this_turn >= 1000u; // ROLL_LIMIT, engine.py:36,242 (raised below, before any store)
        const uint32_t s = seat;
        uint32_t cA = L(F_CA, s), cB = L(F_CB, s), cC = L(F_CC, s), cD = L(F_CD, s), cE = L(F_CE, s);
        int32_t score = (int32_t)L(F_SCORE, s);
        Rng rng{(uint64_t)L(F_HI0, s) | ((uint64_t)L(F_HI1, s) << 32), (uint64_t)L(F_LO0, s) | ((uint64_t)L(F_LO1, s) << 32),
                own_inc_hi, own_inc_lo, L(F_BUF, s), (cE & CE_HAS_BUF) ? 1u : 0u};
        const uint32_t n = dice;
        const uint32_t key = roll_counts<3>(rng, n);
        rolls_this_turn += 1u;
        const Strat sp{own_thr, (own_bits & (0xffu | MIXED)) | (a.uflags & (0xff00u & ~MIXED))};
        const RollResult rr = default_score_lut(a.score_lut, a.discard_lut, key, (int32_t)n, turn_score, sp);
        const bool farkle = rr.score == 0;                              // engine.py:135-137, 247-249
        cA += 1u + (farkle ? 0x10000u : 0u);                            // n_rolls (engine.py:98), n_farkles
        cC += (rr.d5 > 0) ? (1u + ((uint32_t)rr.d5 << 16)) : 0u;        // engine.py:139-144
        cD += (rr.d1 > 0) ? (1u + ((uint32_t)rr.d1 << 16)) : 0u;
        dice = (rr.used == (int32_t)n) ? 6u : (n - (uint32_t)rr.used);  // engine.py:146
        turn_score = farkle ? 0 : (turn_score + rr.score);
        const bool hot = !farkle & sp.has(SF_AUTO_HOT) & (dice == 6u);  // _apply_hot_dice, engine.py:149-154, 253
        cE += hot ? 1u : 0u;
        const bool keep = should_continue(sp, turn_score, (int32_t)dice, (cE & CE_HAS_SCORED) != 0u, final_round != 0u,
                                          score_to_beat, score);
        const bool over = farkle | (!hot & !keep);
        // bank (engine.py:265-273), branch-free: a farkled turn has turn_score 0 and changes nothing
        const uint32_t ts = over ? (uint32_t)turn_score : 0u;
        cE |= (ts >= 500u) ? CE_HAS_SCORED : 0u;
        const uint32_t banked = (cE & CE_HAS_SCORED) ? ts : 0u;
        score += (int32_t)banked;
        cB = (banked > (cB & 0xffffu)) ? ((cB & 0xffff0000u) | banked) : cB;
        // one rare exit for all error conditions: the roll limit, then the u16 guard bands (a turn adds <= 1000 rolls
        // and <= 2000 discarded dice; highest_turn must fit 16 bits)
        const bool overflow = (turn_score > 0xffff) | ((cA & 0xffffu) > 64000u) | ((cC >> 16) > 63000u) | ((cD >> 16) > 63000u);
        if (roll_limit | overflow) {
            raise(roll_limit ? FK_ERR_ROLL_LIMIT : FK_ERR_COUNTER_OVERFLOW);
            return;
        }
        cE = (cE & ~CE_HAS_BUF) | (rng.has_buf ? CE_HAS_BUF : 0u);
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
        if (over) advance(score);
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
}

// ---------------------------------------------------------------------------------------
// single-op probes (parity tests of the device functions above)
// ---------------------------------------------------------------------------------------
__device__ inline uint32_t pack_faces(const uint8_t *f, int32_t n) {
    uint32_t c = 0;
    for (int32_t i = 0; i < n; ++i) c += 1u << (4u * (uint32_t)(f[i] - 1));
    return c;
}

__global__ void fk_score_lut_kernel(uint16_t *lut) { // the score table of fk_device.h, built on the device once per context
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < SCORE_LUT_KEYS) lut[key] = score_lut_entry(key);
}

__global__ void fk_discard_lut_kernel(uint8_t *lut) { // the discard table of fk_device.h
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < DISCARD_LUT_KEYS) lut[key] = discard_lut_entry(key);
}

__global__ void fk_dbg_score_kernel(int64_t n, const uint8_t *faces, const int32_t *len, const int32_t *pre,
                                    const uint2 *strat, const uint16_t *lut, const uint8_t *dlut, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Strat s = unpack_strat(strat[i]);
    // the game kernel's path: 3-bit count key -> score table -> discard choice
    const RollResult r = default_score_lut(lut, dlut, nibbles_to_lut_key(pack_faces(faces + i * 6, len[i])), len[i], pre[i], s);
    out[i * 5 + 0] = r.score;
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

__global__ void fk_dbg_continue_kernel(int64_t n, const int32_t *args, const uint2 *strat, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *g = args + i * 6;
    out[i] = should_continue(unpack_strat(strat[i]), g[0], g[1], g[2] != 0, g[3] != 0, g[4], g[5]) ? 1 : 0;
}

__global__ void fk_dbg_dice_kernel(int64_t n, const uint4 *seeds, const uint64_t *state_in, int32_t n_calls,
                                   const int32_t *sizes, int32_t total, uint8_t *faces, uint64_t *raw64,
                                   uint64_t *state_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r;
    if (seeds) {
        const uint4 s = seeds[i], c = seeds[n + i]; // state plane, increment plane (k = 1)
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

} // namespace
