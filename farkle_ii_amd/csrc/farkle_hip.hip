// farkle_hip.hip — gfx950 kernels + C-ABI (include/farkle_hip.h) of the Farkle simulation engine.
//
// Three kernels per chunk of work, all on the context's stream:
//   fk_perm_kernel  one lane per shuffle: SeedSequence(ns=101) -> PCG64DXSM -> Fisher-Yates
//                   (Generator.permutation, run_tournament.py:312-318), written shuffle-minor.
//   fk_seed_kernel  one lane per game: coordinate -> SeedSequence -> PCG64DXSM (state, inc) of every
//                   seat (random.py:80-188); fully converged, 32 B per seat to HBM.
//   fk_play_kernel  persistent lanes, one lane = one game at a time, one roll per loop trip; seat
//                   contexts in LDS (field-major, conflict free), finished lanes are handed new games
//                   in wave-level batches; per-strategy tallies privatised in LDS when they fit.
//
// No MFMA (integer/table work), no CPU fallback.
#include "../../include/farkle_hip.h"
#include "fk_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace fk;

// ========================================================================================
// device side
// ========================================================================================
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
    const uint32_t *slow_bits; // bitmap over strategies: 1 = never banks voluntarily
    uint32_t n_sh;
    uint32_t *sched;         // [n_games] ticket -> game id, in dealing order (see the kernel)
    const uint32_t *class_ctr; // [2] games whose seats all / partly never bank (fk_class_count_kernel)
    uint32_t *sched_ctr;     // [3] per-class cursors
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

// Sizes of the first two schedule classes of fk_seed_kernel (games whose seats all / partly never bank): one lane per
// game in the seed kernel's walk order (coalesced permutation reads), one atomic per block and class.
__global__ __launch_bounds__(SEED_BLOCK) void fk_class_count_kernel(const uint16_t *perm_T, uint32_t perm_slots, uint32_t S,
                                                                    uint32_t k, uint32_t n_sh, uint32_t n_games,
                                                                    const uint32_t *slow_bits, uint32_t *class_ctr) {
    __shared__ uint32_t cnt[2];
    if (threadIdx.x < 2u) cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t n_all = 0, n_some = 0; // grid-stride: few blocks, so few same-address global atomics at the end
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_games; t += gridDim.x * blockDim.x) {
        const uint32_t g_local = t / n_sh, sh_local = t - g_local * n_sh;
        uint32_t n_slow = 0;
        for (uint32_t s = 0; s < k; ++s) {
            const uint32_t idx = perm_at(perm_T, S, perm_slots, sh_local, g_local * k + s);
            n_slow += (slow_bits[idx >> 5] >> (idx & 31u)) & 1u;
        }
        n_all += (n_slow == k) ? 1u : 0u;
        n_some += (n_slow != 0u && n_slow != k) ? 1u : 0u;
    }
    if (n_all) atomicAdd(&cnt[0], n_all);
    if (n_some) atomicAdd(&cnt[1], n_some);
    __syncthreads();
    if (threadIdx.x < 2u && cnt[threadIdx.x]) atomicAdd(&class_ctr[threadIdx.x], cnt[threadIdx.x]);
}

__global__ __launch_bounds__(SEED_BLOCK) void fk_seed_kernel(SeedArgs a) {
    __shared__ uint32_t wave_cnt[3][SEED_BLOCK / 64];
    __shared__ uint32_t block_base[3];
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
    // Longest-first schedule (tournament mode; scheduling only: results do not depend on the order games are dealt
    // in).  Three classes by the number of seats that never bank voluntarily: all of them (the game runs to the round
    // limit, ~13x the mean length), some (the banking seats decide the game but every turn of a never-banking seat runs
    // to its farkle: 1.6x the mean, tail to 9x), none.  Games are dealt in that order so that the launch drains on the
    // shortest class.  fk_class_count_kernel has counted the first two classes, so a game's ticket is class offset + its rank
    // in the class; ranks come from one returning atomic per block and class (a single word sustains only ~90
    // returning atomics/us).  The seeds are stored at the TICKET position: a wave's 64 consecutive tickets then read
    // 64 consecutive seed records whatever the class mix (stored in walk order, a sparse class dragged a full 128-B line
    // per game through L2: 3.7 GB of HBM fetches per 10^7 games instead of 1.1).
    uint32_t slot = t;
    if (a.sched) {
        uint32_t n_slow = 0;
        if (valid) {
            for (uint32_t s = 0; s < a.k; ++s) {
                const uint32_t idx = perm_at(a.perm_T, a.S, a.perm_slots, sh_local, g_local * a.k + s);
                n_slow += (a.slow_bits[idx >> 5] >> (idx & 31u)) & 1u;
            }
        }
        const uint32_t cls = (n_slow == a.k) ? 0u : (n_slow != 0u ? 1u : 2u); // all / some / none
        uint64_t cls_m[3];
        for (uint32_t cidx = 0; cidx < 3u; ++cidx) cls_m[cidx] = __ballot(valid && cls == cidx);
        const uint32_t wave = threadIdx.x >> 6;
        if (lane_id() == 0u)
            for (uint32_t cidx = 0; cidx < 3u; ++cidx) wave_cnt[cidx][wave] = (uint32_t)__popcll(cls_m[cidx]);
        __syncthreads();
        if (threadIdx.x < 3u) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < SEED_BLOCK / 64; ++w) {
                const uint32_t c = wave_cnt[threadIdx.x][w];
                wave_cnt[threadIdx.x][w] = total; // exclusive prefix
                total += c;
            }
            const uint32_t offset = threadIdx.x == 0u ? 0u : threadIdx.x == 1u ? a.class_ctr[0] : a.class_ctr[0] + a.class_ctr[1];
            block_base[threadIdx.x] = offset + (total ? atomicAdd(&a.sched_ctr[threadIdx.x], total) : 0u);
        }
        __syncthreads();
        if (valid) {
            slot = block_base[cls] + wave_cnt[cls][wave] + mbcnt(cls == 0u ? cls_m[0] : cls == 1u ? cls_m[1] : cls_m[2]);
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

// ========================================================================================
// host side
// ========================================================================================
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct fk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    std::string err;
    fk_timing timing{};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // see TimerSlot
    struct PendingTimer {
        float *acc;
        hipEvent_t a, b;
    };
    std::vector<PendingTimer> pending; // kernel timers recorded on the stream, read after the chunk's one sync
    DevBuf strat, perm, seeds, tally, rows, misc, ov, seatlist, coords, order, slow, score_lut, discard_lut, classes, dbg[6];
    int32_t longest_first = 1;
    int32_t blocks_per_cu = 0; // 0 = as many as fit
    int32_t lean = -1;         // -1 auto, 0 full 17-dword seat records, 1 lean 11-dword records
    int32_t uniform_flags_opt = -1; // -1 auto (scalar-flag instance when the table allows it), 0 never
    uint32_t table_flags = 0;            // set by upload_strategies: flag bits shared by the whole table ...
    uint32_t table_mixed_flags = 0xff00u; // ... and the flag bits that differ between its strategies
    int32_t waves_per_cu = 16; // resident-wave target used to size the grid
    int64_t chunk_bytes = (int64_t)24 << 30;
    int32_t batch_threshold = 8;
    int32_t use_lds_tally = -1;
    int32_t block = 0;
};

namespace {

int fail(fk_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail((c), FK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int ensure(fk_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return FK_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes, 256);
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return FK_OK;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

uint2 pack_strategy(const fk_strategy &s) {
    uint32_t bits = (uint32_t)(uint8_t)(int8_t)s.dice_threshold;
    if (s.smart_five) bits |= SF_SMART_FIVE;
    if (s.smart_one) bits |= SF_SMART_ONE;
    if (s.consider_score) bits |= SF_CONSIDER_SCORE;
    if (s.consider_dice) bits |= SF_CONSIDER_DICE;
    if (s.require_both) bits |= SF_REQUIRE_BOTH;
    if (s.auto_hot_dice) bits |= SF_AUTO_HOT;
    if (s.run_up_score) bits |= SF_RUN_UP;
    if (s.favor_score) bits |= SF_FAVOR_SCORE;
    return make_uint2((uint32_t)s.score_threshold, bits);
}

int validate_strategies(fk_ctx *c, const fk_strategy *s, int32_t S) {
    for (int32_t i = 0; i < S; ++i) {
        if (s[i].dice_threshold < -128 || s[i].dice_threshold > 127)
            return fail(c, FK_ERR_ARG, "strategy %d: dice_threshold %d outside [-128, 127]", i, s[i].dice_threshold);
        if (s[i].smart_one && !s[i].smart_five) // strategies.py:198
            return fail(c, FK_ERR_ARG, "strategy %d: smart_one requires smart_five", i);
        if (s[i].require_both && !(s[i].consider_score && s[i].consider_dice)) // strategies.py:202
            return fail(c, FK_ERR_ARG, "strategy %d: require_both requires consider_score and consider_dice", i);
    }
    return FK_OK;
}

int upload_strategies(fk_ctx *c, const fk_strategy *s, int32_t S) {
    int rc = validate_strategies(c, s, S);
    if (rc) return rc;
    std::vector<uint2> packed((size_t)S);
    for (int32_t i = 0; i < S; ++i) packed[(size_t)i] = pack_strategy(s[i]);
    rc = ensure(c, c->strat, sizeof(uint2) * (size_t)S);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->strat.p, packed.data(), sizeof(uint2) * (size_t)S, hipMemcpyHostToDevice, c->stream));
    uint32_t f_and = 0xff00u, f_or = 0u;
    for (int32_t i = 0; i < S; ++i) {
        f_and &= packed[(size_t)i].y;
        f_or |= packed[(size_t)i].y & 0xff00u;
    }
    c->table_mixed_flags = f_or & ~f_and;
    c->table_flags = f_and;
    // Strategies that never bank voluntarily outside the final round: should_continue's threshold term is
    // always true iff dice are considered with dice_threshold < 1 (dice_left >= 1 always exceeds it) and the
    // score threshold cannot veto (require_both, or score not considered).  Used for scheduling only.
    std::vector<uint32_t> slow(((size_t)S + 31) / 32, 0u);
    for (int32_t i = 0; i < S; ++i)
        if (s[i].consider_dice && s[i].dice_threshold < 1 && (s[i].require_both || !s[i].consider_score))
            slow[(size_t)i >> 5] |= 1u << (i & 31);
    rc = ensure(c, c->slow, slow.size() * 4);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->slow.p, slow.data(), slow.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // host vectors go out of scope
    return FK_OK;
}

SeedPool seed_prefix(uint32_t purpose, uint64_t root_seed, uint64_t k) {
    SeedPool p;
    ss_begin(p, 2u /* RNG_SCHEME_VERSION */, purpose, (uint32_t)root_seed, (uint32_t)(root_seed >> 32));
    ss_absorb64(p, k);
    return p;
}

struct LaunchPlan {
    int block = 0, grid = 0;
    size_t lds = 0;
    bool lds_tally = false;
    bool lean = false; // 11-dword seat records (increment + strategy re-read from HBM/L2 each turn)
    int wpe = 4;       // waves per SIMD the chosen instance is compiled for
    uint32_t mixed_flags = 0xff00u; // flag bits that differ between strategies of the table (selects the kernel instance)
};

constexpr size_t LDS_LIMIT = 160 * 1024;

size_t play_lds_bytes(int32_t k, int block, bool lean, bool lds_tally, int32_t S) {
    const size_t per_lane = (size_t)(lean ? NF - 6 : NF) * 4 * (size_t)k;
    return per_lane * (size_t)block + (lds_tally ? (size_t)S * LT_COLS * 8 : 0);
}

// Pick block size / record layout for the most resident lanes per CU (ties: full records, larger blocks).
// Instances are compiled for 4 waves/SIMD (<= 128 VGPRs); the 768-thread LEAN instance for 6 (80 VGPRs): its 12 waves
// split evenly over the 4 SIMDs, so two blocks (24 waves) co-reside.
LaunchPlan plan_play(const fk_ctx *c, int32_t k, int32_t S, bool single_batch) {
    LaunchPlan best;
    const bool want_tally = single_batch && (c->use_lds_tally != 0);
    int best_lanes = -1;
    for (int lean = 0; lean <= 1; ++lean) {
        if (c->lean >= 0 && lean != c->lean) continue;
        if (lean && S > (1 << (32 - CE_IDX_SHIFT))) continue; // strategy index must fit cE[31:18]
        for (int block : {1024, 768, 512, 256, 128, 64}) {
            if (c->block != 0 && block != c->block) continue;
            if (block == 768 && !lean) continue;
            const int wpe = (block == 768) ? 6 : 4;
            bool tally = want_tally && play_lds_bytes(k, block, lean != 0, true, S) <= LDS_LIMIT;
            size_t lds = play_lds_bytes(k, block, lean != 0, tally, S);
            if (lds > LDS_LIMIT) continue;
            int per_cu = (int)(LDS_LIMIT / std::max<size_t>(lds, 1));
            per_cu = std::min(per_cu, std::max(1, wpe * 4 * 64 / block));
            if (c->blocks_per_cu > 0) per_cu = std::min(per_cu, c->blocks_per_cu);
            per_cu = std::max(per_cu, 1);
            int lanes = (per_cu * block) * 4 + (tally ? 2 : 0) + (lean ? 0 : 1); // tie-breaks: tally, then full records
            if (lanes > best_lanes) {
                best_lanes = lanes;
                best.block = block;
                best.lds = lds;
                best.lds_tally = tally;
                best.lean = lean != 0;
                best.wpe = wpe;
                best.grid = c->prop.multiProcessorCount * per_cu;
            }
        }
    }
    if (best_lanes < 0) { // k too large for one wave: reported by run_chunk
        best.block = 64;
        best.lean = S <= (1 << (32 - CE_IDX_SHIFT));
        best.lds_tally = false;
        best.lds = play_lds_bytes(k, 64, best.lean, false, S);
        best.grid = c->prop.multiProcessorCount;
    }
    return best;
}

template <int BLOCK, bool LEAN, int WPE, uint32_t MIXED>
hipError_t launch_play_u(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_play_kernel<BLOCK, LEAN, WPE, MIXED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(p.lds, 16));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fk_play_kernel<BLOCK, LEAN, WPE, MIXED>), dim3((unsigned)p.grid), dim3(BLOCK), p.lds, s, a);
    return hipGetLastError();
}

constexpr uint32_t MIXED_ALL = 0xff00u, MIXED_NONE = 0u, MIXED_RB_FAV = SF_REQUIRE_BOTH | SF_FAVOR_SCORE;

template <int BLOCK, bool LEAN, int WPE = 4>
hipError_t launch_play_t(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    // the narrowest instance whose MIXED set covers the flags that actually vary in this table
    if (p.mixed_flags == MIXED_NONE) return launch_play_u<BLOCK, LEAN, WPE, MIXED_NONE>(p, a, s);
    if ((p.mixed_flags & ~MIXED_RB_FAV) == 0u) return launch_play_u<BLOCK, LEAN, WPE, MIXED_RB_FAV>(p, a, s);
    return launch_play_u<BLOCK, LEAN, WPE, MIXED_ALL>(p, a, s);
}

hipError_t launch_play(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    if (p.lds > LDS_LIMIT) return hipErrorInvalidValue;
    if (p.lean) {
        switch (p.block) {
        case 1024: return launch_play_t<1024, true>(p, a, s);
        case 768: return launch_play_t<768, true, 6>(p, a, s);
        case 512: return launch_play_t<512, true>(p, a, s);
        case 256: return launch_play_t<256, true>(p, a, s);
        case 128: return launch_play_t<128, true>(p, a, s);
        default: return launch_play_t<64, true>(p, a, s);
        }
    }
    switch (p.block) {
    case 1024: return launch_play_t<1024, false>(p, a, s);
    case 512: return launch_play_t<512, false>(p, a, s);
    case 256: return launch_play_t<256, false>(p, a, s);
    case 128: return launch_play_t<128, false>(p, a, s);
    default: return launch_play_t<64, false>(p, a, s);
    }
}

// Event pairs: kernels of one chunk are enqueued back to back (no host wait between them); their timers are read
// after the stream sync that ends the chunk (collect_timers).
enum TimerSlot : int { SLOT_PERM = 0, SLOT_SEED = 2, SLOT_PLAY = 4, SLOT_CALL = 6 };

struct Timer {
    fk_ctx *c;
    float *acc;
    hipEvent_t a, b;
    Timer(fk_ctx *ctx, float *dst, int slot) : c(ctx), acc(dst), a(ctx->ev[slot]), b(ctx->ev[slot + 1]) {
        (void)hipEventRecord(a, c->stream);
    }
    void stop() {
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({acc, a, b});
    }
};

hipError_t collect_timers(fk_ctx *c) { // the stream has been synchronised
    hipError_t first = hipSuccess;
    for (const auto &t : c->pending) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, t.a, t.b);
        if (e == hipSuccess) *t.acc += ms;
        else if (first == hipSuccess) first = e;
    }
    c->pending.clear();
    return first;
}

int check_device_error(fk_ctx *c, const int32_t *d_err, int64_t game_base, const char *what) {
    int32_t h[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(h, d_err, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h[0] == FK_ERR_ROLL_LIMIT)
        return fail(c, FK_ERR_ROLL_LIMIT, "Turn exceeded 1000 rolls - aborting. (%s game %lld)", what,
                    (long long)(game_base + h[1]));
    if (h[0] == FK_ERR_COUNTER_OVERFLOW)
        return fail(c, FK_ERR_COUNTER_OVERFLOW, "per-seat u16 counter left its guarded range (%s game %lld)", what,
                    (long long)(game_base + h[1]));
    if (h[0] != 0) return fail(c, h[0], "device error %d (%s game %lld)", h[0], what, (long long)(game_base + h[1]));
    return FK_OK;
}

// Run seeds + games for `n_games` games whose seeds/strategy sources are already described by the args.
int run_chunk(fk_ctx *c, const SeedArgs &sa_in, PlayArgs pa, const LaunchPlan &plan, int64_t game_base, const char *what) {
    SeedArgs sa = sa_in;
    if (plan.lds > LDS_LIMIT)
        return fail(c, FK_ERR_ARG, "k=%u needs %zu bytes of LDS per wave; the seat contexts of at most 58 players fit a CU",
                    sa.k, plan.lds);
    int rc = ensure(c, c->seeds, (size_t)sa.n_games * sa.k * 32);
    if (rc) return rc;
    // misc: [0] ticket counter (hammered by atomics), [16] error record, [256] schedule class cursors of the seed kernel
    rc = ensure(c, c->misc, 512);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 512, c->stream));
    sa.seeds = static_cast<uint4 *>(c->seeds.p);
    if (sa.perm_T && c->longest_first) {
        rc = ensure(c, c->order, (size_t)sa.n_games * 4);
        if (rc) return rc;
        sa.sched = static_cast<uint32_t *>(c->order.p);
        sa.sched_ctr = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(c->misc.p) + 256);
        rc = ensure(c, c->classes, 64);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->classes.p, 0, 64, c->stream));
        sa.class_ctr = static_cast<const uint32_t *>(c->classes.p);
        sa.slow_bits = static_cast<const uint32_t *>(c->slow.p);
        pa.sched = sa.sched;
    } else {
        sa.sched = nullptr;
        pa.sched = nullptr;
    }
    {
        Timer t(c, &c->timing.seed_ms, SLOT_SEED);
        if (sa.sched) // class sizes first: a game's ticket is class offset + rank
            hipLaunchKernelGGL(fk_class_count_kernel, dim3(std::min<uint32_t>((sa.n_games + SEED_BLOCK - 1u) / SEED_BLOCK, 1024u)), dim3(SEED_BLOCK), 0,
                               c->stream, sa.perm_T, sa.perm_slots, sa.S, sa.k, sa.n_sh, sa.n_games, sa.slow_bits,
                               static_cast<uint32_t *>(c->classes.p));
        hipLaunchKernelGGL(fk_seed_kernel, dim3((sa.n_games + SEED_BLOCK - 1u) / SEED_BLOCK), dim3(SEED_BLOCK), 0, c->stream, sa);
        t.stop();
        HIPCHK(c, hipGetLastError());
    }
    pa.seeds = sa.seeds;
    pa.ticket = static_cast<uint32_t *>(c->misc.p);
    pa.err = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(c->misc.p) + 16);
    pa.batch_threshold = (uint32_t)std::max(1, std::min(64, c->batch_threshold));
    pa.use_lds_tally = plan.lds_tally ? 1u : 0u;
    {
        Timer t(c, &c->timing.play_ms, SLOT_PLAY);
        LaunchPlan lp = plan;
        lp.mixed_flags = (c->uniform_flags_opt != 0) ? c->table_mixed_flags : 0xff00u;
        pa.uflags = c->table_flags;
        hipError_t e = launch_play(lp, pa, c->stream);
        t.stop();
        HIPCHK(c, e);
    }
    c->timing.play_launches += 1;
    c->timing.play_block = plan.block;
    c->timing.play_grid = plan.grid;
    c->timing.play_lds_bytes = (int32_t)plan.lds;
    c->timing.games += sa.n_games;
    const int rc_dev = check_device_error(c, pa.err, game_base, what); // synchronises the stream
    HIPCHK(c, collect_timers(c));
    return rc_dev;
}

} // namespace

extern "C" {

void fk_destroy(fk_ctx *c);

int fk_init(int device_ordinal, fk_ctx **out) {
    if (!out) return FK_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FK_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return FK_ERR_ARG;
    fk_ctx *c = new fk_ctx();
    c->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fk_destroy(c); // releases whatever was created
        return FK_ERR_HIP;
    }
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            fk_destroy(c);
            return FK_ERR_HIP;
        }
    // score table (fk_device.h): 512 KiB, built by one small kernel, read by every roll of the game kernel
    if (ensure(c, c->score_lut, SCORE_LUT_KEYS * sizeof(uint16_t)) != FK_OK) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    if (ensure(c, c->discard_lut, DISCARD_LUT_KEYS) != FK_OK) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    hipLaunchKernelGGL(fk_score_lut_kernel, dim3(SCORE_LUT_KEYS / 256), dim3(256), 0, c->stream,
                       static_cast<uint16_t *>(c->score_lut.p));
    hipLaunchKernelGGL(fk_discard_lut_kernel, dim3(DISCARD_LUT_KEYS / 256), dim3(256), 0, c->stream,
                       static_cast<uint8_t *>(c->discard_lut.p));
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    *out = c;
    return FK_OK;
}

void fk_destroy(fk_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->strat, &c->perm, &c->seeds, &c->tally, &c->rows, &c->misc, &c->ov, &c->seatlist, &c->coords, &c->order, &c->slow, &c->score_lut, &c->discard_lut, &c->classes})
        release(*b);
    for (auto &b : c->dbg) release(b);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *fk_last_error(fk_ctx *c) { return c ? c->err.c_str() : "null context"; }

int fk_get_device_info(fk_ctx *c, fk_device_info *out) {
    if (!c || !out) return FK_ERR_ARG;
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", c->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", c->prop.gcnArchName);
    out->compute_units = c->prop.multiProcessorCount;
    out->clock_mhz = c->prop.clockRate / 1000;
    out->wavefront_size = c->prop.warpSize;
    out->lds_bytes_per_cu = (int32_t)c->prop.maxSharedMemoryPerMultiProcessor;
    out->hbm_bytes = c->prop.totalGlobalMem;
    return FK_OK;
}

int fk_get_timing(fk_ctx *c, fk_timing *out) {
    if (!c || !out) return FK_ERR_ARG;
    *out = c->timing;
    return FK_OK;
}

int fk_set_option(fk_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return FK_ERR_ARG;
    std::string n(name);
    if (n == "chunk_bytes") c->chunk_bytes = std::max<int64_t>(value, 1 << 20);
    else if (n == "batch_threshold") c->batch_threshold = (int32_t)value;
    else if (n == "use_lds_tally") c->use_lds_tally = (int32_t)value;
    else if (n == "longest_first") c->longest_first = (int32_t)value;
    else if (n == "blocks_per_cu") c->blocks_per_cu = (int32_t)value;
    else if (n == "lean") c->lean = (int32_t)value;
    else if (n == "uniform_flags") c->uniform_flags_opt = (int32_t)value;
    else if (n == "waves_per_cu") c->waves_per_cu = (int32_t)std::max<int64_t>(1, std::min<int64_t>(32, value));
    else if (n == "block") {
        if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512 && value != 768 && value != 1024)
            return fail(c, FK_ERR_ARG, "block must be 0, 64, 128, 256, 512, 768 (lean records only) or 1024");
        c->block = (int32_t)value;
    } else return fail(c, FK_ERR_ARG, "unknown option %s", name);
    return FK_OK;
}

int fk_tournament_run(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                      uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                      int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms) {
    if (!c) return FK_ERR_ARG;
    if (!strategies || !tally) return fail(c, FK_ERR_ARG, "strategies and tally are required");
    if (k < 1 || S < k || S % k != 0) return fail(c, FK_ERR_ARG, "n_players must divide %d", S); // run_tournament.py:274
    if (S > 65535) return fail(c, FK_ERR_ARG, "S=%d exceeds 65535 strategies", S);
    if (max_rounds < 0 || max_rounds > 65535) return fail(c, FK_ERR_ARG, "max_rounds must be in [0, 65535]");
    if (shuffle_end < shuffle_begin || shuffles_per_batch == 0) return fail(c, FK_ERR_ARG, "bad shuffle range / batch size");
    if (n_ov < 0 || (n_ov > 0 && !ov)) return fail(c, FK_ERR_ARG, "bad override list");
    HIPCHK(c, hipSetDevice(c->device));
    c->timing = fk_timing{};
    c->pending.clear();

    const uint64_t n_sh_total = shuffle_end - shuffle_begin;
    const uint32_t gps = (uint32_t)(S / k);
    const uint64_t n_batches = (n_sh_total + shuffles_per_batch - 1) / shuffles_per_batch;
    const size_t tally_bytes = sizeof(int64_t) * (size_t)n_batches * (size_t)S * FK_TALLY_COLS;
    if (n_sh_total == 0) return FK_OK;
    const size_t row_bytes = sizeof(fk_row_hdr) + sizeof(fk_seat) * (size_t)k;

    int rc = upload_strategies(c, strategies, S);
    if (rc) return rc;
    rc = ensure(c, c->tally, tally_bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->tally.p, 0, tally_bytes, c->stream));

    // chunk planning: whole shuffles per chunk inside the workspace budget
    const size_t bytes_per_shuffle = (size_t)S * 2 + (size_t)gps * k * 32 + (rows ? (size_t)gps * row_bytes : 0);
    uint64_t chunk_sh = std::max<uint64_t>(1, (uint64_t)c->chunk_bytes / bytes_per_shuffle);
    chunk_sh = std::min<uint64_t>(chunk_sh, (uint64_t)0x7fffffff / gps);
    chunk_sh = std::min<uint64_t>(chunk_sh, n_sh_total);

    const LaunchPlan plan = plan_play(c, k, S, n_batches == 1);
    const SeedPool perm_prefix = seed_prefix(101u /* SHUFFLE_PERMUTATION */, root_seed, (uint64_t)k);
    const SeedPool seat_prefix = seed_prefix(103u /* TOURNAMENT_PLAYER */, root_seed, (uint64_t)k);

    const hipEvent_t t0 = c->ev[SLOT_CALL], t1 = c->ev[SLOT_CALL + 1];
    HIPCHK(c, hipEventRecord(t0, c->stream));

    std::vector<uint16_t> perm_host;
    for (uint64_t done = 0; done < n_sh_total; done += chunk_sh) {
        const uint32_t n_sh = (uint32_t)std::min<uint64_t>(chunk_sh, n_sh_total - done);
        const uint64_t sh0 = shuffle_begin + done;
        const uint32_t n_games = n_sh * gps;

        const uint32_t slots = (uint32_t)std::max<size_t>(1, std::min<size_t>(PERM_BLOCK, LDS_LIMIT / ((size_t)S * 2)));
        const uint32_t perm_blocks = (n_sh + slots - 1u) / slots;
        rc = ensure(c, c->perm, (size_t)perm_blocks * S * slots * 2);
        if (rc) return rc;
        {
            Timer t(c, &c->timing.perm_ms, SLOT_PERM);
            const size_t perm_lds = (size_t)slots * S * 2;
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_perm_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)perm_lds));
            hipLaunchKernelGGL(fk_perm_kernel, dim3(perm_blocks), dim3(PERM_BLOCK), perm_lds, c->stream,
                               perm_prefix, sh0, n_sh, (uint32_t)S, slots, static_cast<uint16_t *>(c->perm.p));
            t.stop();
            HIPCHK(c, hipGetLastError());
        }
        if (perms) {
            perm_host.resize((size_t)perm_blocks * S * slots);
            HIPCHK(c, hipMemcpyAsync(perm_host.data(), c->perm.p, perm_host.size() * 2, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (uint32_t s = 0; s < n_sh; ++s)
                for (int32_t i = 0; i < S; ++i)
                    perms[(size_t)(done + s) * S + i] = perm_host[((size_t)(s / slots) * S + i) * slots + s % slots];
        }

        // overrides that fall into this chunk -> chunk-local game ids
        std::vector<DevOverride> dov;
        for (int32_t i = 0; i < n_ov; ++i) {
            if (ov[i].root_seed != root_seed || ov[i].k_or_order != (uint32_t)k) continue;
            if (ov[i].a < sh0 || ov[i].a >= sh0 + n_sh || ov[i].b >= gps) continue;
            if (ov[i].max_rounds > 65535u) return fail(c, FK_ERR_ARG, "override max_rounds must be <= 65535");
            dov.push_back(DevOverride{(uint32_t)((ov[i].a - sh0) * gps + ov[i].b), ov[i].max_rounds});
        }
        if (!dov.empty()) {
            rc = ensure(c, c->ov, dov.size() * sizeof(DevOverride));
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->ov.p, dov.data(), dov.size() * sizeof(DevOverride), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (rows) {
            rc = ensure(c, c->rows, (size_t)n_games * row_bytes);
            if (rc) return rc;
        }

        SeedArgs sa{};
        sa.prefix = seat_prefix;
        sa.coords = nullptr;
        sa.shuffle0 = sh0;
        sa.pair = 0;
        sa.order = 0;
        sa.game0 = 0;
        sa.gps = gps;
        sa.k = (uint32_t)k;
        sa.n_games = n_games;
        sa.perm_T = static_cast<const uint16_t *>(c->perm.p);
        sa.perm_slots = slots;
        sa.S = (uint32_t)S;
        sa.n_sh = n_sh;

        PlayArgs pa{};
        pa.strat = static_cast<const uint2 *>(c->strat.p);
        pa.score_lut = static_cast<const uint16_t *>(c->score_lut.p);
        pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
        pa.perm_T = static_cast<const uint16_t *>(c->perm.p);
        pa.perm_slots = slots;
        pa.seat_strategy = nullptr;
        pa.tally = static_cast<unsigned long long *>(c->tally.p);
        pa.rows = rows ? static_cast<uint8_t *>(c->rows.p) : nullptr;
        pa.ov = static_cast<const DevOverride *>(c->ov.p);
        pa.n_ov = (uint32_t)dov.size();
        pa.mode = MODE_PERM;
        pa.n_games = n_games;
        pa.gps = gps;
        pa.n_sh = n_sh;
        pa.k = (uint32_t)k;
        pa.S = (uint32_t)S;
        pa.sh_offset = (uint32_t)done;
        pa.spb = shuffles_per_batch;
        pa.target = target_score;
        pa.max_rounds = (uint32_t)max_rounds;

        rc = run_chunk(c, sa, pa, plan, (int64_t)done * gps, "tournament");
        if (rc) return rc;
        if (rows) {
            HIPCHK(c, hipMemcpyAsync(static_cast<uint8_t *>(rows) + (size_t)done * gps * row_bytes, c->rows.p,
                                     (size_t)n_games * row_bytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    const uint32_t n_rows = (uint32_t)(n_batches * (uint64_t)S);
    hipLaunchKernelGGL(fk_finalize_tally, dim3((n_rows + 255u) / 256u), dim3(256), 0, c->stream,
                       static_cast<unsigned long long *>(c->tally.p), n_rows, (uint32_t)S, shuffles_per_batch, n_sh_total, 1u);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(t1, c->stream));
    HIPCHK(c, hipMemcpyAsync(tally, c->tally.p, tally_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventElapsedTime(&c->timing.total_ms, t0, t1));
    return FK_OK;
}

int fk_play_games(fk_ctx *c, const fk_coord *coords, int64_t n_games, const fk_strategy *table, int32_t S,
                  const int32_t *seat_strategy, int32_t k, int32_t target_score, int32_t max_rounds, void *rows) {
    if (!c) return FK_ERR_ARG;
    if (!coords || !table || !seat_strategy || !rows) return fail(c, FK_ERR_ARG, "coords, table, seat_strategy, rows are required");
    if (k < 1 || S < 1 || n_games < 0 || n_games > 0x7fffffff / std::max(k, 1)) return fail(c, FK_ERR_ARG, "bad k / S / n_games");
    if (max_rounds < 0 || max_rounds > 65535) return fail(c, FK_ERR_ARG, "max_rounds must be in [0, 65535]");
    for (int64_t i = 0; i < n_games * k; ++i)
        if (seat_strategy[i] < 0 || seat_strategy[i] >= S) return fail(c, FK_ERR_ARG, "seat_strategy[%lld] out of range", (long long)i);
    for (int64_t i = 0; i < n_games; ++i) {
        if (coords[i].k != (uint64_t)k) // simulation.py:438
            return fail(c, FK_ERR_ARG, "Player RNG coordinate k does not match the number of seated strategies");
        if (coords[i].seat_index != 0) return fail(c, FK_ERR_ARG, "game coordinates carry seat_index 0 (seats are implied)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->timing = fk_timing{};
    c->pending.clear();
    if (n_games == 0) return FK_OK;
    int rc = upload_strategies(c, table, S);
    if (rc) return rc;
    const size_t row_bytes = sizeof(fk_row_hdr) + sizeof(fk_seat) * (size_t)k;
    rc = ensure(c, c->coords, sizeof(fk_coord) * (size_t)n_games);
    if (rc) return rc;
    rc = ensure(c, c->seatlist, sizeof(int32_t) * (size_t)n_games * k);
    if (rc) return rc;
    rc = ensure(c, c->rows, (size_t)n_games * row_bytes);
    if (rc) return rc;
    rc = ensure(c, c->tally, sizeof(int64_t) * (size_t)S * FK_TALLY_COLS);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->coords.p, coords, sizeof(fk_coord) * (size_t)n_games, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->seatlist.p, seat_strategy, sizeof(int32_t) * (size_t)n_games * k, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->tally.p, 0, sizeof(int64_t) * (size_t)S * FK_TALLY_COLS, c->stream));

    LaunchPlan plan = plan_play(c, k, S, false);

    SeedArgs sa{};
    sa.coords = static_cast<const fk_coord *>(c->coords.p);
    sa.k = (uint32_t)k;
    sa.n_games = (uint32_t)n_games;

    PlayArgs pa{};
    pa.strat = static_cast<const uint2 *>(c->strat.p);
    pa.score_lut = static_cast<const uint16_t *>(c->score_lut.p);
        pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
    pa.seat_strategy = static_cast<const int32_t *>(c->seatlist.p);
    pa.tally = static_cast<unsigned long long *>(c->tally.p);
    pa.rows = static_cast<uint8_t *>(c->rows.p);
    pa.mode = MODE_LIST;
    pa.n_games = (uint32_t)n_games;
    pa.gps = 1;
    pa.n_sh = 1;
    pa.k = (uint32_t)k;
    pa.S = (uint32_t)S;
    pa.spb = 1;
    pa.target = target_score;
    pa.max_rounds = (uint32_t)max_rounds;
    rc = run_chunk(c, sa, pa, plan, 0, "list");
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rows, c->rows.p, (size_t)n_games * row_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->timing.total_ms = c->timing.seed_ms + c->timing.play_ms;
    return FK_OK;
}

int fk_h2h_run(fk_ctx *c, const fk_strategy seats[2], uint64_t root_seed, uint64_t pair_id, uint32_t order, uint64_t target,
               uint64_t max_attempts, uint64_t chunk_games, int32_t target_score, int32_t max_rounds, const fk_override *ov,
               int32_t n_ov, uint64_t state[5]) {
    if (!c) return FK_ERR_ARG;
    if (!seats || !state) return fail(c, FK_ERR_ARG, "seats and state are required");
    if (order > 1u) return fail(c, FK_ERR_ARG, "order must be 0 or 1");
    if (max_rounds < 0 || max_rounds > 65535) return fail(c, FK_ERR_ARG, "max_rounds must be in [0, 65535]");
    if (n_ov < 0 || (n_ov > 0 && !ov)) return fail(c, FK_ERR_ARG, "bad override list");
    HIPCHK(c, hipSetDevice(c->device));
    c->timing = fk_timing{};
    c->pending.clear();
    uint64_t attempted = state[0], completed = state[1], safety = state[2], w1 = state[3], w2 = state[4];
    if (!(completed <= target) || !(attempted <= max_attempts) || completed + safety != attempted || w1 + w2 != completed)
        return fail(c, FK_ERR_ARG, "inconsistent block state"); // h2h_schedule.py:1366, 1422-1468
    uint64_t stop = attempted + chunk_games; // :1172
    if (stop > max_attempts || stop < attempted) stop = max_attempts;
    int rc = upload_strategies(c, seats, 2);
    if (rc) return rc;
    rc = ensure(c, c->tally, sizeof(int64_t) * 2 * FK_TALLY_COLS);
    if (rc) return rc;
    const SeedPool seat_prefix = seed_prefix(203u /* H2H_PLAYER */, root_seed, 2u);
    const LaunchPlan plan = plan_play(c, 2, 2, true);
    const uint64_t max_launch = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->chunk_bytes / 64, 1u << 30));

    // The stop rule is a prefix rule (first attempt index at which `target` games have completed).
    // Launching exactly (target - completed) attempts per pass can never overshoot it: the pass reaches
    // the target only if every attempt in it completes, i.e. at its last attempt.
    while (attempted < stop && completed < target) {
        const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(stop - attempted, target - completed), max_launch);
        std::vector<DevOverride> dov;
        for (int32_t i = 0; i < n_ov; ++i) {
            if (ov[i].root_seed != root_seed || ov[i].k_or_order != order || ov[i].a != pair_id) continue;
            if (ov[i].b < attempted || ov[i].b >= attempted + n) continue;
            if (ov[i].max_rounds > 65535u) return fail(c, FK_ERR_ARG, "override max_rounds must be <= 65535");
            dov.push_back(DevOverride{(uint32_t)(ov[i].b - attempted), ov[i].max_rounds});
        }
        if (!dov.empty()) {
            rc = ensure(c, c->ov, dov.size() * sizeof(DevOverride));
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->ov.p, dov.data(), dov.size() * sizeof(DevOverride), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        HIPCHK(c, hipMemsetAsync(c->tally.p, 0, sizeof(int64_t) * 2 * FK_TALLY_COLS, c->stream));
        SeedArgs sa{};
        sa.prefix = seat_prefix;
        sa.shuffle0 = 0;
        sa.pair = pair_id;
        sa.order = order;
        sa.game0 = attempted;
        sa.gps = 0;
        sa.k = 2;
        sa.n_games = (uint32_t)n;
        PlayArgs pa{};
        pa.strat = static_cast<const uint2 *>(c->strat.p);
        pa.score_lut = static_cast<const uint16_t *>(c->score_lut.p);
        pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
        pa.tally = static_cast<unsigned long long *>(c->tally.p);
        pa.ov = static_cast<const DevOverride *>(c->ov.p);
        pa.n_ov = (uint32_t)dov.size();
        pa.mode = MODE_FIXED;
        pa.n_games = (uint32_t)n;
        pa.gps = 1;
        pa.n_sh = 1;
        pa.k = 2;
        pa.S = 2;
        pa.spb = 1;
        pa.target = target_score;
        pa.max_rounds = (uint32_t)max_rounds;
        rc = run_chunk(c, sa, pa, plan, (int64_t)attempted, "h2h attempt");
        if (rc) return rc;
        int64_t t[2 * FK_TALLY_COLS];
        HIPCHK(c, hipMemcpyAsync(t, c->tally.p, sizeof(t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint64_t comp = (uint64_t)t[2], saf = (uint64_t)t[3]; // seat-1 strategy row: one exposure per game
        if (comp + saf != n) return fail(c, FK_ERR_HIP, "h2h tally conservation failed (%llu + %llu != %llu)",
                                         (unsigned long long)comp, (unsigned long long)saf, (unsigned long long)n);
        attempted += n;
        completed += comp;
        safety += saf;
        w1 += (uint64_t)t[0];
        w2 += (uint64_t)t[FK_TALLY_COLS + 0];
    }
    state[0] = attempted;
    state[1] = completed;
    state[2] = safety;
    state[3] = w1;
    state[4] = w2;
    c->timing.total_ms = c->timing.seed_ms + c->timing.play_ms;
    return FK_OK;
}

// ---- probes ----
int fk_debug_score(fk_ctx *c, int64_t n, const uint8_t *faces, const int32_t *len, const int32_t *pre,
                   const fk_strategy *strategy, int32_t *out) {
    if (!c || n < 0 || !faces || !len || !pre || !strategy || !out) return c ? fail(c, FK_ERR_ARG, "bad arguments") : FK_ERR_ARG;
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    for (int64_t i = 0; i < n; ++i) {
        if (len[i] < 0 || len[i] > 6) return fail(c, FK_ERR_ARG, "roll cannot contain more than six dice");
        for (int32_t j = 0; j < len[i]; ++j)
            if (faces[i * 6 + j] < 1 || faces[i * 6 + j] > 6) return fail(c, FK_ERR_ARG, "dice faces must be between 1 and 6");
    }
    int rc = validate_strategies(c, strategy, (int32_t)std::min<int64_t>(n, 0x7fffffff));
    if (rc) return rc;
    std::vector<uint2> packed((size_t)n);
    for (int64_t i = 0; i < n; ++i) packed[(size_t)i] = pack_strategy(strategy[i]);
    const size_t sz[5] = {(size_t)n * 6, (size_t)n * 4, (size_t)n * 4, (size_t)n * 8, (size_t)n * 20};
    for (int i = 0; i < 5; ++i)
        if ((rc = ensure(c, c->dbg[i], sz[i]))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->dbg[0].p, faces, sz[0], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dbg[1].p, len, sz[1], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dbg[2].p, pre, sz[2], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dbg[3].p, packed.data(), sz[3], hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(fk_dbg_score_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n,
                       static_cast<const uint8_t *>(c->dbg[0].p), static_cast<const int32_t *>(c->dbg[1].p),
                       static_cast<const int32_t *>(c->dbg[2].p), static_cast<const uint2 *>(c->dbg[3].p),
                       static_cast<const uint16_t *>(c->score_lut.p), static_cast<const uint8_t *>(c->discard_lut.p),
                       static_cast<int32_t *>(c->dbg[4].p));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->dbg[4].p, sz[4], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

int fk_debug_should_continue(fk_ctx *c, int64_t n, const int32_t *args, const fk_strategy *strategy, int32_t *out) {
    if (!c || n < 0 || !args || !strategy || !out) return c ? fail(c, FK_ERR_ARG, "bad arguments") : FK_ERR_ARG;
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = validate_strategies(c, strategy, (int32_t)std::min<int64_t>(n, 0x7fffffff));
    if (rc) return rc;
    std::vector<uint2> packed((size_t)n);
    for (int64_t i = 0; i < n; ++i) packed[(size_t)i] = pack_strategy(strategy[i]);
    const size_t sz[3] = {(size_t)n * 24, (size_t)n * 8, (size_t)n * 4};
    for (int i = 0; i < 3; ++i)
        if ((rc = ensure(c, c->dbg[i], sz[i]))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->dbg[0].p, args, sz[0], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dbg[1].p, packed.data(), sz[1], hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(fk_dbg_continue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n,
                       static_cast<const int32_t *>(c->dbg[0].p), static_cast<const uint2 *>(c->dbg[1].p),
                       static_cast<int32_t *>(c->dbg[2].p));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->dbg[2].p, sz[2], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

static int debug_dice_common(fk_ctx *c, int64_t n, const fk_coord *coords, const uint64_t *state, int32_t n_calls,
                             const int32_t *sizes, uint8_t *faces, uint64_t *raw64, uint64_t *state_out) {
    if (!c || n < 0 || n_calls < 0 || !sizes || !faces) return c ? fail(c, FK_ERR_ARG, "bad arguments") : FK_ERR_ARG;
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t total = 0;
    for (int32_t i = 0; i < n_calls; ++i) {
        if (sizes[i] < 1 || sizes[i] > 6) return fail(c, FK_ERR_ARG, "roll sizes must be in [1, 6]");
        total += sizes[i];
    }
    int rc;
    const uint4 *d_seeds = nullptr;
    const uint64_t *d_state = nullptr;
    if (coords) {
        if ((rc = ensure(c, c->coords, sizeof(fk_coord) * (size_t)n))) return rc;
        if ((rc = ensure(c, c->seeds, (size_t)n * 32))) return rc;
        // one 1-seat "game" per coordinate; the seed kernel offsets the seat stream by coords[i].seat_index
        HIPCHK(c, hipMemcpyAsync(c->coords.p, coords, sizeof(fk_coord) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        SeedArgs sa{};
        sa.coords = static_cast<const fk_coord *>(c->coords.p);
        sa.k = 1;
        sa.n_games = (uint32_t)n;
        sa.seeds = static_cast<uint4 *>(c->seeds.p);
        hipLaunchKernelGGL(fk_seed_kernel, dim3((unsigned)((n + SEED_BLOCK - 1) / SEED_BLOCK)), dim3(SEED_BLOCK), 0, c->stream, sa);
        HIPCHK(c, hipGetLastError());
        d_seeds = sa.seeds;
    } else {
        if (!state) return fail(c, FK_ERR_ARG, "state is required");
        if ((rc = ensure(c, c->dbg[0], (size_t)n * 48))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dbg[0].p, state, (size_t)n * 48, hipMemcpyHostToDevice, c->stream));
        d_state = static_cast<const uint64_t *>(c->dbg[0].p);
    }
    if ((rc = ensure(c, c->dbg[1], (size_t)n_calls * 4 + 4))) return rc;
    if ((rc = ensure(c, c->dbg[2], (size_t)n * std::max(total, 1)))) return rc;
    if ((rc = ensure(c, c->dbg[3], (size_t)n * 32))) return rc;
    if ((rc = ensure(c, c->dbg[4], (size_t)n * 48))) return rc;
    if (n_calls) HIPCHK(c, hipMemcpyAsync(c->dbg[1].p, sizes, (size_t)n_calls * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(fk_dbg_dice_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, n, d_seeds, d_state, n_calls,
                       static_cast<const int32_t *>(c->dbg[1].p), total, static_cast<uint8_t *>(c->dbg[2].p),
                       raw64 ? static_cast<uint64_t *>(c->dbg[3].p) : nullptr,
                       state_out ? static_cast<uint64_t *>(c->dbg[4].p) : nullptr);
    HIPCHK(c, hipGetLastError());
    if (total) HIPCHK(c, hipMemcpyAsync(faces, c->dbg[2].p, (size_t)n * total, hipMemcpyDeviceToHost, c->stream));
    if (raw64) HIPCHK(c, hipMemcpyAsync(raw64, c->dbg[3].p, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
    if (state_out) HIPCHK(c, hipMemcpyAsync(state_out, c->dbg[4].p, (size_t)n * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

int fk_coordinate_seeds(fk_ctx *c, int64_t n, const fk_coord *coords, uint32_t *seed32, uint64_t *seed64) {
    if (!c || n < 0 || !coords || (!seed32 && !seed64)) return c ? fail(c, FK_ERR_ARG, "bad arguments") : FK_ERR_ARG;
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure(c, c->coords, sizeof(fk_coord) * (size_t)n);
    if (rc) return rc;
    if ((rc = ensure(c, c->dbg[0], (size_t)n * 4))) return rc;
    if ((rc = ensure(c, c->dbg[1], (size_t)n * 8))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->coords.p, coords, sizeof(fk_coord) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(fk_coordinate_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n,
                       static_cast<const fk_coord *>(c->coords.p), seed32 ? static_cast<uint32_t *>(c->dbg[0].p) : nullptr,
                       seed64 ? static_cast<uint64_t *>(c->dbg[1].p) : nullptr);
    HIPCHK(c, hipGetLastError());
    if (seed32) HIPCHK(c, hipMemcpyAsync(seed32, c->dbg[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (seed64) HIPCHK(c, hipMemcpyAsync(seed64, c->dbg[1].p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

int fk_debug_dice(fk_ctx *c, int64_t n, const fk_coord *coords, int32_t n_calls, const int32_t *sizes, uint8_t *faces,
                  uint64_t *raw64) {
    if (!coords) return c ? fail(c, FK_ERR_ARG, "coords is required") : FK_ERR_ARG;
    return debug_dice_common(c, n, coords, nullptr, n_calls, sizes, faces, raw64, nullptr);
}

int fk_debug_dice_state(fk_ctx *c, int64_t n, const uint64_t *state, int32_t n_calls, const int32_t *sizes, uint8_t *faces,
                        uint64_t *state_out) {
    return debug_dice_common(c, n, nullptr, state, n_calls, sizes, faces, nullptr, state_out);
}

} // extern "C"
