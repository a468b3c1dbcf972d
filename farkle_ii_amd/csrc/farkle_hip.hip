// farkle_hip.hip — host side of the C-ABI (include/farkle_hip.h) of the Farkle simulation engine: context, device
// buffers, launch planning and the extern "C" entry points.  The kernels are in fk_kernels.h, the per-roll device
// functions in fk_device.h.  Per chunk of work, on the context's stream: memset -> fk_perm_kernel ->
// fk_class_count_kernel + fk_seed_kernel -> fk_play_kernel (-> fk_finalize_tally), one host sync at the end.
//
// No MFMA (integer/table work), no CPU fallback.
#include "fk_kernels.h" // device side: every kernel of the engine (pulls in farkle_hip.h, fk_device.h, hip_runtime.h)
#include "fk_shard_writer.h" // host side: column images -> row-shard Parquet files
#include "fk_perm_wave.h"    // device side: a shuffle's Fisher-Yates draws by a whole wave (small launches)
#include "fk_row_columns_seats.h" // device side: column images with one thread per (game, seat)

#include <dlfcn.h>
#include <rccl/rccl.h> // TYPES ONLY (ncclConfig_t, result codes): the library itself is bound with dlopen on first use
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <unordered_map>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace fk;

// ========================================================================================
// host side
// ========================================================================================
constexpr int FK_ROWS_EVENTS = 4; // completion events of async rows calls: a ring (a caller may have this many calls' images not yet awaited)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Everything a chunk's preparation (permutations, schedule, seat seeding) writes and its game kernel reads.  There are two
// sets: the next chunk (of this call, or — after fk_tournament_hint_next — of the next call) is prepared into the other set
// around the game kernel of the current one (option "pipeline", default 1).  Its permutations run in front of that game
// kernel on the main stream: they want most of a CU's LDS, which the resident persistent game kernel holds (on the side
// stream they only started when it ended, with the seeding queued behind them).  Its schedule and seat seeding go to a
// low-priority stream; their kernels allocate ~1 KB of LDS per block, so they are placed as the game kernel's blocks retire,
// i.e. they fill its drain tail.  Measured (rocprofv3 traces + tools/time_pipeline.py, round 2): 10.33 -> 10.16 ms per
// config-2 step, 193.4 -> 191.6 ms per config-3 step.  Running the seeding BESIDE the game kernel instead (an LDS-free
// seed kernel) is zero-sum — both are VALU-issue bound: game kernel 178 -> 184 ms while a 6 ms seeding ran beside it — and
// stalled one run by 120 ms per step behind the low-priority queue: not done.
struct ChunkDesc {
    uint64_t epoch = 0, root = 0, sh0 = 0; // strategy-table upload epoch, root seed, first shuffle
    uint32_t n_sh = 0, S = 0, k = 0, state_dw = 0, sched = 0, slots = 0;
    bool operator==(const ChunkDesc &o) const {
        return epoch == o.epoch && root == o.root && sh0 == o.sh0 && n_sh == o.n_sh && S == o.S && k == o.k && state_dw == o.state_dw &&
               sched == o.sched && slots == o.slots;
    }
};

struct ChunkSet {
    DevBuf perm, draws, state, inc, seat_idx, order, classes, misc, pools;
    DevBuf blocks, game_block, game_row; // batched H2H: the pass's block table and game -> block / table-row maps
    hipEvent_t ready = nullptr;                       // recorded behind the preparation kernels
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // permutation begin / end, seeding begin / end (timing)
    bool prepared = false;
    bool side = false; // prepared on the side stream: its event intervals include waiting behind a game kernel
    ChunkDesc desc;
    SeedArgs sa{}; // the prepared chunk's buffers as the kernels see them
};

struct fk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t prep_stream = nullptr; // low priority: preparation of the next chunk
    ChunkSet sets[2];
    int cur = 0;
    uint64_t table_epoch = 0;
    bool hint_valid = false;           // fk_tournament_hint_next
    uint64_t hint_begin = 0, hint_end = 0;
    int32_t hint_state = 0, pipeline = 1;
    hipDeviceProp_t prop{};
    std::string err;
    fk_timing timing{};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // see TimerSlot
    int32_t *err_host = nullptr;    // pinned: the error record of a call's last game kernel (read with the tally)
    hipEvent_t main_idle = nullptr; // recorded on the main stream in front of a game kernel: what a side-stream preparation waits for
    struct PendingTimer {
        float *acc;
        hipEvent_t a, b;
    };
    std::vector<PendingTimer> pending; // kernel timers recorded on the stream, read after the chunk's one sync
    DevBuf strat, recs, rec0, tally, rows, ov, seatlist, coords, inv, slow, score_lut, discard_lut, block_out, stats, ratios, digest,
        dbg[6];
    std::vector<uint2> strat_host;   // the packed table currently resident in `strat` (uploaded once per table)
    int32_t longest_first = 1;
    int32_t blocks_per_cu = 0; // 0 = as many as fit
    int32_t max_waves = 6;     // resident waves per SIMD the launch plan may count on
    int32_t lean = -1;         // -1 auto, 0 full 17-dword seat records, 1 lean 11-dword records
    int32_t gs = -1;           // -1 / 0: LDS records whenever k of them fit a wave's share of LDS, 1: state-store instances always
    int32_t uniform_flags_opt = -1; // -1 auto (scalar-flag instance when the table allows it), 0 never
    uint32_t table_flags = 0;            // set by upload_strategies: flag bits shared by the whole table ...
    uint32_t table_mixed_flags = 0xff00u; // ... and the flag bits that differ between its strategies
    int64_t chunk_bytes = (int64_t)48 << 30;
    // The workspace budget of one buffer set is min(chunk_bytes, workspace_percent % of the memory this context can have — what is free now
    // plus what its own per-chunk buffers already hold — divided by the two sets); a call that still meets hipErrorOutOfMemory (another
    // process took the memory in between) releases its workspace, halves the budget and is replayed (round-5 finding: four ranks sharing
    // one device each sized two 48-GiB sets and rank 2 died in hipMalloc).
    int32_t workspace_percent = 80; // option "workspace_percent" (tests set it above 100 to provoke the replay)
    int64_t chunk_limit = 0;        // the replay's halved budget (0: none)
    int64_t last_budget = 0;        // the budget the last call planned with (fk_timing has no room: read through option "last_budget")
    bool oom = false;               // ensure() met hipErrorOutOfMemory
    std::vector<void *> hog;        // fk_debug_hold_memory: device memory held on purpose (tests of the budget / the replay)
    int32_t oom_replays = 0;        // how many times the last call was replayed with a smaller workspace
    int32_t batch_threshold = 0;  // 0 = auto: 8 waiting lanes up to eight seats, 12 at nine / ten, 16 at eleven / twelve (swept per k)
    int32_t use_lds_tally = -1;
    int32_t block = 0;
    int32_t hc = -1;           // hot / cold game kernel (fk_play_hc.h): -1 auto, 0 never, 1 whenever the table allows it
    int32_t hc_waves = 5;      // resident waves per SIMD the plan counts on for it
    int32_t hc_block = 256;    // its block size: 256, 768 or 1024
    int32_t hc_tables = 1;     // 1: score / discard tables in LDS (LT instances)
    int32_t hc_inc_regs = 1;   // 1: the seats' PCG increments in registers (k >= 5, 256-thread blocks, LDS tables)
    int32_t hc_cl = -1;        // cold records in LDS beside the hot part (k = 3 .. 5): -1 auto (k = 4), 0 never, 1 always
    DevBuf lds_tables;         // their LDS image (fk_play_hc.h)
    DevBuf cold;
    int32_t clock_stamps = 0;  // option "clock_stamps": every game-kernel block stamps s_memtime / s_memrealtime at both ends
    DevBuf clk;                // [grid][4] stamps of the last game kernel
    int clk_grid = 0;
    DevBuf lag_v, lag_out, lag_lags, lag_edge, lag_tmp; // fk_tournament_run_lags: value matrix, sums, lag list, head / tail rows
    bool ran_hc = false;       // the current tournament call launched the hot / cold kernel
    int32_t perm_split = -1;   // -1 auto, 0 one-kernel Fisher-Yates, 1 draws + serial swap chains, 2 draws + chain-free kernel
    int32_t columns_by_seat = -1; // column images: -1 one thread per (game, seat) up to sixteen seats, per game beyond; 0 per game; 1 per (game, seat)
    int32_t perm_draw_wave = -1; // the draws of a shuffle by -1: a wave up to WAVE_DRAW_MAX_SH shuffles per chunk, a thread beyond; 0: a thread; 1: a wave
    void *comm = nullptr;      // RCCL communicator (fk_comm_init), one per context / GPU
    int comm_rank = 0, comm_world = 1;
    bool comm_async = false;   // (non-blocking communicators: every RCCL call settled by polling; not used — see fk_comm_init)
    bool comm_lost = false;    // a collective failed or timed out and the communicator was aborted: reductions fail until fk_comm_init
    bool comm_timeout_set = false;    // "comm_timeout_ms" was set explicitly: the environment default no longer applies
    int32_t comm_timeout_ms = 120000; // option "comm_timeout_ms" (FK_COMM_TIMEOUT_MS): deadline of communicator creation and of each
                                      // collective; 0 = the blocking calls of round 3 (no deadline)
    DevBuf comm_buf;
    // resident tally (option "resident_tally"): every tournament call adds its [n_batches][S][26] tally to this device
    // accumulator; fk_tally_resident_reduce sums it over the communicator on the device (the tally never leaves HBM before
    // the reduce; one D2H, on the root)
    // rows mode: the device row buffer exists twice and a copy stream moves chunk i's rows to the host while chunk i + 1 plays
    DevBuf rows_alt;
    // fk_tournament_run_columns: the rows buffer receives per-shuffle column images (fk_row_columns_kernel); the strategy ids they name
    const int32_t *columns_ids = nullptr;
    uint32_t *want_shuffle_seeds = nullptr, *want_game_seeds = nullptr; // fk_tournament_run_columns_seeds: host destinations of this call
    DevBuf ids;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
    // option "rows_async": a rows call returns when its last copy to the host is QUEUED, not done; fk_rows_wait(slot) waits for it.  The caller
    // (farkle run: two page-locked buffers) then starts the next launch group at once, whose game kernel runs beside that copy.
    int32_t rows_async = 0;
    hipEvent_t ev_call_copy[FK_ROWS_EVENTS] = {};
    uint32_t rows_calls = 0;      // rows calls made in async mode: call n records ev_call_copy[n % FK_ROWS_EVENTS]
    // The mailbox: page-locked host memory the device STORES small results into (tallies, error records, game seeds) while a rows DMA
    // is in flight — a copy-engine transfer would queue behind those 256 MB and the call would wait for them after all (measured: an
    // async call took exactly as long as a waiting one until its tally left this way).
    uint8_t *mail = nullptr;
    size_t mail_cap = 0, mail_used = 0;
    struct Letter { void *dst; size_t off, bytes; };
    std::vector<Letter> letters;
    int32_t last_rows_event = -1; // ... and this is the slot of the last one (option "rows_event")
    int64_t rows_chunk_games = 4000000; // rows mode plays in chunks of about this many games (overlap granularity)
    int32_t resident = 0;
    size_t last_tally_bytes = 0; // bytes of the last successful tournament call's tally in `tally`
    DevBuf acc;
    size_t acc_n = 0;
};

#define CSET(c) ((c)->sets[(c)->cur])

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
    const hipError_t e = hipMalloc(&b.p, want);
    if (e == hipErrorOutOfMemory) {
        c->oom = true;
        b.p = nullptr;
        (void)hipGetLastError(); // (the sticky error would fail the next, unrelated call)
        return fail(c, FK_ERR_HIP, "hipMalloc(%zu bytes) failed: out of memory", want);
    }
    HIPCHK(c, e);
    b.cap = want;
    return FK_OK;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// per-chunk buffers: what workspace_budget counts as this context's own and release_workspace gives back
static std::vector<DevBuf *> workspace_buffers(fk_ctx *c) {
    std::vector<DevBuf *> out;
    for (auto &cs : c->sets)
        for (DevBuf *b : {&cs.perm, &cs.draws, &cs.state, &cs.inc, &cs.seat_idx, &cs.order, &cs.classes, &cs.misc, &cs.pools, &cs.blocks,
                          &cs.game_block, &cs.game_row})
            out.push_back(b);
    // (only buffers that every call sizes and ensures per chunk: c->slow — the patience table upload_strategies fills once per TABLE — is not one)
    for (DevBuf *b : {&c->recs, &c->rec0, &c->rows, &c->rows_alt, &c->digest, &c->inv, &c->lag_v, &c->lag_tmp}) out.push_back(b);
    return out;
}

// bytes one buffer set of a call may plan for (see fk_ctx::workspace_percent)
static int64_t workspace_budget(fk_ctx *c) {
    int64_t budget = c->chunk_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        size_t held = 0;
        for (DevBuf *b : workspace_buffers(c)) held += b->cap;
        const double have = (double)free_b + (double)held;
        budget = std::min<int64_t>(budget, std::max<int64_t>((int64_t)(have * (double)c->workspace_percent / 100.0 / 2.0), (int64_t)32 << 20));
    } else {
        (void)hipGetLastError();
    }
    if (c->chunk_limit > 0) budget = std::min(budget, c->chunk_limit);
    c->last_budget = budget;
    return budget;
}

static void release_workspace(fk_ctx *c) {
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    for (DevBuf *b : workspace_buffers(c)) release(*b);
    for (auto &cs : c->sets) cs.prepared = false;
    c->hint_valid = false;
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
    return make_uint2((uint32_t)ceil_div50(s.score_threshold), bits); // the kernels compare turn scores in units of 50 (fk_device.h)
}

int validate_strategies(fk_ctx *c, const fk_strategy *s, int64_t S) {
    for (int64_t i = 0; i < S; ++i) {
        if (s[i].dice_threshold < -128 || s[i].dice_threshold > 127)
            return fail(c, FK_ERR_ARG, "strategy %lld: dice_threshold %d outside [-128, 127]", (long long)i, s[i].dice_threshold);
        if (s[i].smart_one && !s[i].smart_five) // strategies.py:198
            return fail(c, FK_ERR_ARG, "strategy %lld: smart_one requires smart_five", (long long)i);
        if (s[i].require_both && !(s[i].consider_score && s[i].consider_dice)) // strategies.py:202
            return fail(c, FK_ERR_ARG, "strategy %lld: require_both requires consider_score and consider_dice", (long long)i);
    }
    return FK_OK;
}

// The packed table (+ the per-strategy patience of the longest-first schedule) goes to the device once per TABLE: a call
// whose table equals the resident one uploads nothing.
int upload_strategies(fk_ctx *c, const fk_strategy *s, int64_t S) {
    int rc = validate_strategies(c, s, S);
    if (rc) return rc;
    std::vector<uint2> packed((size_t)S);
    for (int64_t i = 0; i < S; ++i) packed[(size_t)i] = pack_strategy(s[i]);
    if (packed.size() == c->strat_host.size() && memcmp(packed.data(), c->strat_host.data(), packed.size() * sizeof(uint2)) == 0)
        return FK_OK;
    c->strat_host.clear();
    c->table_epoch += 1; // anything prepared for the previous table is void
    for (auto &cs : c->sets) cs.prepared = false;
    if (c->prep_stream) HIPCHK(c, hipStreamSynchronize(c->prep_stream));
    rc = ensure(c, c->strat, sizeof(uint2) * (size_t)S);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->strat.p, packed.data(), sizeof(uint2) * (size_t)S, hipMemcpyHostToDevice, c->stream));
    uint32_t f_and = 0xff00u, f_or = 0u;
    for (int64_t i = 0; i < S; ++i) {
        f_and &= packed[(size_t)i].y;
        f_or |= packed[(size_t)i].y & 0xff00u;
    }
    c->table_mixed_flags = f_or & ~f_and;
    c->table_flags = f_and;
    // Patience of a strategy, for the longest-first schedule only (fk_kernels.h, schedule_class).  It never banks
    // voluntarily outside the final round (3) iff should_continue's threshold term is always true: dice are considered
    // with dice_threshold < 1 (dice_left >= 1 always exceeds it) and the score threshold cannot veto (require_both, or
    // score not considered).  With the same "either condition keeps rolling" rule and dice_threshold 1 / 2 it rolls
    // down to one / two dice before it may bank (2 / 1).
    std::vector<uint8_t> patience((size_t)S, 0);
    for (int64_t i = 0; i < S; ++i)
        if (s[i].consider_dice && (s[i].require_both || !s[i].consider_score))
            patience[(size_t)i] = s[i].dice_threshold < 1 ? 3 : s[i].dice_threshold == 1 ? 2 : s[i].dice_threshold == 2 ? 1 : 0;
    rc = ensure(c, c->slow, patience.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->slow.p, patience.data(), patience.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // host vectors go out of scope
    c->strat_host.swap(packed);
    return FK_OK;
}

SeedPool seed_prefix(uint32_t purpose, uint64_t root_seed, uint64_t k) {
    SeedPool p;
    ss_begin(p, 2u /* RNG_SCHEME_VERSION */, purpose, (uint32_t)root_seed, (uint32_t)(root_seed >> 32));
    ss_absorb64(p, k);
    return p;
}

struct LaunchPlan {
    int block = 0, grid = 0, cus = 1;
    mutable int launched_grid = 0; // the grid after the occupancy clamp of the launch
    size_t lds = 0;
    bool lds_tally = false;
    bool lean = false; // 10-dword seat records (increment + strategy re-read from HBM/L2 each turn, 16-bit score / 50)
    bool gs = false;   // state-store instance: one LDS record per lane, the others in HBM
    bool blk = false;  // batched-H2H instance (strategy index from the lane's block index)
    bool hc = false;   // hot / cold instance (fk_play_hc.h): 20 bytes of LDS per seat, cold records in an L2-resident plane
    bool hc_lt = false; // ... with the score / discard tables in LDS
    int hc_ki = 0;      // ... with every seat's PCG increment in registers (2: the four-wave instances of k = 5 .. 7)
    bool hc_cl = false; // ... with the cold records in LDS (32 bytes per seat and lane, no plane)
    int wpe = 4;       // waves per SIMD the chosen instance is compiled for
    uint32_t mixed_flags = 0xff00u; // flag bits that differ between strategies of the table (selects the kernel instance)
};

constexpr size_t LDS_LIMIT = 160 * 1024;

size_t play_lds_bytes(int32_t k, int block, bool lean, bool gs, bool lds_tally, int32_t S, bool blocks_mode = false) {
    const size_t per_lane = (size_t)(lean ? LEAN_DW : NF) * 4 * (size_t)(gs ? 1 : k) + (blocks_mode ? 4 : 0);
    return per_lane * (size_t)block + (lds_tally ? (size_t)S * LT_COLS * 8 : 0);
}

// Pick block size / record layout for the most resident lanes per CU (ties: full records, larger blocks).
// Instances are compiled for 4 waves/SIMD (<= 128 VGPRs); the 768-thread LEAN instances for 6 (80 VGPRs): their 12 waves
// split evenly over the 4 SIMDs, so two blocks (24 waves) co-reside.  State-store (GS) instances are chosen for k >= 3:
// their LDS use does not grow with k.
// Lean records carry the banked total / 50 in 16 bits: tables whose target is above 50 * LEAN_MAX_TARGET50 points play with
// full records; block == 0 in the result = no instance fits (such a target with batched H2H, or with more seats than
// LDS holds full records for).
LaunchPlan plan_play(const fk_ctx *c, int32_t k, int64_t S, bool single_batch, int32_t target_score, bool blocks_mode = false) {
    LaunchPlan best;
    const bool lean_ok = ceil_div50(target_score) <= LEAN_MAX_TARGET50;
    if (blocks_mode) {
        if (!lean_ok) return best;
        // batched H2H: the one instance built for it (768 threads, lean LDS records of both seats + block index)
        best.block = 768;
        best.lean = true;
        best.blk = true;
        best.wpe = 6;
        best.lds = play_lds_bytes(2, 768, true, false, false, 0, true);
        const int per_cu = c->blocks_per_cu > 0 ? std::min(2, c->blocks_per_cu) : 2;
        best.grid = c->prop.multiProcessorCount * per_cu;
        best.cus = c->prop.multiProcessorCount;
        return best;
    }
    const bool want_tally = single_batch && !blocks_mode && (c->use_lds_tally != 0) && S <= 4096;
    int best_lanes = -1;
    // state-store instances only on request: measured 2x slower than LDS records at k = 4 / 8 (the per-turn record
    // exchange is bound by L2 / Infinity-Cache request throughput); they remain the path for tables too wide for LDS
    const bool gs_wanted = c->gs == 1;
    for (int gs = 0; gs <= 1; ++gs) {
        if (gs != (gs_wanted ? 1 : 0) && best_lanes >= 0) continue; // the other layout only if the wanted one does not fit
        for (int lean = 0; lean <= 1; ++lean) {
            if (gs && !lean) continue;
            if (lean && !lean_ok) continue;
            if (!gs && c->lean >= 0 && lean != c->lean) continue;
            if (!gs && lean && !blocks_mode && S > (1 << (32 - CE_IDX_SHIFT))) continue; // strategy index must fit cE[31:18]
            for (int block : {1024, 768, 512, 256, 128, 64}) {
                if (gs && block != 768) continue; // (one record per lane: 768-thread blocks seat six waves per SIMD whatever k is)
                if (c->block != 0 && block != c->block && !gs) continue;
                if (block == 768 && !lean) continue;
                const int wpe = (block == 768) ? 6 : 4;
                bool tally = want_tally && play_lds_bytes(k, block, lean != 0, gs != 0, true, (int32_t)S) <= LDS_LIMIT / (gs ? 2 : 1);
                size_t lds = play_lds_bytes(k, block, lean != 0, gs != 0, tally, (int32_t)S, blocks_mode);
                if (lds > LDS_LIMIT) continue;
                int per_cu = (int)(LDS_LIMIT / std::max<size_t>(lds, 1));
                // waves per SIMD: the occupancy an instance is compiled for (WPE) is a floor, not a ceiling — every instance
                // allocates at most 80 VGPRs, so six waves fit (launch_play_u trims the grid to the occupancy HIP reports).
                // Measured at k = 2 / 5160 strategies: 4 waves 25.8 ms, 6 waves (80 VGPRs) 22.4 ms, 7 waves (72 VGPRs) 23.5 ms.
                (void)wpe;
                per_cu = std::min(per_cu, std::max(1, c->max_waves * 4 * 64 / block));
                if (c->blocks_per_cu > 0) per_cu = std::min(per_cu, c->blocks_per_cu);
                per_cu = std::max(per_cu, 1);
                int lanes = (per_cu * block) * 4 + (tally ? 2 : 0) + (lean ? 0 : 1); // tie-breaks: tally, then full records
                if (gs == (gs_wanted ? 1 : 0)) lanes += 1 << 24;                     // the wanted layout wins when it fits
                if (lanes > best_lanes) {
                    best_lanes = lanes;
                    best.block = block;
                    best.lds = lds;
                    best.lds_tally = tally;
                    best.lean = lean != 0;
                    best.gs = gs != 0;
                    best.wpe = wpe;
                    best.grid = c->prop.multiProcessorCount * per_cu;
                    best.cus = c->prop.multiProcessorCount;
                }
            }
        }
    }
    return best; // feasible whenever lean records are: a GS instance needs 40 bytes of LDS per lane whatever k is
}

// The hot / cold instance (fk_play_hc.h) for a tournament launch, when the table allows it and it seats more waves than
// the LDS-record plan: 20 bytes of LDS per seat and lane, 256-thread blocks (one wave per SIMD each).
bool plan_play_hc(const fk_ctx *c, int32_t k, int32_t target_score, const LaunchPlan &base, LaunchPlan &out) {
    if (c->hc == 0 || c->gs == 1 || base.lds_tally) return false;
    if (k < 3 || k > (int32_t)HC_MAX_K || ceil_div50(target_score) > HC_MAX_TARGET50) return false;
    const int max_waves = std::max(1, std::min(c->hc_waves, c->max_waves));
    if (k <= 5 && (c->hc_cl > 0 || (c->hc_cl < 0 && k == 4))) {
        // cold records in LDS: 32 k bytes per lane; six / five / four waves per SIMD at k = 3 / 4 / 5.  Auto at k = 4 only (+5 %
        // over ten-dword records; k = 3 and k = 5 measured +-0 against ten-dword records / the register instance).  k = 4 runs four
        // 320-thread blocks: five 256-thread blocks of 32 768 bytes do NOT fit the 160 KB once each is rounded up to the LDS
        // allocation granule (measured: the fifth block never became resident, tools/exp_occupancy.py).
        const int block_cl = (k == 4 && c->hc_cl != 2) ? 320 : 256; // (2: the 5 x 256 launch, for the record)
        const int waves = k == 3 ? 6 : k == 4 ? 5 : 4;
        const size_t lds_cl = (size_t)block_cl * 32 * (size_t)k;
        int per_cu = (int)std::min<size_t>(LDS_LIMIT / lds_cl, (size_t)(waves * 256 / block_cl));
        if (c->blocks_per_cu > 0) per_cu = std::min(per_cu, c->blocks_per_cu);
        per_cu = std::max(per_cu, 1);
        out = base;
        out.hc = true;
        out.hc_lt = false;
        out.hc_ki = 0;
        out.hc_cl = true;
        out.lean = true;
        out.gs = false;
        out.blk = false;
        out.block = block_cl;
        out.lds = lds_cl;
        out.wpe = per_cu * block_cl / 256;
        out.grid = c->prop.multiProcessorCount * per_cu;
        out.cus = c->prop.multiProcessorCount;
        return true;
    }
    const bool lt = c->hc_tables != 0;
    // register instances (increments of every seat in registers, tables in LDS).  k = 5 .. 7 run FOUR waves per SIMD — 128
    // registers hold the increments when the packed strategies are loaded per turn instead — in whatever block size lets
    // the hot planes and the table images fit: 4 x 256 threads at k = 5, 2 x 512 at k = 6, 1 x 1 024 at k = 7; k = 8 (hot
    // planes alone 160 KB at four waves) stays at 3 x 256, and so does everything when the option caps the waves.
    const bool ki = c->hc_inc_regs != 0 && lt && c->hc_block != 1024 && c->hc_block != 768;
    const bool four = ki && k >= 5 && k <= 7 && max_waves >= 4;
    // nine to twelve seats (round 5): ONE 768-thread block per CU = three waves per SIMD, 168 registers per lane; the hot planes (16 k bytes
    // per lane since the buffered half word moved to the cold slot: 147 456 bytes at twelve seats) fit beside the table image
    const bool wide = ki && k >= 9 && max_waves >= 3;
    // the shipped library holds the plan's own instances only: k = 5 .. 7 at four waves, k = 8 at three, k = 9 .. 12 as above.  A cap below
    // that (option max_waves) sends the call to the LDS-record kernel instead of to an instance that was not compiled.
    if (k >= 5 && k <= 7 && !four) return false;
    if (k == 8 && !(ki && max_waves >= 3)) return false;
    if (k >= 9 && !wide) return false;
    const int block = wide ? 768 : four ? (k == 5 ? 256 : k == 6 ? 512 : 1024) : ki ? 256 : (c->hc_block == 1024 || c->hc_block == 768) ? c->hc_block : 256;
    // hot part: the generator state, 16 bytes per seat and lane (the buffered half word rides in the cold-plane slot)
    const size_t hot_bytes = 16;
    const size_t lds = (size_t)block * hot_bytes * (size_t)k + (lt ? LT_BYTES : 0);
    if (lds > LDS_LIMIT) return false;
    int per_cu = (int)std::min<size_t>(LDS_LIMIT / lds, (size_t)std::max(1, 256 * max_waves / block));
    if (wide) per_cu = 1;
    else if (four) per_cu = std::min(per_cu, 1024 / block);
    else if (ki) per_cu = std::min(per_cu, k <= 4 ? 4 : 3); // 128 / 168 registers per lane
    if (c->blocks_per_cu > 0) per_cu = std::min(per_cu, c->blocks_per_cu);
    per_cu = std::max(per_cu, 1);
    const int base_lanes = base.block * std::max(1, base.grid / std::max(1, base.cus));
    // auto: k >= 5.  Measured on the 5 160-strategy grid against the LDS-record kernel in the same process (round 3,
    // tools/exp_hc2.py, tools/exp_hc3.py): k = 8 +27 %, k = 7 +25 %, k = 6 +23 %, k = 5 +10 % games/s; k = 4 +-0 (five waves,
    // spilling) or -2 % (four), k = 3 -7 %.
    if (c->hc < 0 && (k < 5 || per_cu * block <= base_lanes)) return false;
    out = base;
    out.hc = true;
    out.hc_lt = lt;
    out.hc_ki = ki ? (wide ? 3 : four ? 2 : 1) : 0;
    out.lean = true;
    out.gs = false;
    out.blk = false;
    out.block = block;
    out.lds = lds;
    out.wpe = (per_cu * block + 255) / 256;
    out.grid = c->prop.multiProcessorCount * per_cu;
    out.cus = c->prop.multiProcessorCount;
    return true;
}

// LDS image of the score / discard tables (fk_play_hc.h), from the same __host__ __device__ functions that build the
// global tables (fk_device.h: lt_build_image — also what tests/native/device_header_host_check.hip checks the LDS path on)
std::vector<uint8_t> build_lds_tables() {
    std::vector<uint8_t> img(LT_BYTES, 0);
    lt_build_image(img.data());
    return img;
}

template <int BLOCK, bool LEAN, int WPE, uint32_t MIXED, bool GS, bool BLK = false, int KC = 0>
hipError_t launch_play_u(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    static int configured_dev = -1; // the dynamic-LDS ceiling of an instance is raised once per device, not per launch
    static size_t occ_lds = ~(size_t)0;
    static int occ_blocks = 0;
    const void *fn = reinterpret_cast<const void *>(&fk_play_kernel<BLOCK, LEAN, WPE, MIXED, GS, BLK, KC>);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (configured_dev != dev) { // a context on another device of this process: the attribute and the occupancy are per device
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        if (e != hipSuccess) return e;
        configured_dev = dev;
        occ_lds = ~(size_t)0;
    }
    const size_t lds = std::max<size_t>(p.lds, 16);
    if (occ_lds != lds) { // resident blocks per CU as the runtime counts them (registers, LDS, wave slots)
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, lds);
        if (e != hipSuccess) return e;
        occ_blocks = std::max(nb, 1);
        occ_lds = lds;
    }
    // a persistent grid: blocks beyond the resident ones would only queue behind them
    const int grid = std::min(p.grid, occ_blocks * p.cus);
    p.launched_grid = grid;
    hipLaunchKernelGGL((fk_play_kernel<BLOCK, LEAN, WPE, MIXED, GS, BLK, KC>), dim3((unsigned)grid), dim3(BLOCK), lds, s, a);
    return hipGetLastError();
}

constexpr uint32_t MIXED_ALL = 0xff00u, MIXED_NONE = 0u, MIXED_RB_FAV = SF_REQUIRE_BOTH | SF_FAVOR_SCORE;

template <int BLOCK, bool LEAN, int WPE = 4, bool GS = false, bool BLK = false, int KC = 0>
hipError_t launch_play_t(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    // the narrowest instance whose MIXED set covers the flags that actually vary in this table
    if (p.mixed_flags == MIXED_NONE) return launch_play_u<BLOCK, LEAN, WPE, MIXED_NONE, GS, BLK, KC>(p, a, s);
    if ((p.mixed_flags & ~MIXED_RB_FAV) == 0u) return launch_play_u<BLOCK, LEAN, WPE, MIXED_RB_FAV, GS, BLK, KC>(p, a, s);
    return launch_play_u<BLOCK, LEAN, WPE, MIXED_ALL, GS, BLK, KC>(p, a, s);
}

template <int BLOCK, uint32_t MIXED, bool LT, int KI = 0, int WPE = 0, bool PKR = true, bool CL = false, int NS = 8>
hipError_t launch_play_hc_u(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    static int configured_dev = -1; // dynamic-LDS ceiling and occupancy are per device
    static size_t occ_lds = ~(size_t)0;
    static int occ_blocks = 0;
    const void *fn = reinterpret_cast<const void *>(&fk_play_hc_kernel<BLOCK, MIXED, LT, KI, WPE, PKR, CL, NS>);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (configured_dev != dev) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        if (e != hipSuccess) return e;
        configured_dev = dev;
        occ_lds = ~(size_t)0;
    }
    if (occ_lds != p.lds) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, p.lds);
        if (e != hipSuccess) return e;
        occ_blocks = std::max(nb, 1);
        occ_lds = p.lds;
    }
    const int grid = std::min(p.grid, occ_blocks * p.cus);
    p.launched_grid = grid;
    hipLaunchKernelGGL((fk_play_hc_kernel<BLOCK, MIXED, LT, KI, WPE, PKR, CL, NS>), dim3((unsigned)grid), dim3(BLOCK), p.lds, s, a);
    return hipGetLastError();
}

template <int BLOCK, bool LT, int KI = 0, int WPE = 0, bool PKR = true, bool CL = false, int NS = 8>
hipError_t launch_play_hc_t(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    if (p.mixed_flags == MIXED_NONE) return launch_play_hc_u<BLOCK, MIXED_NONE, LT, KI, WPE, PKR, CL, NS>(p, a, s);
    if ((p.mixed_flags & ~MIXED_RB_FAV) == 0u) return launch_play_hc_u<BLOCK, MIXED_RB_FAV, LT, KI, WPE, PKR, CL, NS>(p, a, s);
    return launch_play_hc_u<BLOCK, MIXED_ALL, LT, KI, WPE, PKR, CL, NS>(p, a, s);
}

// The instances the launch plan can reach (plan_play_hc): k = 4 cold records in LDS (four 320-thread blocks, five waves per
// SIMD), k = 5 .. 7 four waves per SIMD with the increments in registers, k = 8 three.  Every other variant that was built and
// measured (profiles/HISTORY.md, section 4.9 of the round-4 document: global tables, increments / strategies loaded or held, three-wave forms, other block sizes, cold
// records in LDS at k = 3 / 5, cold records in registers, increments in the plane) lost or tied; their code left the tree in round 6
// (the A/B logs stay under profiles/, the sources in the repository's history).
hipError_t launch_play_hc(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    if (p.hc_cl) { // cold records in LDS
        // (77 VGPRs: a SIMD must be able to take six waves, or the 2 + 1 + 1 + 1 waves of four 320-thread blocks do not all find a
        // slot — a 96-register build seated three blocks.  Strategies in registers as well: 26.7 against 26.5 ms, not kept)
        if (p.block == 320 && a.k <= 4u) return launch_play_hc_t<320, false, 0, 6, false, true, 4>(p, a, s);
        return hipErrorInvalidValue;
    }
    if (p.hc_ki == 3) { // nine to twelve seats: one block per CU, increments in registers, strategies loaded per turn
        if (p.block != 768) return hipErrorInvalidValue;
        if (a.k <= 10u) return launch_play_hc_t<768, true, 10, 3, false, false, 10>(p, a, s);
        return launch_play_hc_t<768, true, 12, 3, false, false, 12>(p, a, s);
    }
    if (p.hc_ki == 2) { // four waves per SIMD: increments in registers, strategies loaded per turn
        // (register arrays and select trees sized for the launch's own seat count: five seats take four selects per increment dword
        // instead of six seats' five, and three index words instead of four — the reference's default configuration plays k = 5)
        if (a.k == 5u && p.block == 256) return launch_play_hc_t<256, true, 5, 4, false, false, 6>(p, a, s);
        if (a.k == 6u && p.block == 512) return launch_play_hc_t<512, true, 6, 4, false, false, 6>(p, a, s);
        if (a.k == 7u && p.block == 1024) return launch_play_hc_t<1024, true, 7, 4, false>(p, a, s);
        return hipErrorInvalidValue;
    }
    if (p.hc_ki && p.block == 256 && p.hc_lt) { // three waves: increments in registers, 256-thread blocks with LDS tables
        if (a.k == 8u) return launch_play_hc_t<256, true, 8>(p, a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_play(const LaunchPlan &p, const PlayArgs &a, hipStream_t s) {
    if (p.lds > LDS_LIMIT) return hipErrorInvalidValue;
    if (p.hc) return launch_play_hc(p, a, s);
    if (p.blk) return launch_play_t<768, true, 6, false, true, 2>(p, a, s); // batched H2H: k = 2, lean LDS records
    if (p.gs) { // state-store instances: the path of tables too wide for LDS records (k > 64); 2 x 768 threads per CU
        if (p.block == 768) return launch_play_t<768, true, 6, true>(p, a, s);
        return hipErrorInvalidValue;
    }
    if (p.lean) {
        switch (p.block) {
        case 1024: return launch_play_t<1024, true>(p, a, s);
        case 768: return a.k == 2u ? launch_play_t<768, true, 6, false, false, 2>(p, a, s) : launch_play_t<768, true, 6>(p, a, s);
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

// the error record of a game kernel, once it is on the host
int report_device_error(fk_ctx *c, const int32_t *h, int64_t game_base, const char *what) {
    if (h[0] == FK_ERR_ROLL_LIMIT)
        return fail(c, FK_ERR_ROLL_LIMIT, "Turn exceeded 1000 rolls - aborting. (%s game %lld)", what,
                    (long long)(game_base + h[1]));
    if (h[0] == FK_ERR_COUNTER_OVERFLOW)
        return fail(c, FK_ERR_COUNTER_OVERFLOW, "per-seat u16 counter left its guarded range (%s game %lld)", what,
                    (long long)(game_base + h[1]));
    if (h[0] != 0) return fail(c, h[0], "device error %d (%s game %lld)", h[0], what, (long long)(game_base + h[1]));
    return FK_OK;
}

// Device -> mailbox by stores (no copy engine), 4-byte words.
__global__ void fk_mail_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t n_words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// Is a rows DMA (option "rows_async") in flight or about to be queued?  Then small results leave through the mailbox.
bool rows_dma_pending(fk_ctx *c) { return c->rows_async || (c->rows_calls && hipStreamQuery(c->copy_stream) == hipErrorNotReady); }

// `bytes` (a multiple of 4) of device memory -> `dst` on the host, behind what is queued on the main stream; complete after the
// stream has been synchronised AND deliver_mail() has run.  Through the mailbox while a rows DMA is pending, else one async copy.
int post_d2h(fk_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return FK_OK;
    if ((bytes & 3u) == 0 && bytes < ((size_t)1 << 31) && rows_dma_pending(c)) {
        const size_t off = (c->mail_used + 63u) & ~(size_t)63u;
        if (off + bytes > c->mail_cap && c->letters.empty()) { // grow (nothing of the old one is awaited)
            if (c->mail) (void)hipHostFree(c->mail);
            c->mail = nullptr;
            c->mail_cap = 0;
            const size_t cap = std::max<size_t>((size_t)4 << 20, (bytes + 4095u) & ~(size_t)4095u);
            if (hipHostMalloc(reinterpret_cast<void **>(&c->mail), cap, hipHostMallocDefault) == hipSuccess) c->mail_cap = cap;
            else (void)hipGetLastError();
        }
        if (off + bytes <= c->mail_cap) {
            const uint32_t n_words = (uint32_t)(bytes / 4);
            hipLaunchKernelGGL(fk_mail_kernel, dim3(std::min<uint32_t>((n_words + 255u) / 256u, 1024u)), dim3(256), 0, c->stream,
                               static_cast<const uint32_t *>(src), reinterpret_cast<uint32_t *>(c->mail + off), n_words);
            HIPCHK(c, hipGetLastError());
            c->letters.push_back({dst, off, bytes});
            c->mail_used = off + bytes;
            return FK_OK;
        }
    }
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return FK_OK;
}

// Room for `bytes` of letters before the first one is posted (a letter that does not fit goes by the copy engine — behind a rows DMA).
void reserve_mail(fk_ctx *c, size_t bytes) {
    if (!c->letters.empty() || bytes <= c->mail_cap || !rows_dma_pending(c)) return;
    if (c->mail) (void)hipHostFree(c->mail);
    c->mail = nullptr;
    c->mail_cap = 0;
    const size_t cap = std::max<size_t>((size_t)4 << 20, (bytes + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1));
    if (hipHostMalloc(reinterpret_cast<void **>(&c->mail), cap, hipHostMallocDefault) == hipSuccess) c->mail_cap = cap;
    else (void)hipGetLastError();
}

void deliver_mail(fk_ctx *c) { // the main stream has been synchronised
    for (const auto &l : c->letters) std::memcpy(l.dst, c->mail + l.off, l.bytes);
    c->letters.clear();
    c->mail_used = 0;
}

int check_device_error(fk_ctx *c, const int32_t *d_err, int64_t game_base, const char *what) {
    int32_t h[2] = {0, 0};
    const int rc = post_d2h(c, h, d_err, sizeof(h));
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    deliver_mail(c);
    return report_device_error(c, h, game_base, what);
}

// chunk-local overrides, sorted by game id, on the device
int upload_overrides(fk_ctx *c, std::vector<DevOverride> &dov) {
    if (dov.empty()) return FK_OK;
    std::sort(dov.begin(), dov.end(), [](const DevOverride &x, const DevOverride &y) { return x.game < y.game; });
    // duplicates: the last entry of the caller's list wins (as the linear scan it replaces did)
    int rc = ensure(c, c->ov, dov.size() * sizeof(DevOverride));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->ov.p, dov.data(), dov.size() * sizeof(DevOverride), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}


// Preparation of one chunk into chunk set `si` on `st`: (batched H2H) game -> block map, schedule class sizes, seat seeding.
// Fills the pointers of `sa`.  `full_state`: 48-byte state records (state-store instances, rows, all-seat statistics).
int seed_stage(fk_ctx *c, int si, hipStream_t st, SeedArgs &sa, bool full_state, bool timed) {
    ChunkSet &cs = c->sets[si];
    sa.state_dw = full_state ? STATE_DW : 4u;
    int rc = ensure(c, cs.state, (size_t)sa.n_games * sa.k * sa.state_dw * 4);
    if (rc) return rc;
    rc = ensure(c, cs.inc, (size_t)sa.n_games * sa.k * 16);
    if (rc) return rc;
    // misc: [0] ticket counter (hammered by atomics), [16] error record, [256] schedule class cursors of the seed kernel
    rc = ensure(c, cs.misc, 512);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(cs.misc.p, 0, 512, st));
    sa.state = static_cast<uint32_t *>(cs.state.p);
    sa.inc = static_cast<uint4 *>(cs.inc.p);
    sa.seat_idx = nullptr;
    if (sa.perm_T && !full_state) { // the seats' strategy indices, resolved once by the seed kernel
        rc = ensure(c, cs.seat_idx, (size_t)sa.n_games * sa.k * 2);
        if (rc) return rc;
        sa.seat_idx = static_cast<uint16_t *>(cs.seat_idx.p);
    }
    if (sa.blocks) { // batched H2H: game -> block map first (the schedule classes and the seed kernel read it)
        hipLaunchKernelGGL(fk_block_map_kernel, dim3((sa.n_games + 255u) / 256u), dim3(256), 0, st, sa.blocks, sa.n_blocks,
                           sa.n_games, const_cast<uint32_t *>(sa.game_block), const_cast<uint32_t *>(sa.game_row));
        HIPCHK(c, hipGetLastError());
    }
    if ((sa.perm_T || sa.blocks) && c->longest_first) {
        rc = ensure(c, cs.order, (size_t)sa.n_games * 4);
        if (rc) return rc;
        sa.sched = static_cast<uint32_t *>(cs.order.p);
        sa.sched_ctr = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(cs.misc.p) + 256);
        rc = ensure(c, cs.classes, 64);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(cs.classes.p, 0, 64, st));
        sa.class_ctr = static_cast<const uint32_t *>(cs.classes.p);
        sa.patience = static_cast<const uint8_t *>(c->slow.p);
    } else {
        sa.sched = nullptr;
    }
    if (timed) (void)hipEventRecord(cs.ev[2], st);
    sa.pools = nullptr;
    if (!sa.coords) { // the pool every game of a shuffle / of a block starts from
        const uint32_t n = sa.blocks ? sa.n_blocks : (sa.gps ? (sa.n_games + sa.gps - 1u) / sa.gps : 1u);
        rc = ensure(c, cs.pools, (size_t)n * 16);
        if (rc) return rc;
        sa.pools = static_cast<const uint4 *>(cs.pools.p);
        hipLaunchKernelGGL(fk_pool_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, sa.prefix, sa.shuffle0, sa.pair, sa.order, sa.blocks, n,
                           static_cast<uint4 *>(cs.pools.p));
    }
    if (sa.sched) // class sizes first: a game's ticket is class offset + rank
        hipLaunchKernelGGL(fk_class_count_kernel, dim3(std::min<uint32_t>((sa.n_games + SEED_BLOCK - 1u) / SEED_BLOCK, 1024u)), dim3(SEED_BLOCK), 0,
                           st, sa.perm_T, sa.perm_slots, sa.S, sa.k, sa.n_sh, sa.n_games, sa.patience,
                           sa.blocks ? sa.game_row : nullptr, static_cast<uint32_t *>(cs.classes.p));
    hipLaunchKernelGGL(fk_seed_kernel, dim3((sa.n_games + SEED_BLOCK - 1u) / SEED_BLOCK), dim3(SEED_BLOCK), 0, st, sa);
    if (timed) (void)hipEventRecord(cs.ev[3], st);
    HIPCHK(c, hipGetLastError());
    return FK_OK;
}

// bytes of device workspace one game needs in a chunk (state records, increments, schedule, result record, row)
size_t game_workspace_bytes(int32_t k, bool full_state, bool recs, bool rows) {
    return (size_t)k * ((full_state ? STATE_DW * 4 : 18) + 16) + 8 + (recs ? REC_DW * 4 + 4 : 0) +
           (rows ? sizeof(fk_row_hdr) + sizeof(fk_seat) * (size_t)k : 0);
}

// The game kernel of the chunk prepared in the current chunk set, on the main stream.  `want_state`: the final state records
// of every seat must be in the state store afterwards (rows, all-seat statistics); `want_rec0` / `want_recs`: the rec0 word /
// the full result record of every game.  Launch only: the caller checks the error record (finish_play).
int launch_play_stage(fk_ctx *c, const SeedArgs &sa, PlayArgs &pa, const LaunchPlan &plan, bool want_state, bool want_rec0, bool want_recs) {
    ChunkSet &cs = CSET(c);
    int rc;
    if (want_rec0) {
        rc = ensure(c, c->rec0, (size_t)sa.n_games * 4);
        if (rc) return rc;
    }
    if (want_recs) {
        rc = ensure(c, c->recs, (size_t)sa.n_games * REC_DW * 4);
        if (rc) return rc;
    }
    pa.sched = sa.sched;
    pa.seat_idx = sa.seat_idx;
    pa.state = sa.state;
    pa.state_dw = sa.state_dw;
    pa.inc = sa.inc;
    pa.rec0 = want_rec0 ? static_cast<uint32_t *>(c->rec0.p) : nullptr;
    pa.recs = want_recs ? static_cast<uint32_t *>(c->recs.p) : nullptr;
    pa.gs_out = (want_state && !plan.gs) ? 1u : 0u;
    pa.ticket = static_cast<uint32_t *>(cs.misc.p);
    pa.err = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(cs.misc.p) + 16);
    // hand-over threshold: the measured optimum is 8 up to eight seats (rounds 1 - 3) and grows with the game length beyond (round 5,
    // 5 160-strategy grid: k = 12 at 16 -1.5 %, k = 10 at 12 -0.6 % kernel time against 8)
    const int auto_thr = sa.k >= 11 ? 16 : sa.k >= 9 ? 12 : 8;
    pa.batch_threshold = (uint32_t)std::max(1, std::min(64, c->batch_threshold > 0 ? c->batch_threshold : auto_thr));
    pa.use_lds_tally = plan.lds_tally ? 1u : 0u;
    pa.clk = nullptr;
    c->clk_grid = 0;
    if (c->clock_stamps) {
        const size_t bytes = (size_t)std::max(plan.grid, 1) * 4 * sizeof(unsigned long long);
        rc = ensure(c, c->clk, bytes);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->clk.p, 0, bytes, c->stream));
        pa.clk = static_cast<unsigned long long *>(c->clk.p);
    }
    {
        Timer t(c, &c->timing.play_ms, SLOT_PLAY);
        LaunchPlan lp = plan;
        lp.mixed_flags = (c->uniform_flags_opt != 0) ? c->table_mixed_flags : 0xff00u;
        pa.uflags = c->table_flags;
        hipError_t e = launch_play(lp, pa, c->stream);
        t.stop();
        HIPCHK(c, e);
        c->timing.play_grid = lp.launched_grid;
        c->timing.play_mixed_flags = lp.mixed_flags == MIXED_NONE ? (int32_t)MIXED_NONE
                                     : (lp.mixed_flags & ~MIXED_RB_FAV) == 0u ? (int32_t)MIXED_RB_FAV : (int32_t)MIXED_ALL;
        if (pa.clk) c->clk_grid = lp.launched_grid;
    }
    c->timing.play_launches += 1;
    c->timing.play_block = plan.block;
    c->timing.play_lds_bytes = (int32_t)plan.lds;
    c->timing.games += sa.n_games;
    return FK_OK;
}

// the kernel timers of the chunk just played (incl. those of its preparation); the main stream has been synchronised
int finish_timers(fk_ctx *c) {
    HIPCHK(c, collect_timers(c));
    if (c->clk_grid > 0) { // shader clock of the last game kernel: median over its blocks of d(s_memtime) / d(s_memrealtime) x 100 MHz
        std::vector<unsigned long long> h((size_t)c->clk_grid * 4);
        HIPCHK(c, hipMemcpy(h.data(), c->clk.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> mhz, ends;
        unsigned long long first = ~0ull;
        for (int b = 0; b < c->clk_grid; ++b) {
            const unsigned long long t0 = h[(size_t)b * 4], r0 = h[(size_t)b * 4 + 1], t1 = h[(size_t)b * 4 + 2], r1 = h[(size_t)b * 4 + 3];
            if (r1 > r0 && t1 > t0) {
                mhz.push_back(100.0 * (double)(t1 - t0) / (double)(r1 - r0));
                ends.push_back((double)r1);
                first = r0 < first ? r0 : first;
            }
        }
        if (!mhz.empty()) {
            std::nth_element(mhz.begin(), mhz.begin() + (std::ptrdiff_t)(mhz.size() / 2), mhz.end());
            c->timing.play_clock_mhz = (int32_t)(mhz[mhz.size() / 2] + 0.5);
            std::nth_element(ends.begin(), ends.begin() + (std::ptrdiff_t)(ends.size() / 2), ends.end());
            c->timing.play_block_end_p50_ms = (float)((ends[ends.size() / 2] - (double)first) * 1e-5); // 100 MHz ticks -> ms
            c->timing.play_block_end_max_ms = (float)((*std::max_element(ends.begin(), ends.end()) - (double)first) * 1e-5);
        }
        c->clk_grid = 0;
    }
    ChunkSet &cs = CSET(c);
    float ms = 0.f;
    // the permutations of a pipelined chunk ran on the main stream, in front of the previous game kernel: a clean interval
    if (hipEventElapsedTime(&ms, cs.ev[0], cs.ev[1]) == hipSuccess) c->timing.perm_ms += ms;
    if (cs.side) { // seeded behind another game kernel: no kernel-time reading for it (the interval includes the sharing)
        c->timing.prefetched_chunks += 1;
    } else if (hipEventElapsedTime(&ms, cs.ev[2], cs.ev[3]) == hipSuccess) {
        c->timing.seed_ms += ms;
    }
    return FK_OK;
}

// waits for the game kernel, reads its error record and the kernel timers
int finish_play(fk_ctx *c, const PlayArgs &pa, int64_t game_base, const char *what) {
    const int rc_dev = check_device_error(c, pa.err, game_base, what); // synchronises the main stream
    const int rc = finish_timers(c);
    return rc_dev ? rc_dev : rc;
}

// Seeds + games of one chunk (explicit game lists, batched H2H): preparation and game kernel back to back on the main stream.
int run_chunk(fk_ctx *c, const SeedArgs &sa_in, PlayArgs pa, const LaunchPlan &plan, bool want_state, bool want_rec0, bool want_recs,
              int64_t game_base, const char *what) {
    SeedArgs sa = sa_in;
    ChunkSet &cs = CSET(c);
    cs.prepared = false;
    cs.side = false;
    HIPCHK(c, hipStreamWaitEvent(c->stream, cs.ready, 0)); // an unused side-stream preparation may still own the set
    (void)hipEventRecord(cs.ev[0], c->stream); // no permutations here: an empty interval
    (void)hipEventRecord(cs.ev[1], c->stream);
    int rc = seed_stage(c, c->cur, c->stream, sa, plan.gs || want_state, true);
    if (rc) return rc;
    rc = launch_play_stage(c, sa, pa, plan, want_state, want_rec0, want_recs);
    if (rc) return rc;
    return finish_play(c, pa, game_base, what);
}

// Tournament chunk into chunk set `si`: permutations of shuffles [d.sh0, d.sh0 + d.n_sh) on the main stream, then schedule +
// seat seeding on `st`.  `sa` comes back with every pointer the game kernel and the post-passes need.
//   st == main stream: the chunk is about to be played.
//   st == prep_stream: the chunk is the NEXT one; the caller launches the current game kernel right after this returns.  The
//   permutation kernels want most of a CU's LDS, which a resident persistent game kernel does not leave them (on the side
//   stream they would only start when it ends, with the seeding behind them): they go in front of that game kernel on the
//   main stream (0.1 ms at 64 strategies, 5 ms per 10^8 games at 5 160), and only the LDS-free part — schedule classes and
//   seat seeding — shares the chip with the game kernel.
int prep_tournament_chunk(fk_ctx *c, int si, hipStream_t st_seed, const ChunkDesc &d, SeedArgs &sa) {
    ChunkSet &cs = c->sets[si];
    cs.prepared = false;
    hipStream_t st = c->stream;
    HIPCHK(c, hipStreamWaitEvent(st, cs.ready, 0)); // an unused side-stream preparation may still own the set
    const int32_t S = (int32_t)d.S;
    const uint32_t n_sh = d.n_sh, slots = d.slots, gps = d.S / d.k;
    const uint32_t perm_blocks = (n_sh + slots - 1u) / slots;
    const SeedPool perm_prefix = seed_prefix(101u /* SHUFFLE_PERMUTATION */, d.root, (uint64_t)d.k);
    int rc = ensure(c, cs.perm, (size_t)perm_blocks * S * slots * 2);
    if (rc) return rc;
    (void)hipEventRecord(cs.ev[0], st);
    {
        const size_t perm_lds = (size_t)slots * S * 2;
        static int perm_configured = -1; // per device (a second context may sit on another GPU of this process)
        if (perm_configured != c->device) {
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_perm_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_perm_apply_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_perm_parallel_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_LIMIT - 1024))); // + its static words
            perm_configured = c->device;
        }
        // large tables: the draws at full occupancy first (fk_perm_draw_kernel); then either the chain-free
        // permutation (fk_perm_parallel_kernel, one workgroup per shuffle, 14 B of LDS per strategy) or, for tables
        // beyond its LDS reach, the bare swap chains over LDS arrays (fk_perm_apply_kernel)
        const bool split = c->perm_split >= 1 || (c->perm_split < 0 && S >= 1024 && n_sh >= 256);
        const size_t pp_lds = (size_t)S * 14;
        const bool parallel = split && c->perm_split != 1 && pp_lds <= LDS_LIMIT - 1024;
        // the draws: a wave per shuffle while the launch is small (fk_perm_wave.h: ~20 us instead of the thread-per-shuffle kernel's
        // 1 ms latency), a thread per shuffle when there are enough shuffles to fill the chip with them
        const bool wave_draw = c->perm_draw_wave == 1 || (c->perm_draw_wave < 0 && n_sh <= WAVE_DRAW_MAX_SH);
        if (parallel) {
            const uint32_t groups = ((uint32_t)S - 1u + 7u) / 8u, row_u4 = groups + 1u; // a row also holds the S results
            rc = ensure(c, cs.draws, (size_t)n_sh * row_u4 * 16);
            if (rc) return rc;
            if (wave_draw)
                hipLaunchKernelGGL(fk_perm_draw_wave_kernel, dim3((n_sh + WAVE_DRAW_BLOCK / 64 - 1u) / (WAVE_DRAW_BLOCK / 64)), dim3(WAVE_DRAW_BLOCK), 0, st,
                                   perm_prefix, d.sh0, n_sh, (uint32_t)S, 1u, row_u4, static_cast<uint4 *>(cs.draws.p));
            else
                hipLaunchKernelGGL(fk_perm_draw_kernel, dim3((n_sh + DRAW_BLOCK - 1u) / DRAW_BLOCK), dim3(DRAW_BLOCK), 0, st,
                                   perm_prefix, d.sh0, n_sh, (uint32_t)S, 1u, row_u4, static_cast<uint4 *>(cs.draws.p));
            hipLaunchKernelGGL(fk_perm_parallel_kernel, dim3(n_sh), dim3(PP_BLOCK), pp_lds, st,
                               static_cast<uint4 *>(cs.draws.p), row_u4, n_sh, (uint32_t)S);
            hipLaunchKernelGGL(fk_perm_block_kernel, dim3(((uint32_t)S + 255u) / 256u, perm_blocks), dim3(256), 0, st,
                               static_cast<const uint16_t *>(cs.draws.p), row_u4 * 8u, n_sh, (uint32_t)S, slots,
                               static_cast<uint16_t *>(cs.perm.p));
        } else if (split) {
            const uint32_t n_sh_pad = (n_sh + 63u) & ~63u, groups = ((uint32_t)S - 1u + 7u) / 8u;
            rc = ensure(c, cs.draws, (size_t)groups * n_sh_pad * 16);
            if (rc) return rc;
            if (wave_draw)
                hipLaunchKernelGGL(fk_perm_draw_wave_kernel, dim3((n_sh + WAVE_DRAW_BLOCK / 64 - 1u) / (WAVE_DRAW_BLOCK / 64)), dim3(WAVE_DRAW_BLOCK), 0, st,
                                   perm_prefix, d.sh0, n_sh, (uint32_t)S, n_sh_pad, 1u, static_cast<uint4 *>(cs.draws.p));
            else
                hipLaunchKernelGGL(fk_perm_draw_kernel, dim3((n_sh + DRAW_BLOCK - 1u) / DRAW_BLOCK), dim3(DRAW_BLOCK), 0, st,
                                   perm_prefix, d.sh0, n_sh, (uint32_t)S, n_sh_pad, 1u, static_cast<uint4 *>(cs.draws.p));
            hipLaunchKernelGGL(fk_perm_apply_kernel, dim3(perm_blocks), dim3(PERM_BLOCK), perm_lds, st,
                               static_cast<const uint4 *>(cs.draws.p), n_sh_pad, n_sh, (uint32_t)S, slots,
                               static_cast<uint16_t *>(cs.perm.p));
        } else {
            hipLaunchKernelGGL(fk_perm_kernel, dim3(perm_blocks), dim3(PERM_BLOCK), perm_lds, st,
                               perm_prefix, d.sh0, n_sh, (uint32_t)S, slots, static_cast<uint16_t *>(cs.perm.p));
        }
        HIPCHK(c, hipGetLastError());
    }
    (void)hipEventRecord(cs.ev[1], st);
    sa = SeedArgs{};
    sa.prefix = seed_prefix(103u /* TOURNAMENT_PLAYER */, d.root, (uint64_t)d.k);
    sa.coords = nullptr;
    sa.shuffle0 = d.sh0;
    sa.gps = gps;
    sa.k = d.k;
    sa.n_games = n_sh * gps;
    sa.perm_T = static_cast<const uint16_t *>(cs.perm.p);
    sa.perm_slots = slots;
    sa.S = d.S;
    sa.n_sh = n_sh;
    if (st_seed != st) {
        HIPCHK(c, hipStreamWaitEvent(st_seed, cs.ev[1], 0));     // the permutations
        HIPCHK(c, hipStreamWaitEvent(st_seed, c->main_idle, 0)); // the set's previous users (recorded by the caller)
    }
    rc = seed_stage(c, si, st_seed, sa, d.state_dw == STATE_DW, true);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(cs.ready, st_seed));
    cs.desc = d;
    cs.prepared = true;
    cs.side = (st_seed == c->prep_stream);
    return FK_OK;
}

// state store + result records of the chunk just played -> rows at `d_rows` (device), game-id order
int rows_pass(fk_ctx *c, const SeedArgs &sa, bool scheduled, uint32_t n_games, uint32_t gps, uint32_t n_sh, bool perm_mode, uint8_t *d_rows) {
    // `scheduled`: the caller has inverted the schedule into c->inv (fk_invert_sched_kernel)
    const uint32_t *inv = scheduled ? static_cast<const uint32_t *>(c->inv.p) : nullptr;
    const uint32_t row_dw = 1u + 7u * sa.k;
    // rows of a block leave through an LDS tile as whole lines: the largest block whose tile fits 64 KiB
    uint32_t block = 256;
    while (block > 64 && (size_t)block * row_dw * 4 > 65536) block -= 64;
    const size_t tile = (size_t)block * row_dw * 4;
    if (tile <= 65536) {
        static int configured = -1;
        if (configured != c->device) {
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            configured = c->device;
        }
        hipLaunchKernelGGL(fk_rows_kernel<true>, dim3((n_games + block - 1u) / block), dim3(block), tile, c->stream,
                           static_cast<const uint32_t *>(CSET(c).state.p), static_cast<const uint32_t *>(c->recs.p), inv, n_games, gps, n_sh,
                           sa.k, perm_mode ? 1u : 0u, d_rows);
    } else { // tables of more than 36 seats: one thread stores its own row
        hipLaunchKernelGGL(fk_rows_kernel<false>, dim3((n_games + 255u) / 256u), dim3(256), 0, c->stream,
                           static_cast<const uint32_t *>(CSET(c).state.p), static_cast<const uint32_t *>(c->recs.p), inv, n_games, gps, n_sh,
                           sa.k, perm_mode ? 1u : 0u, d_rows);
    }
    HIPCHK(c, hipGetLastError());
    return FK_OK;
}

// ---- RCCL, bound at run time ----
// The tally reduction is the path's only exchange: one ncclReduce(sum, int64) over xGMI.  librccl is
// dlopen'ed on first use, so the library loads (and every single-GPU entry point works) on hosts without RCCL.
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, fk_comm_id, int) = nullptr; // ncclUniqueId is a 128-byte struct passed by value
    // non-blocking initialisation (a rank that never joins must not park its peers inside RCCL): optional symbols
    int (*CommInitRankConfig)(void **, int, fk_comm_id, int, ncclConfig_t *) = nullptr;
    int (*CommGetAsyncError)(void *, int *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    if (r.lib || !r.why.empty()) return r;
    // The RCCL that belongs to the HIP runtime this process actually runs on: the one next to the loaded libamdhip64
    // (a process that imported PyTorch first runs on the runtime bundled with it, and so must the communicator), then
    // the system one.
    std::vector<std::string> names;
    Dl_info info{};
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            names.push_back(dir + "/librccl.so.1");
            names.push_back(dir + "/librccl.so");
        }
    }
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) names.emplace_back(name);
    for (const auto &name : names) {
        r.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        r.why = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "not found");
        return r;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(r.lib, "ncclReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.CommInitRankConfig = reinterpret_cast<decltype(r.CommInitRankConfig)>(dlsym(r.lib, "ncclCommInitRankConfig"));
    r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.lib, "ncclCommGetAsyncError"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.Reduce || !r.CommDestroy) {
        r.why = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy";
        dlclose(r.lib);
        r.lib = nullptr;
    }
    return r;
}

int rccl_fail(fk_ctx *c, const char *what, int code) {
    Rccl &r = rccl();
    return fail(c, FK_ERR_COMM, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(code) : "RCCL error");
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Give the communicator up: abort it (ncclCommAbort ends a collective kernel that waits for a peer) and forget it.
void comm_abandon(fk_ctx *c) {
    Rccl &r = rccl();
    const bool dbg = getenv("FK_DEBUG_COMM") != nullptr;
    if (c->comm) {
        if (dbg) fprintf(stderr, "[fk comm] abandoning the communicator (%s)\n", r.CommAbort ? "ncclCommAbort" : "ncclCommDestroy");
        if (r.CommAbort) (void)r.CommAbort(c->comm);
        else (void)r.CommDestroy(c->comm);
        if (dbg) fprintf(stderr, "[fk comm] abandoned\n");
    }
    c->comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
    c->comm_async = false;
    c->comm_lost = true; // (never silently fall back to a one-rank "reduction" of a multi-rank job)
}

// A non-blocking communicator reports ncclInProgress from every call until the operation has been issued: poll its state
// under the context's deadline ("comm_timeout_ms").  On expiry or error the communicator is aborted -> FK_ERR_COMM.
int comm_settle(fk_ctx *c, const char *what, int rc, double t0) {
    Rccl &r = rccl();
    if (rc != ncclSuccess && rc != ncclInProgress) {
        const int out = rccl_fail(c, what, rc);
        comm_abandon(c);
        return out;
    }
    if (!c->comm_async) return FK_OK;
    for (;;) {
        int st = ncclSuccess;
        const int q = r.CommGetAsyncError(c->comm, &st);
        if (q != ncclSuccess || (st != ncclSuccess && st != ncclInProgress)) {
            const int out = rccl_fail(c, what, q != ncclSuccess ? q : st);
            comm_abandon(c);
            return out;
        }
        if (st == ncclSuccess) return FK_OK;
        if (now_ms() - t0 > (double)c->comm_timeout_ms) {
            comm_abandon(c);
            return fail(c, FK_ERR_COMM, "%s did not complete within %d ms (comm_timeout_ms): a peer rank never joined; communicator aborted", what,
                        c->comm_timeout_ms);
        }
        usleep(500);
    }
}

// Wait for the context's stream behind a collective, under the same deadline: a peer that never enters the collective leaves
// the RCCL kernel spinning, and hipStreamSynchronize would wait with it.
int comm_wait_stream(fk_ctx *c, const char *what, double t0) {
    hipEvent_t ev = c->ev[SLOT_CALL + 1];
    HIPCHK(c, hipEventRecord(ev, c->stream));
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return FK_OK;
        if (q != hipErrorNotReady) return fail(c, FK_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        if (c->comm_timeout_ms > 0 && now_ms() - t0 > (double)c->comm_timeout_ms) {
            comm_abandon(c);
            (void)hipStreamSynchronize(c->stream); // the aborted collective returns
            return fail(c, FK_ERR_COMM, "%s did not complete within %d ms (comm_timeout_ms): a peer rank never entered the collective; communicator aborted",
                        what, c->comm_timeout_ms);
        }
        usleep(200);
    }
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
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        bool ok = hipStreamCreateWithPriority(&c->prep_stream, hipStreamNonBlocking, least) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2; ++i) {
            ok = ok && hipEventCreateWithFlags(&c->ev_rows[i], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming) == hipSuccess;
        }
        for (int i = 0; i < FK_ROWS_EVENTS; ++i) {
            ok = ok && hipEventCreateWithFlags(&c->ev_call_copy[i], hipEventDisableTiming) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&c->main_idle, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->err_host), 2 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
        for (auto &cs : c->sets) {
            ok = ok && hipEventCreateWithFlags(&cs.ready, hipEventDisableTiming) == hipSuccess;
            for (auto &e : cs.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        }
        if (!ok) {
            fk_destroy(c);
            return FK_ERR_HIP;
        }
    }
    // score table (fk_device.h): 512 KiB, built by one small kernel, read by every roll of the game kernel
    if (ensure(c, c->score_lut, SCORE_LUT_KEYS * sizeof(uint32_t)) != FK_OK) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    if (ensure(c, c->discard_lut, DISCARD_LUT_KEYS) != FK_OK) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    hipLaunchKernelGGL(fk_score_lut_kernel, dim3(SCORE_LUT_KEYS / 256), dim3(256), 0, c->stream,
                       static_cast<uint32_t *>(c->score_lut.p));
    hipLaunchKernelGGL(fk_discard_lut_kernel, dim3(DISCARD_LUT_KEYS / 256), dim3(256), 0, c->stream,
                       static_cast<uint8_t *>(c->discard_lut.p));
    {
        const std::vector<uint8_t> img = build_lds_tables();
        if (ensure(c, c->lds_tables, img.size()) != FK_OK ||
            hipMemcpyAsync(c->lds_tables.p, img.data(), img.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            fk_destroy(c);
            return FK_ERR_HIP;
        }
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        fk_destroy(c);
        return FK_ERR_HIP;
    }
    *out = c;
    return FK_OK;
}

void fk_destroy(fk_ctx *c) {
    if (!c) return;
    for (void *p : c->hog) (void)hipFree(p);
    c->hog.clear();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr;
    release(c->comm_buf);
    if (c->prep_stream) (void)hipStreamSynchronize(c->prep_stream);
    for (DevBuf *b : {&c->strat, &c->recs, &c->rec0, &c->tally, &c->rows, &c->ov, &c->seatlist, &c->coords, &c->inv, &c->slow, &c->digest, &c->score_lut,
                      &c->discard_lut, &c->block_out, &c->stats, &c->ratios, &c->cold, &c->clk, &c->lds_tables, &c->acc, &c->rows_alt, &c->ids, &c->lag_v, &c->lag_out, &c->lag_lags, &c->lag_edge, &c->lag_tmp})
        release(*b);
    for (auto &cs : c->sets) {
        for (DevBuf *b : {&cs.perm, &cs.draws, &cs.state, &cs.inc, &cs.seat_idx, &cs.order, &cs.classes, &cs.misc, &cs.pools, &cs.blocks, &cs.game_block, &cs.game_row}) release(*b);
        if (cs.ready) (void)hipEventDestroy(cs.ready);
        for (auto &e : cs.ev)
            if (e) (void)hipEventDestroy(e);
    }
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->ev_rows[i]) (void)hipEventDestroy(c->ev_rows[i]);
        if (c->ev_copy[i]) (void)hipEventDestroy(c->ev_copy[i]);
    }
    for (auto &e : c->ev_call_copy)
        if (e) (void)hipEventDestroy(e);
    if (c->main_idle) (void)hipEventDestroy(c->main_idle);
    if (c->err_host) (void)hipHostFree(c->err_host);
    if (c->mail) (void)hipHostFree(c->mail);
    if (c->prep_stream) (void)hipStreamDestroy(c->prep_stream);
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

// Page-locked host memory for the rows / column-image destinations.  Anonymous pages populated by the kernel (mmap MAP_POPULATE: outside
// the HIP runtime, so several buffers can be made at once and no launch waits meanwhile) and then registered with the runtime — measured on
// the MI355X host for 256 MB: 12 - 21 ms + 1 - 3 ms, against 41 - 55 ms for hipHostMalloc (two of those at once: 126 ms, and every HIP call
// of the process queues behind them); device-to-host copies into either run at the same 57 GB/s.  hipHostMalloc remains the fallback.
static std::mutex g_host_lock;
static std::unordered_map<void *, size_t> g_host_mapped; // registered mmap blocks -> their length

int fk_host_alloc(fk_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return FK_ERR_ARG;
    *out = nullptr;
    const size_t page = (size_t)2 << 20, size = (std::max<size_t>(bytes, 1) + page - 1) & ~(page - 1);
    void *p = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    if (p != MAP_FAILED) {
        if (hipSetDevice(c->device) == hipSuccess && hipHostRegister(p, size, hipHostRegisterDefault) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_host_lock);
            g_host_mapped[p] = size;
            *out = p;
            return FK_OK;
        }
        (void)hipGetLastError();
        munmap(p, size);
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    return FK_OK;
}

int fk_host_free(fk_ctx *c, void *p) {
    if (!c) return FK_ERR_ARG;
    if (!p) return FK_OK;
    size_t mapped = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_lock);
        auto it = g_host_mapped.find(p);
        if (it != g_host_mapped.end()) {
            mapped = it->second;
            g_host_mapped.erase(it);
        }
    }
    if (mapped) {
        const hipError_t e = hipHostUnregister(p);
        munmap(p, mapped);
        HIPCHK(c, e);
        return FK_OK;
    }
    HIPCHK(c, hipHostFree(p));
    return FK_OK;
}

int fk_get_timing(fk_ctx *c, fk_timing *out) {
    if (!c || !out) return FK_ERR_ARG;
    *out = c->timing;
    return FK_OK;
}

// Tests of the memory budget: take device memory until about `leave_free` bytes are free (held by the context until the next call
// with leave_free < 0, or fk_destroy); free_now / total as hipMemGetInfo reports them afterwards.
int fk_debug_hold_memory(fk_ctx *c, int64_t leave_free, int64_t *free_now, int64_t *total) {
    if (!c) return FK_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (leave_free < 0) {
        for (void *p : c->hog) (void)hipFree(p);
        c->hog.clear();
    }
    size_t free_b = 0, total_b = 0;
    for (;;) {
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        if (leave_free < 0 || (int64_t)free_b - leave_free < ((int64_t)64 << 20)) break;
        void *p = nullptr;
        const size_t step = std::min<size_t>((size_t)((int64_t)free_b - leave_free), (size_t)32 << 30);
        if (hipMalloc(&p, step) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        c->hog.push_back(p);
    }
    if (free_now) *free_now = (int64_t)free_b;
    if (total) *total = (int64_t)total_b;
    return FK_OK;
}

static int32_t effective_comm_timeout_ms(const fk_ctx *c);

int fk_get_option(fk_ctx *c, const char *name, int64_t *value) {
    if (!c || !name || !value) return FK_ERR_ARG;
    const std::string n(name);
    if (n == "chunk_bytes") *value = c->chunk_bytes;
    else if (n == "workspace_percent") *value = c->workspace_percent;
    else if (n == "last_budget") *value = c->last_budget;
    else if (n == "oom_replays") *value = c->oom_replays;
    else if (n == "comm_timeout_ms") *value = effective_comm_timeout_ms(c);
    else if (n == "rows_chunk_games") *value = c->rows_chunk_games;
    else if (n == "rows_async") *value = c->rows_async;
    else if (n == "rows_event") *value = c->last_rows_event;
    else return fail(c, FK_ERR_ARG, "fk_get_option: unknown option %s", name);
    return FK_OK;
}

int fk_set_option(fk_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return FK_ERR_ARG;
    std::string n(name);
    if (n == "chunk_bytes") c->chunk_bytes = std::max<int64_t>(value, 1 << 20);
    else if (n == "workspace_percent") c->workspace_percent = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 100000);
    else if (n == "comm_timeout_ms") {
        c->comm_timeout_ms = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 86400000);
        c->comm_timeout_set = true; // from now on the FK_COMM_TIMEOUT_MS environment default is not consulted
    }
    else if (n == "batch_threshold") c->batch_threshold = (int32_t)value;
    else if (n == "use_lds_tally") c->use_lds_tally = (int32_t)value;
    else if (n == "longest_first") c->longest_first = (int32_t)value;
    else if (n == "blocks_per_cu") c->blocks_per_cu = (int32_t)value;
    else if (n == "max_waves") c->max_waves = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 8);
    else if (n == "lean") c->lean = (int32_t)value;
    else if (n == "state_store") c->gs = (int32_t)value;
    else if (n == "rows_chunk_games") c->rows_chunk_games = std::max<int64_t>(value, 1);
    else if (n == "rows_async") c->rows_async = value != 0;
    else if (n == "resident_tally") {
        c->resident = value != 0;
        c->acc_n = 0; // the next tournament call starts a fresh accumulator
    } else if (n == "hot_cold") {
        if (value != -1 && value != 0) return fail(c, FK_ERR_ARG, "hot_cold is -1 (auto: four and more seats) or 0 (never)");
        c->hc = (int32_t)value;
    }
    else if (n == "clock_stamps") c->clock_stamps = value != 0;
    else if (n == "perm_split") c->perm_split = (int32_t)value;
    else if (n == "perm_draw_wave") c->perm_draw_wave = (int32_t)value;
    else if (n == "columns_by_seat") c->columns_by_seat = (int32_t)value;
    else if (n == "pipeline") c->pipeline = (int32_t)value;
    else if (n == "uniform_flags") c->uniform_flags_opt = (int32_t)value;
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
    return fk_tournament_run_stats(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score,
                                   max_rounds, ov, n_ov, tally, rows, perms, nullptr);
}

int fk_tournament_run_columns(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed, uint64_t shuffle_begin,
                              uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score, int32_t max_rounds,
                              const fk_override *ov, int32_t n_ov, int64_t *tally, const int32_t *strategy_ids, void *columns) {
    if (!c) return FK_ERR_ARG;
    if (!strategy_ids || !columns) return fail(c, FK_ERR_ARG, "strategy_ids and columns are required");
    c->columns_ids = strategy_ids;
    const int rc = fk_tournament_run_stats(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score,
                                           max_rounds, ov, n_ov, tally, columns, nullptr, nullptr);
    c->columns_ids = nullptr;
    return rc;
}

int fk_tournament_run_columns_seeds(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed, uint64_t shuffle_begin,
                                    uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score, int32_t max_rounds,
                                    const fk_override *ov, int32_t n_ov, int64_t *tally, const int32_t *strategy_ids, void *columns,
                                    uint32_t *shuffle_seeds, uint32_t *game_seeds) {
    if (!c) return FK_ERR_ARG;
    c->want_shuffle_seeds = shuffle_seeds;
    c->want_game_seeds = game_seeds;
    const int rc = fk_tournament_run_columns(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds,
                                             ov, n_ov, tally, strategy_ids, columns);
    c->want_shuffle_seeds = c->want_game_seeds = nullptr;
    return rc;
}

int fk_rows_wait(fk_ctx *c, int32_t slot) {
    if (!c || slot < 0 || slot >= FK_ROWS_EVENTS) return FK_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_call_copy[slot]));
    return FK_OK;
}

size_t fk_row_columns_bytes(int32_t k, int32_t games_per_shuffle) {
    return (k < 1 || games_per_shuffle < 1) ? 0 : fksw::shard_image_bytes(k, games_per_shuffle);
}

int fk_write_row_shards(const fk_shard_job *job, int64_t *byte_length, uint8_t *sha256, uint8_t *sidecar_sha256, char *error, size_t error_len) {
    auto say = [&](const std::string &m) {
        if (error && error_len) std::snprintf(error, error_len, "%s", m.c_str());
        return FK_ERR_ARG;
    };
    if (!job || !byte_length || !sha256) return say("job, byte_length and sha256 are required");
    if (!job->shuffle_index || !job->shuffle_seed || !job->batch_id || !job->game_seed || !job->columns || !job->directory || !job->footer_head ||
        !job->footer_kv || !job->footer_orders || !job->leaf_type || !job->leaf_paths)
        return say("a required field of the shard job is null");
    if ((job->side_body != nullptr) != (job->side_full != nullptr) || (job->side_body && (!job->side_directory || !sidecar_sha256)))
        return say("the sidecar template needs body, full, directory and a sidecar_sha256 output");
    fksw::Job j{};
    j.k = job->k; j.gps = job->games_per_shuffle; j.n_shuffles = job->n_shuffles; j.threads = job->threads; j.atomic = job->atomic;
    j.root_seed = job->root_seed; j.rng_purpose_namespace = job->rng_purpose_namespace;
    j.shuffle_index = job->shuffle_index; j.shuffle_seed = job->shuffle_seed; j.batch_id = job->batch_id; j.game_seed = job->game_seed;
    j.columns = static_cast<const uint8_t *>(job->columns); j.shard_stride = job->shard_stride; j.directory = job->directory;
    j.footer_head = job->footer_head; j.footer_head_len = job->footer_head_len; j.footer_kv = job->footer_kv; j.footer_kv_len = job->footer_kv_len;
    j.footer_orders = job->footer_orders; j.footer_orders_len = job->footer_orders_len; j.leaf_type = job->leaf_type; j.leaf_paths = job->leaf_paths;
    j.n_leaves = job->n_leaves; j.side_body = job->side_body; j.side_full = job->side_full; j.side_directory = job->side_directory;
    std::string err;
    if (fksw::write_shards(j, byte_length, sha256, sidecar_sha256, err) != 0) {
        if (error && error_len) std::snprintf(error, error_len, "%s", err.c_str());
        return FK_ERR_IO;
    }
    return FK_OK;
}

int fk_debug_sha256(const void *data, size_t n, uint8_t *out32, int32_t portable) {
    if ((!data && n) || !out32) return FK_ERR_ARG;
    if (portable == 2) { // the two-message form the shard writer uses: data = [first half | second half], out32 = 64 bytes
        const uint8_t *p = static_cast<const uint8_t *>(data);
        fksw::sha256_pair(p, n / 2, out32, p + n / 2, n - n / 2, out32 + 32);
        return FK_OK;
    }
    fksw::sha256(static_cast<const uint8_t *>(data), n, out32, portable != 0);
    return FK_OK;
}

struct LagReq { // fk_tournament_run_lags: host pointers of the request
    const int32_t *lags;
    int32_t n_lags, max_lag;
    int64_t *sums;
    uint16_t *head, *tail;
};

static int tournament_run_impl(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                               uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                               int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                               int64_t *seat_stats, const LagReq *lag, double *seat_ratios = nullptr);

static int tournament_call(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                           uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                           int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                           int64_t *seat_stats, const LagReq *lag, double *seat_ratios = nullptr);

int fk_tournament_run_stats(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                            uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                            int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                            int64_t *seat_stats) {
    return tournament_call(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds, ov, n_ov,
                           tally, rows, perms, seat_stats, nullptr);
}

int fk_tournament_run_all_player(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                                 uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                                 int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                                 int64_t *seat_stats, double *seat_ratio_sums) {
    if (!c) return FK_ERR_ARG;
    if (!seat_stats || !seat_ratio_sums) return fail(c, FK_ERR_ARG, "seat_stats and seat_ratio_sums are required");
    return tournament_call(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds, ov, n_ov,
                           tally, rows, perms, seat_stats, nullptr, seat_ratio_sums);
}

int fk_tournament_run_lags(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed, uint64_t shuffle_begin,
                           uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score, int32_t max_rounds,
                           const fk_override *ov, int32_t n_ov, int64_t *tally, const int32_t *lags, int32_t n_lags, int64_t *lag_sums,
                           uint16_t *edge_head, uint16_t *edge_tail) {
    if (!c) return FK_ERR_ARG;
    if (!lags || !lag_sums || !edge_head || !edge_tail || n_lags < 1 || n_lags > FK_MAX_LAGS)
        return fail(c, FK_ERR_ARG, "lags, lag_sums, edge_head, edge_tail are required; 1 <= n_lags <= %d", FK_MAX_LAGS);
    for (int32_t i = 0; i < n_lags; ++i)
        if (lags[i] < 1 || lags[i] > 65535 || (i > 0 && lags[i] <= lags[i - 1]))
            return fail(c, FK_ERR_ARG, "lags must be strictly increasing positive integers (rng_diagnostic_lags, config.py:1933-1939)");
    if (max_rounds > 32767) return fail(c, FK_ERR_ARG, "lag statistics carry n_rounds in 15 bits: max_rounds must be <= 32767");
    for (int32_t i = 0; i < n_ov; ++i)
        if (ov && ov[i].max_rounds > 32767u) return fail(c, FK_ERR_ARG, "lag statistics carry n_rounds in 15 bits: override max_rounds must be <= 32767");
    const LagReq req{lags, n_lags, lags[n_lags - 1], lag_sums, edge_head, edge_tail};
    return tournament_call(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds, ov, n_ov,
                           tally, nullptr, nullptr, nullptr, &req);
}

static int tournament_call(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                           uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                           int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                           int64_t *seat_stats, const LagReq *lag, double *seat_ratios) {
    if (!c) return FK_ERR_ARG;
    c->ran_hc = false;
    c->last_tally_bytes = 0;
    c->oom = false;
    c->oom_replays = 0;
    c->chunk_limit = 0;
    int rc = tournament_run_impl(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds,
                                 ov, n_ov, tally, rows, perms, seat_stats, lag, seat_ratios);
    while (rc == FK_ERR_HIP && c->oom && c->oom_replays < 8 && c->last_budget > ((int64_t)32 << 20)) {
        // out of device memory although the budget was sized from hipMemGetInfo: somebody else's allocation came in between.  Give the
        // workspace back, plan with half, play the call again from its first shuffle (every output is overwritten).
        c->oom = false;
        ++c->oom_replays;
        release_workspace(c);
        c->chunk_limit = c->last_budget / 2;
        if (getenv("FK_DEBUG_REPLAY")) fprintf(stderr, "out of device memory: replay %d with a %lld-byte workspace\n", c->oom_replays, (long long)c->chunk_limit);
        rc = tournament_run_impl(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds,
                                 ov, n_ov, tally, rows, perms, seat_stats, lag, seat_ratios);
    }
    c->chunk_limit = 0;
    if (rc == FK_ERR_COUNTER_OVERFLOW && c->ran_hc) {
        // the hot / cold kernel's narrower counter fields (fk_play_hc.h) left their guard bands: the call is replayed on
        // fk_play_kernel, whose 16-bit fields are the ABI's stated limits
        if (getenv("FK_DEBUG_REPLAY")) fprintf(stderr, "hot / cold kernel replayed: %s\n", c->err.c_str());
        const int32_t saved = c->hc;
        c->hc = 0;
        for (auto &cs : c->sets) cs.prepared = false;
        rc = tournament_run_impl(c, strategies, S, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch, target_score, max_rounds,
                                 ov, n_ov, tally, rows, perms, seat_stats, lag, seat_ratios);
        c->hc = saved;
    }
    if (rc == 0 && c->resident && c->last_tally_bytes) {
        // the call's tally (still in c->tally) joins the resident accumulator — only now: a call that raised a device error,
        // the overflow that is replayed above included, must not have added anything (shape changes start a new accumulator)
        const size_t tally_bytes = c->last_tally_bytes, n_el = tally_bytes / sizeof(int64_t);
        if (c->acc_n != n_el) {
            rc = ensure(c, c->acc, tally_bytes);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(c->acc.p, 0, tally_bytes, c->stream));
            c->acc_n = n_el;
        }
        hipLaunchKernelGGL(fk_add_i64_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, c->stream,
                           static_cast<unsigned long long *>(c->acc.p), static_cast<const unsigned long long *>(c->tally.p), n_el);
        HIPCHK(c, hipGetLastError());
    }
    return rc;
}

static int tournament_run_impl(fk_ctx *c, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                               uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                               int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                               int64_t *seat_stats, const LagReq *lag, double *seat_ratios) {
    if (!strategies || !tally) return fail(c, FK_ERR_ARG, "strategies and tally are required");
    if (k < 1 || S < k || S % k != 0) return fail(c, FK_ERR_ARG, "n_players must divide %d", S); // run_tournament.py:274
    if (S > 65535) return fail(c, FK_ERR_ARG, "S=%d exceeds 65535 strategies", S);
    if (max_rounds < 0 || max_rounds > 65535) return fail(c, FK_ERR_ARG, "max_rounds must be in [0, 65535]");
    if (shuffle_end < shuffle_begin || shuffles_per_batch == 0) return fail(c, FK_ERR_ARG, "bad shuffle range / batch size");
    if (n_ov < 0 || (n_ov > 0 && !ov)) return fail(c, FK_ERR_ARG, "bad override list");
    HIPCHK(c, hipSetDevice(c->device));
    c->timing = fk_timing{};
    c->pending.clear();
    c->letters.clear(); // (a failed call may have left some)
    c->mail_used = 0;

    const uint64_t n_sh_total = shuffle_end - shuffle_begin;
    const uint32_t gps = (uint32_t)(S / k);
    const uint64_t n_batches = (n_sh_total + shuffles_per_batch - 1) / shuffles_per_batch;
    const size_t tally_bytes = sizeof(int64_t) * (size_t)n_batches * (size_t)S * FK_TALLY_COLS;
    if (n_sh_total == 0) return FK_OK;
    const size_t row_bytes = sizeof(fk_row_hdr) + sizeof(fk_seat) * (size_t)k;
    const bool columns = rows != nullptr && c->columns_ids != nullptr; // fk_tournament_run_columns
    const size_t rows_per_shuffle = columns ? fksw::shard_image_bytes(k, (int)gps) : (size_t)gps * row_bytes;

    int rc = upload_strategies(c, strategies, S);
    if (rc) return rc;
    if (columns) {
        if (k > 64) return fail(c, FK_ERR_ARG, "column images hold tables of at most 64 seats, got %d", (int)k);
        rc = ensure(c, c->ids, sizeof(int32_t) * (size_t)S);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->ids.p, c->columns_ids, sizeof(int32_t) * (size_t)S, hipMemcpyHostToDevice, c->stream));
    }
    rc = ensure(c, c->tally, tally_bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->tally.p, 0, tally_bytes, c->stream));

    LaunchPlan plan = plan_play(c, k, S, n_batches == 1, target_score);
    if (plan.block == 0)
        return fail(c, FK_ERR_ARG, "target_score %d: no kernel instance (lean records hold totals up to %d points; %d full records do not fit LDS)",
                    target_score, 50 * LEAN_MAX_TARGET50, (int)k);
    {
        LaunchPlan hc_plan;
        if (plan_play_hc(c, k, target_score, plan, hc_plan)) plan = hc_plan;
    }
    if (plan.hc) c->ran_hc = true;
    if (plan.hc && !plan.hc_cl) { // cold seat records of every lane the grid can seat
        rc = ensure(c, c->cold, (size_t)plan.grid * (size_t)plan.block * (size_t)k * 16);
        if (rc) return rc;
    }
    const bool want_state = rows != nullptr || seat_stats != nullptr;
    const bool want_recs = !plan.lds_tally || want_state || lag != nullptr;
    const size_t stats_bytes = sizeof(int64_t) * (size_t)n_batches * (size_t)S * FK_SEAT_STAT_COLS;
    if (seat_stats) {
        rc = ensure(c, c->stats, stats_bytes);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->stats.p, 0, stats_bytes, c->stream));
    }
    const size_t ratio_bytes = sizeof(double) * (size_t)n_batches * (size_t)S * FK_SEAT_RATIO_COLS;
    if (seat_ratios) { // running float64 sums, carried from chunk to chunk on the device (all-zero bits = 0.0)
        rc = ensure(c, c->ratios, ratio_bytes);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->ratios.p, 0, ratio_bytes, c->stream));
    }

    // chunk planning: whole shuffles per chunk inside the workspace budget
    const size_t bytes_per_shuffle = (size_t)S * 2 + (size_t)gps * (game_workspace_bytes(k, plan.gs || want_state, want_recs, rows != nullptr) +
                                                                    (seat_stats ? (size_t)k * 32 : 0)) // + the exposure digests
                                     + (lag ? (size_t)S * 2 : 0);                                        // + the lag value matrix row
    // (column images are larger than AoS rows: the workspace figure above counts 4 + 28 k bytes per game)
    uint64_t chunk_sh = std::max<uint64_t>(1, (uint64_t)workspace_budget(c) / bytes_per_shuffle);
    chunk_sh = std::min<uint64_t>(chunk_sh, (uint64_t)0x7fffffff / gps);
    if (rows) // rows mode: several chunks per call, so that the rows of chunk i cross PCIe while chunk i + 1 plays
        chunk_sh = std::min<uint64_t>(chunk_sh, std::max<uint64_t>(1, (uint64_t)c->rows_chunk_games / gps));
    chunk_sh = std::min<uint64_t>(chunk_sh, n_sh_total);

    const hipEvent_t t0 = c->ev[SLOT_CALL], t1 = c->ev[SLOT_CALL + 1];
    HIPCHK(c, hipEventRecord(t0, c->stream));

    // lag statistics (fk_tournament_run_lags): V = [max_lag carry rows][chunk rows] x S values, sums [S][n_lags][11]
    const uint32_t L = lag ? (uint32_t)lag->max_lag : 0u;
    const uint32_t edge_rows = lag ? (uint32_t)std::min<uint64_t>(L, n_sh_total) : 0u;
    const size_t lag_sum_bytes = lag ? sizeof(int64_t) * (size_t)S * (size_t)lag->n_lags * FK_LAG_COLS : 0;
    uint32_t head_filled = 0;
    if (lag) {
        if ((rc = ensure(c, c->lag_v, ((size_t)L + chunk_sh) * (size_t)S * 2))) return rc;
        if ((rc = ensure(c, c->lag_out, lag_sum_bytes))) return rc;
        if ((rc = ensure(c, c->lag_lags, sizeof(int32_t) * (size_t)lag->n_lags))) return rc;
        if ((rc = ensure(c, c->lag_edge, 2 * (size_t)std::max<uint32_t>(edge_rows, 1) * (size_t)S * 2))) return rc;
        if ((rc = ensure(c, c->lag_tmp, (size_t)std::max<uint32_t>(L, 1) * (size_t)S * 2))) return rc;
        HIPCHK(c, hipMemsetAsync(c->lag_out.p, 0, lag_sum_bytes, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->lag_lags.p, lag->lags, sizeof(int32_t) * (size_t)lag->n_lags, hipMemcpyHostToDevice, c->stream));
    }

    const uint32_t slots = (uint32_t)std::max<size_t>(1, std::min<size_t>(PERM_BLOCK, LDS_LIMIT / ((size_t)S * 2)));
    const uint32_t state_dw = (plan.gs || want_state) ? STATE_DW : 4u;
    auto describe = [&](uint64_t first_shuffle, uint32_t count, uint32_t dw) {
        ChunkDesc d;
        d.epoch = c->table_epoch;
        d.root = root_seed;
        d.sh0 = first_shuffle;
        d.n_sh = count;
        d.S = (uint32_t)S;
        d.k = (uint32_t)k;
        d.state_dw = dw;
        d.sched = c->longest_first ? 1u : 0u;
        d.slots = slots;
        return d;
    };
    // the hint (fk_tournament_hint_next) is for the call that FOLLOWS this one
    const bool hinted = c->hint_valid && c->hint_end > c->hint_begin;
    const uint64_t hint_begin = c->hint_begin, hint_end = c->hint_end;
    const uint32_t hint_dw = (plan.gs || c->hint_state) ? STATE_DW : 4u;
    c->hint_valid = false;

    std::vector<uint16_t> perm_host;
    bool deferred = false;
    int64_t deferred_base = 0;
    for (uint64_t done = 0; done < n_sh_total; done += chunk_sh) {
        const uint32_t n_sh = (uint32_t)std::min<uint64_t>(chunk_sh, n_sh_total - done);
        const uint64_t sh0 = shuffle_begin + done;
        const uint32_t n_games = n_sh * gps;
        const uint32_t perm_blocks = (n_sh + slots - 1u) / slots;

        // the chunk's preparation: already made on the side stream (by the previous chunk, or by the previous call after a
        // hint), or made now in front of the game kernel
        const ChunkDesc d = describe(sh0, n_sh, state_dw);
        int ready_set = -1;
        for (int si : {c->cur ^ 1, c->cur})
            if (c->sets[si].prepared && c->sets[si].desc == d) ready_set = si;
        // A chunk prepared on the side stream: its seat seeding may still be running in the previous game kernel's drain tail.
        // The main stream waits for it only in front of THIS chunk's game kernel (below) — what is enqueued before that point,
        // the next chunk's permutation kernels above all, runs beside the seeding's tail instead of behind it.
        bool wait_ready = false;
        if (ready_set >= 0) {
            c->cur = ready_set;
            wait_ready = true;
        } else {
            if (c->sets[c->cur].prepared) c->cur ^= 1; // keep a prepared (hinted) chunk for its own call if there is room
            rc = prep_tournament_chunk(c, c->cur, c->stream, d, CSET(c).sa);
            if (rc) return rc;
        }
        CSET(c).prepared = false; // consumed
        const SeedArgs sa = CSET(c).sa;
        if (perms) {
            if (wait_ready) HIPCHK(c, hipStreamWaitEvent(c->stream, CSET(c).ready, 0));
            wait_ready = false;
            perm_host.resize((size_t)perm_blocks * S * slots);
            HIPCHK(c, hipMemcpyAsync(perm_host.data(), CSET(c).perm.p, perm_host.size() * 2, hipMemcpyDeviceToHost, c->stream));
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
            const uint32_t game = (uint32_t)((ov[i].a - sh0) * gps + ov[i].b);
            bool replaced = false;
            for (auto &dv : dov)
                if (dv.game == game) {
                    dv.max_rounds = ov[i].max_rounds;
                    replaced = true;
                }
            if (!replaced) dov.push_back(DevOverride{game, ov[i].max_rounds});
        }
        rc = upload_overrides(c, dov);
        if (rc) return rc;
        // the device row buffers alternate over chunks — and, in async mode, over calls: a one-chunk call's row kernel then has not to wait for the previous call's copy
        const int rb = (int)((done / chunk_sh + (c->rows_async ? c->rows_calls : 0u)) & 1u);
        DevBuf &row_buf = rb ? c->rows_alt : c->rows;
        if (rows) {
            if ((size_t)n_sh * rows_per_shuffle > row_buf.cap) HIPCHK(c, hipStreamSynchronize(c->copy_stream)); // growing: no copy may be reading it
            rc = ensure(c, row_buf, (size_t)n_sh * rows_per_shuffle);
            if (rc) return rc;
        }

        PlayArgs pa{};
        pa.strat = static_cast<const uint2 *>(c->strat.p);
        pa.score_lut = static_cast<const uint32_t *>(c->score_lut.p);
        pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
        pa.perm_T = static_cast<const uint16_t *>(CSET(c).perm.p);
        pa.perm_slots = slots;
        pa.seat_strategy = nullptr;
        pa.tally = static_cast<unsigned long long *>(c->tally.p);
        pa.ov = static_cast<const DevOverride *>(c->ov.p);
        pa.n_ov = (uint32_t)dov.size();
        pa.mode = MODE_PERM;
        pa.n_games = n_games;
        pa.gps = gps;
        pa.n_sh = n_sh;
        pa.k = (uint32_t)k;
        pa.S = (uint32_t)S;
        pa.target50 = ceil_div50(target_score);
        pa.beat50 = floor_div50(target_score);
        pa.max_rounds = (uint32_t)max_rounds;
        pa.cold = static_cast<uint4 *>(c->cold.p);
        pa.lds_tables = static_cast<const uint8_t *>(c->lds_tables.p);

        // The next chunk (or the hinted next call) is prepared into the other chunk set around this game kernel: its permutations
        // in front of it on the main stream, its schedule + seat seeding on the low-priority stream while it runs.
        HIPCHK(c, hipEventRecord(c->main_idle, c->stream)); // everything that used the other chunk set is in front of this point
        if (c->pipeline) {
            ChunkDesc next{};
            bool have_next = false;
            if (done + chunk_sh < n_sh_total) {
                next = describe(sh0 + n_sh, (uint32_t)std::min<uint64_t>(chunk_sh, n_sh_total - done - chunk_sh), state_dw);
                have_next = true;
            } else if (hinted) {
                next = describe(hint_begin, (uint32_t)std::min<uint64_t>(chunk_sh, hint_end - hint_begin), hint_dw);
                have_next = true;
            }
            if (have_next) {
                rc = prep_tournament_chunk(c, c->cur ^ 1, c->prep_stream, next, c->sets[c->cur ^ 1].sa);
                if (rc) return rc;
            }
        }
        if (wait_ready) HIPCHK(c, hipStreamWaitEvent(c->stream, c->sets[c->cur].ready, 0));
        rc = launch_play_stage(c, sa, pa, plan, want_state, want_recs, want_recs);
        if (rc) return rc;
        // The last chunk of a call without rows: its error record travels with the tally, behind the post-passes — one host
        // round trip per call instead of two (the post-passes only read; on an error their output is discarded).
        const bool defer_check = done + chunk_sh >= n_sh_total && !rows && c->err_host;
        if (defer_check) {
            HIPCHK(c, hipMemcpyAsync(c->err_host, pa.err, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            deferred = true;
            deferred_base = (int64_t)done * gps;
        } else {
            rc = finish_play(c, pa, (int64_t)done * gps, "tournament");
            if (rc) return rc;
        }
        const bool scheduled = c->longest_first != 0;
        if (want_state && scheduled) { // game id -> slot of its state records
            rc = ensure(c, c->inv, (size_t)n_games * 4);
            if (rc) return rc;
            hipLaunchKernelGGL(fk_invert_sched_kernel, dim3((n_games + 255u) / 256u), dim3(256), 0, c->stream,
                               static_cast<const uint32_t *>(CSET(c).order.p), n_games, static_cast<uint32_t *>(c->inv.p));
        }
        if (seat_stats) {
            rc = ensure(c, CSET(c).draws, (size_t)perm_blocks * S * slots * 2); // the draws buffer is free again: inverse permutations
            if (rc) return rc;
            const uint32_t cells = perm_blocks * (uint32_t)S * slots;
            hipLaunchKernelGGL(fk_invert_perm_kernel, dim3((cells + 255u) / 256u), dim3(256), 0, c->stream,
                               static_cast<const uint16_t *>(CSET(c).perm.p), (uint32_t)S, slots, n_sh, static_cast<uint16_t *>(CSET(c).draws.p));
            const uint32_t first_batch = (uint32_t)(done / shuffles_per_batch);
            const uint32_t nb = (uint32_t)((done + n_sh - 1) / shuffles_per_batch) - first_batch + 1u;
            const uint32_t s_blocks = ((uint32_t)S + 255u) / 256u;
            // enough (strategy block, batch, part) workgroups to fill the chip; a part is at least 8 shuffles
            const uint32_t ppb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(shuffles_per_batch, n_sh) / 8, (4096u + nb * s_blocks - 1u) / (nb * s_blocks)));
            // phase 1, game-major: one 32-byte digest per exposure; phase 2 gathers them per strategy
            rc = ensure(c, c->digest, (size_t)n_games * k * 32);
            if (rc) return rc;
            const size_t pairs = (size_t)n_games * k;
            hipLaunchKernelGGL(fk_seat_digest_kernel, dim3((unsigned)((pairs + 255u) / 256u)), dim3(256), 0, c->stream,
                               static_cast<const uint32_t *>(CSET(c).state.p), static_cast<const uint32_t *>(c->recs.p),
                               scheduled ? static_cast<const uint32_t *>(c->inv.p) : nullptr, n_games, gps, n_sh, (uint32_t)k,
                               static_cast<uint4 *>(c->digest.p));
            hipLaunchKernelGGL(fk_seat_stats_kernel, dim3(s_blocks, nb * ppb), dim3(256), 0, c->stream,
                               static_cast<const uint4 *>(c->digest.p), static_cast<const uint16_t *>(CSET(c).draws.p),
                               slots, (uint32_t)S, (uint32_t)k, gps, n_sh, (uint32_t)done, shuffles_per_batch, ppb, first_batch,
                               static_cast<long long *>(c->stats.p));
            if (seat_ratios) // the four float64 sums: one thread per (strategy, batch), the batch's shuffles of this chunk in order
                hipLaunchKernelGGL(fk_seat_ratio_kernel, dim3(s_blocks, nb), dim3(256), 0, c->stream,
                                   static_cast<const uint4 *>(c->digest.p), static_cast<const uint16_t *>(CSET(c).draws.p),
                                   slots, (uint32_t)S, (uint32_t)k, gps, n_sh, (uint32_t)done, shuffles_per_batch, first_batch,
                                   static_cast<double *>(c->ratios.p));
            HIPCHK(c, hipGetLastError());
        }
        if (lag) {
            uint16_t *V = static_cast<uint16_t *>(c->lag_v.p);
            const size_t row = (size_t)S; // values per row
            const uint32_t carry = (uint32_t)std::min<uint64_t>(done, L);
            hipLaunchKernelGGL(fk_lag_values_kernel, dim3((unsigned)(((size_t)n_games * k + 255u) / 256u)), dim3(256), 0, c->stream,
                               static_cast<const uint32_t *>(c->recs.p), static_cast<const uint16_t *>(CSET(c).perm.p), slots, (uint32_t)S,
                               (uint32_t)k, gps, n_games, V + (size_t)L * row);
            const uint32_t s_blocks = ((uint32_t)S + 255u) / 256u;
            // enough (strategy block, segment) workgroups to fill the chip, at least 64 rows per segment
            const uint32_t want_seg = std::max<uint32_t>(1u, (2048u + s_blocks - 1u) / s_blocks);
            const uint32_t rows_per_seg = std::max<uint32_t>(64u, (n_sh + want_seg - 1u) / want_seg);
            const uint32_t n_seg = (n_sh + rows_per_seg - 1u) / rows_per_seg;
            hipLaunchKernelGGL(fk_lag_sums_kernel, dim3(s_blocks, n_seg), dim3(256), 0, c->stream, V, (uint32_t)S, L - carry, L, L + n_sh,
                               rows_per_seg, static_cast<const int32_t *>(c->lag_lags.p), (uint32_t)lag->n_lags,
                               static_cast<long long *>(c->lag_out.p));
            HIPCHK(c, hipGetLastError());
            uint16_t *edge = static_cast<uint16_t *>(c->lag_edge.p);
            if (head_filled < edge_rows) { // the first rows of the call's range (they may span chunks shorter than the largest lag)
                const uint32_t take = std::min<uint32_t>(edge_rows - head_filled, n_sh);
                HIPCHK(c, hipMemcpyAsync(edge + (size_t)head_filled * row, V + (size_t)L * row, (size_t)take * row * 2, hipMemcpyDeviceToDevice, c->stream));
                head_filled += take;
            }
            if (done + chunk_sh >= n_sh_total) { // last chunk: the last rows of the range (carry rows + this chunk's)
                HIPCHK(c, hipMemcpyAsync(edge + (size_t)edge_rows * row, V + ((size_t)L + n_sh - edge_rows) * row, (size_t)edge_rows * row * 2,
                                         hipMemcpyDeviceToDevice, c->stream));
            } else { // the next chunk's carry = the last L rows of [carry | chunk] (through a scratch copy: the ranges may overlap)
                HIPCHK(c, hipMemcpyAsync(c->lag_tmp.p, V + (size_t)n_sh * row, (size_t)L * row * 2, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(V, c->lag_tmp.p, (size_t)L * row * 2, hipMemcpyDeviceToDevice, c->stream));
            }
        }
        if (!plan.lds_tally) { // result records -> per-batch tallies
            const uint64_t games_per_batch = (uint64_t)shuffles_per_batch * gps;
            unsigned long long *d_tally = static_cast<unsigned long long *>(c->tally.p);
            if (games_per_batch >= 4096) {
                // 896 x 22 x 8 B = 154 KiB of LDS: one 1024-thread workgroup per CU, half as many passes over rec0 as two
                // smaller ones would need
                const uint32_t slice = (uint32_t)std::min<int64_t>(S, 896);
                const uint32_t n_slices = ((uint32_t)S + slice - 1u) / slice;
                const uint32_t first_batch = (uint32_t)(done / shuffles_per_batch);
                const uint32_t last_batch = (uint32_t)((done + n_sh - 1) / shuffles_per_batch);
                const uint32_t nb = last_batch - first_batch + 1u;
                // enough parts to fill the chip, at least ~16 K games each
                uint32_t ppb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(games_per_batch / 16384, (2048u + nb * n_slices - 1u) / (nb * n_slices)));
                static int reduce_configured = -1;
                if (reduce_configured != c->device) {
                    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_tally_reduce_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
                    reduce_configured = c->device;
                }
                hipLaunchKernelGGL(fk_tally_reduce_kernel, dim3(ppb * nb, n_slices), dim3(REDUCE_BLOCK), (size_t)slice * RT_COLS * 8, c->stream,
                                   static_cast<const uint32_t *>(c->rec0.p), static_cast<const uint32_t *>(c->recs.p), n_games, gps,
                                   (uint32_t)k, (uint32_t)S, static_cast<const uint16_t *>(CSET(c).perm.p), slots, (uint32_t)done,
                                   shuffles_per_batch, n_sh, ppb, slice, first_batch, d_tally);
            } else {
                hipLaunchKernelGGL(fk_tally_direct_kernel, dim3((n_games + 255u) / 256u), dim3(256), 0, c->stream,
                                   static_cast<const uint32_t *>(c->recs.p), n_games, gps, (uint32_t)k, (uint32_t)S,
                                   static_cast<const uint16_t *>(CSET(c).perm.p), slots, (uint32_t)done, shuffles_per_batch, d_tally);
            }
            HIPCHK(c, hipGetLastError());
        }
        if (rows) {
            // rows kernel on the main stream (behind this buffer's previous copy), the copy to the host on the copy stream: it
            // runs beside the next chunk's game kernel.  With a pinned destination (fk_host_alloc) it is one DMA at PCIe rate.
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy[rb], 0));
            if (columns && k <= 16 && c->columns_by_seat != 0) {
                hipLaunchKernelGGL(fk_row_columns_seats_kernel, dim3((n_games + 63u) / 64u), dim3(64u * (uint32_t)k), 0, c->stream,
                                   static_cast<const uint32_t *>(CSET(c).state.p), static_cast<const uint32_t *>(c->recs.p),
                                   scheduled ? static_cast<const uint32_t *>(c->inv.p) : nullptr, n_games, gps, n_sh, (uint32_t)k, 1u,
                                   static_cast<const int32_t *>(c->ids.p), static_cast<uint8_t *>(row_buf.p), rows_per_shuffle);
                HIPCHK(c, hipGetLastError());
            } else if (columns) {
                hipLaunchKernelGGL(fk_row_columns_kernel, dim3((n_games + 255u) / 256u), dim3(256), 0, c->stream,
                                   static_cast<const uint32_t *>(CSET(c).state.p), static_cast<const uint32_t *>(c->recs.p),
                                   scheduled ? static_cast<const uint32_t *>(c->inv.p) : nullptr, n_games, gps, n_sh, (uint32_t)k, 1u,
                                   static_cast<const int32_t *>(c->ids.p), static_cast<uint8_t *>(row_buf.p), rows_per_shuffle);
                HIPCHK(c, hipGetLastError());
            } else {
                rc = rows_pass(c, sa, scheduled, n_games, gps, n_sh, true, static_cast<uint8_t *>(row_buf.p));
                if (rc) return rc;
            }
            HIPCHK(c, hipEventRecord(c->ev_rows[rb], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_rows[rb], 0));
            HIPCHK(c, hipMemcpyAsync(static_cast<uint8_t *>(rows) + (size_t)done * rows_per_shuffle, row_buf.p,
                                     (size_t)n_sh * rows_per_shuffle, hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(c, hipEventRecord(c->ev_copy[rb], c->copy_stream));
        }
    }
    if (rows && c->rows_async) { // the caller waits (fk_rows_wait): the next call's game kernel may run beside this call's last copy
        c->last_rows_event = (int32_t)(c->rows_calls % FK_ROWS_EVENTS);
        HIPCHK(c, hipEventRecord(c->ev_call_copy[c->last_rows_event], c->copy_stream));
        ++c->rows_calls;
    } else if (rows) {
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    }
    const uint32_t n_rows = (uint32_t)(n_batches * (uint64_t)S);
    hipLaunchKernelGGL(fk_finalize_tally, dim3((n_rows + 255u) / 256u), dim3(256), 0, c->stream,
                       static_cast<unsigned long long *>(c->tally.p), n_rows, (uint32_t)S, shuffles_per_batch, n_sh_total, 1u);
    HIPCHK(c, hipGetLastError());
    c->last_tally_bytes = tally_bytes; // (fk_tournament_run_stats adds it to the resident accumulator once the call has succeeded)
    HIPCHK(c, hipEventRecord(t1, c->stream));
    const size_t game_seed_bytes = c->want_game_seeds ? (size_t)n_sh_total * gps * 4 : 0, shuffle_seed_bytes = c->want_shuffle_seeds ? (size_t)n_sh_total * 4 : 0;
    reserve_mail(c, tally_bytes + game_seed_bytes + shuffle_seed_bytes + 256);
    rc = post_d2h(c, tally, c->tally.p, tally_bytes);
    if (rc) return rc;
    if (game_seed_bytes + shuffle_seed_bytes) {
        // the fingerprints the row shards carry (game_seed column: ns 102, run_tournament.py:340-350) and their manifest names
        // (shuffle_seed: ns 100 = the same coordinate with game_index 0): with the tally, one host round trip for the launch group
        const size_t shuffle_at = (game_seed_bytes + 255u) & ~(size_t)255u;
        rc = ensure(c, c->dbg[5], shuffle_at + shuffle_seed_bytes);
        if (rc) return rc;
        uint8_t *d_seeds = static_cast<uint8_t *>(c->dbg[5].p);
        if (game_seed_bytes) {
            const size_t n = (size_t)n_sh_total * gps;
            hipLaunchKernelGGL(fk_game_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, seed_prefix(102u /* TOURNAMENT_GAME */, root_seed, (uint64_t)k),
                               shuffle_begin, (uint32_t)n_sh_total, gps, reinterpret_cast<uint32_t *>(d_seeds));
            HIPCHK(c, hipGetLastError());
            rc = post_d2h(c, c->want_game_seeds, d_seeds, game_seed_bytes);
            if (rc) return rc;
        }
        if (shuffle_seed_bytes) {
            hipLaunchKernelGGL(fk_game_seed_kernel, dim3((unsigned)((n_sh_total + 255) / 256)), dim3(256), 0, c->stream, seed_prefix(100u /* TOURNAMENT_SHUFFLE */, root_seed, (uint64_t)k),
                               shuffle_begin, (uint32_t)n_sh_total, 1u, reinterpret_cast<uint32_t *>(d_seeds + shuffle_at));
            HIPCHK(c, hipGetLastError());
            rc = post_d2h(c, c->want_shuffle_seeds, d_seeds + shuffle_at, shuffle_seed_bytes);
            if (rc) return rc;
        }
    }
    if (seat_stats) HIPCHK(c, hipMemcpyAsync(seat_stats, c->stats.p, stats_bytes, hipMemcpyDeviceToHost, c->stream));
    if (seat_ratios) HIPCHK(c, hipMemcpyAsync(seat_ratios, c->ratios.p, ratio_bytes, hipMemcpyDeviceToHost, c->stream));
    if (lag) {
        const size_t edge_bytes = (size_t)edge_rows * (size_t)S * 2;
        HIPCHK(c, hipMemcpyAsync(lag->sums, c->lag_out.p, lag_sum_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(lag->head, c->lag_edge.p, edge_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(lag->tail, static_cast<uint8_t *>(c->lag_edge.p) + edge_bytes, edge_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    deliver_mail(c);
    if (deferred) {
        rc = report_device_error(c, c->err_host, deferred_base, "tournament");
        const int rc_t = finish_timers(c);
        if (rc) return rc;
        if (rc_t) return rc_t;
    }
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
    c->letters.clear(); // (a failed call may have left some)
    c->mail_used = 0;
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
    HIPCHK(c, hipMemcpyAsync(c->coords.p, coords, sizeof(fk_coord) * (size_t)n_games, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->seatlist.p, seat_strategy, sizeof(int32_t) * (size_t)n_games * k, hipMemcpyHostToDevice, c->stream));

    LaunchPlan plan = plan_play(c, k, S, false, target_score);
    if (plan.block == 0)
        return fail(c, FK_ERR_ARG, "target_score %d: no kernel instance (lean records hold totals up to %d points; %d full records do not fit LDS)",
                    target_score, 50 * LEAN_MAX_TARGET50, (int)k);

    SeedArgs sa{};
    sa.coords = static_cast<const fk_coord *>(c->coords.p);
    sa.seat_strategy = static_cast<const int32_t *>(c->seatlist.p);
    sa.k = (uint32_t)k;
    sa.n_games = (uint32_t)n_games;

    PlayArgs pa{};
    pa.strat = static_cast<const uint2 *>(c->strat.p);
    pa.score_lut = static_cast<const uint32_t *>(c->score_lut.p);
    pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
    pa.lds_tables = static_cast<const uint8_t *>(c->lds_tables.p);
    pa.seat_strategy = static_cast<const int32_t *>(c->seatlist.p);
    pa.mode = MODE_LIST;
    pa.n_games = (uint32_t)n_games;
    pa.gps = 1;
    pa.n_sh = 1;
    pa.k = (uint32_t)k;
    pa.S = (uint32_t)S;
    pa.target50 = ceil_div50(target_score);
    pa.beat50 = floor_div50(target_score);
    pa.max_rounds = (uint32_t)max_rounds;
    rc = run_chunk(c, sa, pa, plan, true, true, true, 0, "list");
    if (rc) return rc;
    rc = rows_pass(c, sa, false, (uint32_t)n_games, 1, 1, false, static_cast<uint8_t *>(c->rows.p));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rows, c->rows.p, (size_t)n_games * row_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->timing.total_ms = c->timing.seed_ms + c->timing.play_ms;
    return FK_OK;
}

// Many H2H blocks advanced together.  Each pass plays, for every block that still needs games, exactly
// min(target - completed, stop - attempted) attempts in attempt order: a pass can never overshoot the prefix rule (the
// first attempt index at which `target` games have completed, h2h_schedule.py:1165-1235), because it reaches the target
// only if every one of its attempts completes, i.e. at its last attempt.  Blocks with safety-limit games need further
// (geometrically smaller) passes; all blocks share each pass's launch.
static int h2h_run_blocks_impl(fk_ctx *c, fk_h2h_block *blocks, int64_t n_blocks, uint64_t root_seed, uint64_t chunk_games,
                               int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov);

int fk_h2h_run_blocks(fk_ctx *c, fk_h2h_block *blocks, int64_t n_blocks, uint64_t root_seed, uint64_t chunk_games,
                      int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov) {
    if (!c) return FK_ERR_ARG;
    c->oom = false;
    c->oom_replays = 0;
    c->chunk_limit = 0;
    // the blocks are advanced in place: a replay after an out-of-memory failure starts from the states the caller handed in
    std::vector<fk_h2h_block> saved;
    if (blocks && n_blocks > 0 && n_blocks <= (1 << 22)) saved.assign(blocks, blocks + n_blocks);
    int rc = h2h_run_blocks_impl(c, blocks, n_blocks, root_seed, chunk_games, target_score, max_rounds, ov, n_ov);
    while (rc == FK_ERR_HIP && c->oom && c->oom_replays < 8 && c->last_budget > ((int64_t)32 << 20)) {
        c->oom = false;
        ++c->oom_replays;
        release_workspace(c);
        c->chunk_limit = c->last_budget / 2;
        std::copy(saved.begin(), saved.end(), blocks);
        if (getenv("FK_DEBUG_REPLAY")) fprintf(stderr, "out of device memory: replay %d with a %lld-byte workspace\n", c->oom_replays, (long long)c->chunk_limit);
        rc = h2h_run_blocks_impl(c, blocks, n_blocks, root_seed, chunk_games, target_score, max_rounds, ov, n_ov);
    }
    c->chunk_limit = 0;
    return rc;
}

static int h2h_run_blocks_impl(fk_ctx *c, fk_h2h_block *blocks, int64_t n_blocks, uint64_t root_seed, uint64_t chunk_games,
                               int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov) {
    if (!blocks || n_blocks < 0 || n_blocks > (1 << 22)) return fail(c, FK_ERR_ARG, "blocks are required (at most 2^22 per call)");
    if (max_rounds < 0 || max_rounds > 65535) return fail(c, FK_ERR_ARG, "max_rounds must be in [0, 65535]");
    if (n_ov < 0 || (n_ov > 0 && !ov)) return fail(c, FK_ERR_ARG, "bad override list");
    HIPCHK(c, hipSetDevice(c->device));
    c->timing = fk_timing{};
    c->pending.clear();
    c->letters.clear(); // (a failed call may have left some)
    c->mail_used = 0;
    if (n_blocks == 0) return FK_OK;
    std::vector<fk_strategy> table((size_t)n_blocks * 2);
    std::vector<uint64_t> stop((size_t)n_blocks);
    for (int64_t b = 0; b < n_blocks; ++b) {
        const fk_h2h_block &blk = blocks[b];
        if (blk.order > 1u) return fail(c, FK_ERR_ARG, "block %lld: order must be 0 or 1", (long long)b);
        const uint64_t attempted = blk.state[0], completed = blk.state[1], safety = blk.state[2], w1 = blk.state[3], w2 = blk.state[4];
        if (!(completed <= blk.target) || !(attempted <= blk.max_attempts) || completed + safety != attempted || w1 + w2 != completed)
            return fail(c, FK_ERR_ARG, "block %lld: inconsistent block state", (long long)b); // h2h_schedule.py:1366, 1422-1468
        uint64_t st = attempted + chunk_games; // :1172
        if (st > blk.max_attempts || st < attempted) st = blk.max_attempts;
        stop[(size_t)b] = st;
        table[(size_t)b * 2] = blk.seats[0];
        table[(size_t)b * 2 + 1] = blk.seats[1];
    }
    int rc = upload_strategies(c, table.data(), n_blocks * 2);
    if (rc) return rc;
    const SeedPool seat_prefix = seed_prefix(203u /* H2H_PLAYER */, root_seed, 2u);
    const LaunchPlan plan = plan_play(c, 2, n_blocks * 2, false, target_score, true);
    if (plan.block == 0)
        return fail(c, FK_ERR_ARG, "target_score %d: batched head-to-head plays with lean records (totals up to %d points)", target_score,
                    50 * LEAN_MAX_TARGET50);
    const uint64_t max_launch = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)workspace_budget(c) / (game_workspace_bytes(2, plan.gs, false, false) + 8), 1u << 30));
    rc = ensure(c, c->block_out, (size_t)n_blocks * 4 * 8);
    if (rc) return rc;

    // A GENERATION plays, for every block that still needs games, exactly min(stop - attempted, target - completed) attempts —
    // all of them planned up front from the blocks' current states and cut into passes (launches) of at most max_launch games.
    // The passes of a generation do not depend on each other's results, so pass i + 1 is prepared (block table, game -> block
    // map, pools, schedule classes, seat seeding) into the other chunk set on the low-priority stream while pass i plays, as
    // tournament chunks are (15.5 ms of seeding per 6 x 10^8-attempt launch of BASELINE config 5 used to sit between the game
    // kernels); the strategy table (two rows per block of the CALL) and the per-block counts are per call, the error records
    // are read once per generation.  Blocks with safety-limit games need further (geometrically smaller) generations.
    struct Pass {
        std::vector<DevBlock> blocks;
        uint32_t n_games = 0;
    };
    constexpr size_t MAX_PASS_BLOCKS = (size_t)1 << 22;
    std::vector<unsigned long long> out((size_t)n_blocks * 4);
    std::vector<uint64_t> planned((size_t)n_blocks);
    const bool pipelined = c->pipeline != 0 && n_ov == 0; // per-pass override lists live in one device buffer
    while (true) {
        std::vector<Pass> passes;
        std::fill(planned.begin(), planned.end(), 0);
        for (int64_t b = 0; b < n_blocks; ++b) {
            const fk_h2h_block &blk = blocks[b];
            if (blk.state[0] >= stop[(size_t)b] || blk.state[1] >= blk.target) continue;
            uint64_t n = std::min<uint64_t>(stop[(size_t)b] - blk.state[0], blk.target - blk.state[1]);
            planned[(size_t)b] = n;
            uint64_t attempt0 = blk.state[0];
            while (n) { // a block may straddle passes
                if (passes.empty() || passes.back().n_games >= max_launch || passes.back().blocks.size() >= MAX_PASS_BLOCKS) passes.emplace_back();
                Pass &p = passes.back();
                const uint64_t take = std::min<uint64_t>(n, max_launch - p.n_games);
                p.blocks.push_back(DevBlock{blk.pair_id, attempt0, blk.order, p.n_games, (uint32_t)b, 0u});
                p.n_games += (uint32_t)take;
                attempt0 += take;
                n -= take;
            }
        }
        if (passes.empty()) break;
        HIPCHK(c, hipMemsetAsync(c->block_out.p, 0, (size_t)n_blocks * 4 * 8, c->stream));

        // preparation of pass `i` into chunk set `si` on stream `st`
        auto prepare = [&](size_t i, int si, hipStream_t st, SeedArgs &sa) -> int {
            const Pass &p = passes[i];
            ChunkSet &cs = c->sets[si];
            const uint32_t nb = (uint32_t)p.blocks.size();
            cs.prepared = false;
            int rc2 = ensure(c, cs.blocks, (size_t)nb * sizeof(DevBlock));
            if (!rc2) rc2 = ensure(c, cs.game_block, (size_t)p.n_games * 4);
            if (!rc2) rc2 = ensure(c, cs.game_row, (size_t)p.n_games * 4);
            if (rc2) return rc2;
            HIPCHK(c, hipMemcpyAsync(cs.blocks.p, p.blocks.data(), (size_t)nb * sizeof(DevBlock), hipMemcpyHostToDevice, st));
            sa = SeedArgs{};
            sa.prefix = seat_prefix;
            sa.gps = 0;
            sa.k = 2;
            sa.n_games = p.n_games;
            sa.blocks = static_cast<const DevBlock *>(cs.blocks.p);
            sa.n_blocks = nb;
            sa.game_block = static_cast<const uint32_t *>(cs.game_block.p);
            sa.game_row = static_cast<const uint32_t *>(cs.game_row.p);
            rc2 = seed_stage(c, si, st, sa, false, true);
            if (rc2) return rc2;
            HIPCHK(c, hipEventRecord(cs.ready, st));
            cs.side = (st == c->prep_stream);
            return FK_OK;
        };

        std::vector<SeedArgs> sas(passes.size());
        std::vector<int> set_of(passes.size());
        set_of[0] = c->cur;
        HIPCHK(c, hipStreamWaitEvent(c->stream, CSET(c).ready, 0)); // an unused side-stream preparation may still own the set
        (void)hipEventRecord(CSET(c).ev[0], c->stream);             // no permutations here: an empty interval
        (void)hipEventRecord(CSET(c).ev[1], c->stream);
        rc = prepare(0, c->cur, c->stream, sas[0]);
        if (rc) return rc;
        for (size_t i = 0; i < passes.size(); ++i) {
            const Pass &p = passes[i];
            c->cur = set_of[i];
            // overrides -> pass-local game ids (not pipelined: one device list)
            std::vector<DevOverride> dov;
            for (int32_t o = 0; o < n_ov; ++o) {
                if (ov[o].root_seed != root_seed) continue;
                if (ov[o].max_rounds > 65535u) return fail(c, FK_ERR_ARG, "override max_rounds must be <= 65535");
                for (size_t q = 0; q < p.blocks.size(); ++q) {
                    const DevBlock &db = p.blocks[q];
                    const uint32_t n_q = (q + 1 < p.blocks.size() ? p.blocks[q + 1].start : p.n_games) - db.start;
                    if (ov[o].a != db.pair || ov[o].k_or_order != db.order) continue;
                    if (ov[o].b < db.attempt0 || ov[o].b >= db.attempt0 + n_q) continue;
                    const uint32_t game = db.start + (uint32_t)(ov[o].b - db.attempt0);
                    bool replaced = false;
                    for (auto &d : dov)
                        if (d.game == game) {
                            d.max_rounds = ov[o].max_rounds;
                            replaced = true;
                        }
                    if (!replaced) dov.push_back(DevOverride{game, ov[o].max_rounds});
                }
            }
            rc = upload_overrides(c, dov);
            if (rc) return rc;
            HIPCHK(c, hipEventRecord(c->main_idle, c->stream)); // everything that used the other chunk set is in front of this point
            if (i + 1 < passes.size()) {
                set_of[i + 1] = c->cur ^ 1;
                if (pipelined) {
                    HIPCHK(c, hipStreamWaitEvent(c->prep_stream, c->main_idle, 0));
                    (void)hipEventRecord(c->sets[c->cur ^ 1].ev[0], c->prep_stream);
                    (void)hipEventRecord(c->sets[c->cur ^ 1].ev[1], c->prep_stream);
                    rc = prepare(i + 1, c->cur ^ 1, c->prep_stream, sas[i + 1]);
                    if (rc) return rc;
                }
            }
            PlayArgs pa{};
            pa.strat = static_cast<const uint2 *>(c->strat.p);
            pa.score_lut = static_cast<const uint32_t *>(c->score_lut.p);
            pa.discard_lut = static_cast<const uint8_t *>(c->discard_lut.p);
            pa.lds_tables = static_cast<const uint8_t *>(c->lds_tables.p);
            pa.game_block = sas[i].game_row; // the kernel wants table rows: seat s of a game plays strategy 2 * row + s
            pa.ov = static_cast<const DevOverride *>(c->ov.p);
            pa.n_ov = (uint32_t)dov.size();
            pa.mode = MODE_BLOCKS;
            pa.n_games = p.n_games;
            pa.gps = 1;
            pa.n_sh = 1;
            pa.k = 2;
            pa.S = (uint32_t)n_blocks * 2u;
            pa.target50 = ceil_div50(target_score);
            pa.beat50 = floor_div50(target_score);
            pa.max_rounds = (uint32_t)max_rounds;
            HIPCHK(c, hipStreamWaitEvent(c->stream, CSET(c).ready, 0));
            rc = launch_play_stage(c, sas[i], pa, plan, false, true, false);
            if (rc) return rc;
            hipLaunchKernelGGL(fk_h2h_reduce_kernel, dim3((p.n_games + 256u * H2H_RUN - 1u) / (256u * H2H_RUN)), dim3(256), 0, c->stream,
                               static_cast<const uint32_t *>(c->rec0.p), p.n_games, (uint32_t)n_blocks, static_cast<unsigned long long *>(c->block_out.p));
            HIPCHK(c, hipGetLastError());
            if (!pipelined || n_ov) { // the next pass re-uses the one override buffer: finish this pass first
                rc = finish_play(c, pa, 0, "h2h attempt (pass-local index)");
                if (rc) return rc;
                if (i + 1 < passes.size()) {
                    (void)hipEventRecord(c->sets[c->cur ^ 1].ev[0], c->stream);
                    (void)hipEventRecord(c->sets[c->cur ^ 1].ev[1], c->stream);
                    rc = prepare(i + 1, c->cur ^ 1, c->stream, sas[i + 1]);
                    if (rc) return rc;
                }
            } else {
                // the pass's error record and timers are read at the end of the generation; its kernel timer events must not
                // be reused by the next launch
                HIPCHK(c, hipMemcpyAsync(c->err_host, pa.err, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                rc = report_device_error(c, c->err_host, 0, "h2h attempt (pass-local index)");
                const int rc_t = finish_timers(c);
                if (rc) return rc;
                if (rc_t) return rc_t;
            }
        }
        HIPCHK(c, hipMemcpyAsync(out.data(), c->block_out.p, (size_t)n_blocks * 4 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t b = 0; b < n_blocks; ++b) {
            const uint64_t n_b = planned[(size_t)b];
            if (!n_b) continue;
            const uint64_t comp = out[(size_t)b * 4], saf = out[(size_t)b * 4 + 1], a1 = out[(size_t)b * 4 + 2], a2 = out[(size_t)b * 4 + 3];
            if (comp + saf != n_b || a1 + a2 != comp)
                return fail(c, FK_ERR_HIP, "h2h block conservation failed (block %lld: %llu + %llu != %llu)", (long long)b,
                            (unsigned long long)comp, (unsigned long long)saf, (unsigned long long)n_b);
            fk_h2h_block &blk = blocks[b];
            blk.state[0] += n_b;
            blk.state[1] += comp;
            blk.state[2] += saf;
            blk.state[3] += a1;
            blk.state[4] += a2;
        }
    }
    c->timing.total_ms = c->timing.seed_ms + c->timing.play_ms;
    return FK_OK;
}

int fk_h2h_run(fk_ctx *c, const fk_strategy seats[2], uint64_t root_seed, uint64_t pair_id, uint32_t order, uint64_t target,
               uint64_t max_attempts, uint64_t chunk_games, int32_t target_score, int32_t max_rounds, const fk_override *ov,
               int32_t n_ov, uint64_t state[5]) {
    if (!c) return FK_ERR_ARG;
    if (!seats || !state) return fail(c, FK_ERR_ARG, "seats and state are required");
    if (order > 1u) return fail(c, FK_ERR_ARG, "order must be 0 or 1");
    fk_h2h_block blk{};
    blk.seats[0] = seats[0];
    blk.seats[1] = seats[1];
    blk.pair_id = pair_id;
    blk.order = order;
    blk.target = target;
    blk.max_attempts = max_attempts;
    memcpy(blk.state, state, sizeof(blk.state));
    const int rc = fk_h2h_run_blocks(c, &blk, 1, root_seed, chunk_games, target_score, max_rounds, ov, n_ov);
    if (rc == FK_OK) memcpy(state, blk.state, sizeof(blk.state));
    return rc;
}

// The NEXT fk_tournament_run / _stats call on this context will play shuffles [shuffle_begin, shuffle_end) of the same table,
// k and root seed as the call that follows this hint (need_state: it will ask for rows or all-seat statistics).  That call
// then enqueues the hinted range's permutations and seat seeding on a low-priority stream behind its own game kernel, where
// they fill the kernel's drain tail; the hinted call finds them done.  Purely a scheduling hint: a wrong one wastes work only.
int fk_tournament_hint_next(fk_ctx *c, uint64_t shuffle_begin, uint64_t shuffle_end, int32_t need_state) {
    if (!c) return FK_ERR_ARG;
    if (shuffle_end < shuffle_begin) return fail(c, FK_ERR_ARG, "bad shuffle range");
    c->hint_valid = shuffle_end > shuffle_begin;
    c->hint_begin = shuffle_begin;
    c->hint_end = shuffle_end;
    c->hint_state = need_state ? 1 : 0;
    return FK_OK;
}

// ---- multi-GPU: one RCCL sum of the tally (the analogue of OutcomeCounter.absorb, run_tournament.py:197-213) ----
int fk_comm_unique_id(fk_comm_id *out) {
    if (!out) return FK_ERR_ARG;
    Rccl &r = rccl();
    if (!r.lib) return FK_ERR_COMM;
    return r.GetUniqueId(out) == 0 ? FK_OK : FK_ERR_COMM;
}

// A blocking initialisation under a deadline, on a helper thread.  Ownership of what `init` made is decided by ONE atomic state
// (0 running, 1 done, 2 abandoned): the helper EXCHANGES in "done" when `init` returns and, if the waiter had already abandoned the call,
// tears its own product down with `orphaned`; the waiter, at the deadline, COMPARE-EXCHANGES running -> abandoned and, if that fails, the
// helper has finished and the product is the waiter's.  Exactly one side ever touches the product (the round-5 handshake of two flags
// let both act when `init` returned at the deadline itself: the waiter installed a communicator the helper was about to abort).
// Returns true when `init` finished in time (rc / product handed over), false when the call was abandoned (the helper stays detached).
extern "C++" {
struct DeadlineCall {
    std::atomic<int> state{0};
    int rc = 0;
    void *product = nullptr;
};
template <class Init, class Orphaned>
static bool init_under_deadline(Init init, Orphaned orphaned, int timeout_ms, int &rc, void *&product, std::atomic<int> *helper_exits = nullptr) {
    auto call = std::make_shared<DeadlineCall>();
    const double t0 = now_ms();
    std::thread([call, init, orphaned, helper_exits]() {
        call->rc = init(&call->product);
        if (call->state.exchange(1) == 2 && call->rc == 0 && call->product) { // abandoned before this returned: built for nobody
            void *mine = call->product;
            call->product = nullptr;
            orphaned(mine);
        }
        if (helper_exits) helper_exits->fetch_add(1);
    }).detach();
    while (call->state.load(std::memory_order_acquire) != 1) {
        if (now_ms() - t0 > (double)timeout_ms) {
            int expected = 0;
            if (call->state.compare_exchange_strong(expected, 2)) return false; // the helper will find "abandoned" and clean up after itself
            break;                                                               // it finished in this very moment: the product is ours
        }
        usleep(200);
    }
    rc = call->rc;
    product = call->product;
    return true;
}
} // extern "C++"

// FK_COMM_TIMEOUT_MS is the default for contexts whose "comm_timeout_ms" option was never set (an explicit fk_set_option wins);
// anything that is not a non-negative integer is ignored rather than silently turned into the blocking path
static int32_t effective_comm_timeout_ms(const fk_ctx *c) {
    if (!c->comm_timeout_set) {
        if (const char *env = getenv("FK_COMM_TIMEOUT_MS")) {
            char *endp = nullptr;
            const long v = strtol(env, &endp, 10);
            if (endp != env && *endp == '\0' && v >= 0 && v <= 86400000L) return (int32_t)v;
        }
    }
    return c->comm_timeout_ms;
}

// The handshake above with a stand-in for ncclCommInitRank that sleeps `init_ms` and hands out a token: out[0] = 1 the caller received
// the product (in time), 0 abandoned; out[1] = how many products were torn down as orphans, counted after the helper has exited.  For
// every timing exactly one of the two happens (tests/test_abi_cpu.py sweeps init_ms across the deadline).
int fk_debug_deadline_handshake(int32_t init_ms, int32_t timeout_ms, int64_t *out) {
    if (!out || init_ms < 0 || timeout_ms < 1) return FK_ERR_ARG;
    auto orphans = std::make_shared<std::atomic<int>>(0);
    std::atomic<int> exits{0};
    int rc = 0;
    void *product = nullptr;
    static int token;
    const bool in_time = init_under_deadline(
        [init_ms](void **made) {
            usleep((useconds_t)init_ms * 1000u);
            *made = &token;
            return 0;
        },
        [orphans](void *) { orphans->fetch_add(1); }, timeout_ms, rc, product, &exits);
    while (!exits.load()) usleep(200); // (a test wants the helper's verdict; fk_comm_init never waits for an abandoned helper)
    out[0] = in_time && product == &token ? 1 : 0;
    out[1] = orphans->load();
    return FK_OK;
}

int fk_comm_init(fk_ctx *c, const fk_comm_id *id, int32_t rank, int32_t world_size) {
    if (!c) return FK_ERR_ARG;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(c, FK_ERR_ARG, "bad communicator id / rank / world size");
    Rccl &r = rccl();
    if (!r.lib) return fail(c, FK_ERR_COMM, "%s", r.why.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) {
        (void)r.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_timeout_ms = effective_comm_timeout_ms(c);
    c->comm_async = false;
    const bool dbg = getenv("FK_DEBUG_COMM") != nullptr;
    if (c->comm_timeout_ms > 0) {
        // ncclCommInitRank blocks until every rank has joined (measured on RCCL 2.27.7: so does ncclCommInitRankConfig with
        // blocking = 0 when it is called outside a group).  A rank that died between the rendezvous and this call must not park its
        // peers here until the job's time limit: the call runs on a helper thread and this thread waits for it under the deadline.
        // On expiry the helper stays parked in RCCL's bootstrap (a socket wait; it owns nothing of this context and is never
        // joined), the context has no communicator, and the caller gets FK_ERR_COMM — bench.py / farkle run then agree on gloo.
        // Should the peer join after all, the helper finds the call abandoned and aborts the communicator it just made itself
        // (ncclCommAbort: nobody else holds it), so a late join leaks nothing — init_under_deadline decides with ONE atomic who owns the
        // result.  A process that has seen this error should end (exit code != 0 is the caller's call): a thread may still sit inside librccl.
        const fk_comm_id id_copy = *id;
        const int device = c->device;
        const double t0 = now_ms();
        Rccl *rp = &r;
        int init_rc = 0;
        void *made = nullptr;
        const bool in_time = init_under_deadline(
            [id_copy, device, world_size, rank, rp](void **out) {
                (void)hipSetDevice(device);
                return rp->CommInitRank(out, world_size, id_copy, rank);
            },
            [rp](void *orphan) { (void)(rp->CommAbort ? rp->CommAbort(orphan) : rp->CommDestroy(orphan)); }, c->comm_timeout_ms, init_rc, made);
        if (!in_time) {
            if (dbg) fprintf(stderr, "[fk comm] ncclCommInitRank still waiting after %.0f ms: giving up\n", now_ms() - t0);
            return fail(c, FK_ERR_COMM, "ncclCommInitRank did not complete within %d ms (comm_timeout_ms): a peer rank never joined",
                        c->comm_timeout_ms);
        }
        struct { int rc; void *comm; } done{init_rc, made}, *pending = &done;
        if (dbg) fprintf(stderr, "[fk comm] ncclCommInitRank returned %d after %.1f ms\n", pending->rc, now_ms() - t0);
        if (pending->rc != 0) return rccl_fail(c, "ncclCommInitRank", pending->rc);
        if (!pending->comm) return fail(c, FK_ERR_COMM, "ncclCommInitRank returned no communicator");
        c->comm = pending->comm;
        c->comm_rank = rank;
        c->comm_world = world_size;
        c->comm_lost = false;
        return FK_OK;
    }
    const int rc = r.CommInitRank(&c->comm, world_size, *id, rank);
    if (rc != 0) {
        c->comm = nullptr;
        return rccl_fail(c, "ncclCommInitRank", rc);
    }
    c->comm_rank = rank;
    c->comm_world = world_size;
    c->comm_lost = false;
    return FK_OK;
}

int fk_reduce_tally(fk_ctx *c, int64_t *tally, int64_t n, int32_t root_rank) {
    if (!c) return FK_ERR_ARG;
    if (!tally || n < 0) return fail(c, FK_ERR_ARG, "tally is required");
    if (c->comm_lost) return fail(c, FK_ERR_COMM, "the communicator was aborted after a failed or timed-out collective: call fk_comm_init again");
    if (!c->comm) return fail(c, FK_ERR_COMM, "no communicator: call fk_comm_init first");
    if (root_rank < 0 || root_rank >= c->comm_world) return fail(c, FK_ERR_ARG, "root rank outside the communicator");
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * sizeof(int64_t);
    int rc = ensure(c, c->comm_buf, bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->comm_buf.p, tally, bytes, hipMemcpyHostToDevice, c->stream));
    const double t0 = now_ms();
    const bool is_root = c->comm_rank == root_rank;
    const int nrc = rccl().Reduce(c->comm_buf.p, c->comm_buf.p, (size_t)n, 4 /* ncclInt64 */, 0 /* ncclSum */, root_rank, c->comm, c->stream);
    if ((rc = comm_settle(c, "ncclReduce", nrc, t0))) return rc;
    if (is_root) HIPCHK(c, hipMemcpyAsync(tally, c->comm_buf.p, bytes, hipMemcpyDeviceToHost, c->stream));
    return comm_wait_stream(c, "ncclReduce", t0);
}

// The resident accumulator summed over the communicator ON THE DEVICE (ncclReduce on the engine's stream) and copied to the
// host on the root rank only; single-rank contexts (no communicator) just copy it.  The accumulator is cleared afterwards.
int fk_tally_resident_reduce(fk_ctx *c, int64_t *out, int64_t n, int32_t root_rank) {
    if (!c) return FK_ERR_ARG;
    if (n < 0 || (size_t)n != c->acc_n || !c->acc.p) return fail(c, FK_ERR_ARG, "resident tally holds %lld elements, %lld asked for", (long long)c->acc_n, (long long)n);
    if (c->comm_lost) return fail(c, FK_ERR_COMM, "the communicator was aborted after a failed or timed-out collective: call fk_comm_init again");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * sizeof(int64_t);
    bool root = true;
    if (c->comm && c->comm_world > 1) {
        if (root_rank < 0 || root_rank >= c->comm_world) return fail(c, FK_ERR_ARG, "root rank outside the communicator");
        const double t0 = now_ms();
        root = c->comm_rank == root_rank;
        const int nrc = rccl().Reduce(c->acc.p, c->acc.p, (size_t)n, 4 /* ncclInt64 */, 0 /* ncclSum */, root_rank, c->comm, c->stream);
        int rc = comm_settle(c, "ncclReduce", nrc, t0);
        if (!rc) rc = comm_wait_stream(c, "ncclReduce", t0);
        if (rc) return rc;
    }
    if (root) {
        if (!out) return fail(c, FK_ERR_ARG, "the root rank needs an output buffer");
        HIPCHK(c, hipMemcpyAsync(out, c->acc.p, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->acc.p, 0, bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

// ranks of the communicator as RCCL itself counts them (ncclCommCount); 1 without a communicator
int fk_comm_ranks(fk_ctx *c) {
    if (!c) return FK_ERR_ARG;
    if (!c->comm) return 1;
    int n = 0;
    auto count = reinterpret_cast<int (*)(void *, int *)>(dlsym(rccl().lib, "ncclCommCount"));
    if (!count || count(c->comm, &n) != 0) return c->comm_world;
    return n;
}

int fk_comm_destroy(fk_ctx *c) {
    if (!c) return FK_ERR_ARG;
    if (c->comm) {
        HIPCHK(c, hipSetDevice(c->device));
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_world = 1;
    c->comm_rank = 0;
    c->comm_async = false;
    c->comm_lost = false;
    return FK_OK;
}

// ---- probes ----
int fk_debug_score(fk_ctx *c, int64_t n, const uint8_t *faces, const int32_t *len, const int32_t *pre,
                   const fk_strategy *strategy, int32_t *out) {
    if (!c || n < 0 || !faces || !len || !pre || !strategy || !out) return c ? fail(c, FK_ERR_ARG, "bad arguments") : FK_ERR_ARG;
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    for (int64_t i = 0; i < n; ++i) {
        if (pre[i] < 0 || pre[i] % 50) return fail(c, FK_ERR_ARG, "turn_score_pre must be a non-negative multiple of 50 (every Farkle score is)");
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
                       static_cast<const uint32_t *>(c->score_lut.p), static_cast<const uint8_t *>(c->discard_lut.p),
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
    for (int64_t i = 0; i < n; ++i)
        if (args[i * 6] < 0 || args[i * 6] % 50 || args[i * 6 + 5] < 0 || args[i * 6 + 5] % 50)
            return fail(c, FK_ERR_ARG, "turn_score and player_score must be non-negative multiples of 50 (every Farkle score is)");
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
    const uint4 *d_seeds = nullptr, *d_incs = nullptr;
    const uint64_t *d_state = nullptr;
    if (coords) {
        if ((rc = ensure(c, c->coords, sizeof(fk_coord) * (size_t)n))) return rc;
        if ((rc = ensure(c, CSET(c).state, (size_t)n * 16))) return rc;
        if ((rc = ensure(c, CSET(c).inc, (size_t)n * 16))) return rc;
        // one 1-seat "game" per coordinate; the seed kernel offsets the seat stream by coords[i].seat_index
        HIPCHK(c, hipMemcpyAsync(c->coords.p, coords, sizeof(fk_coord) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        SeedArgs sa{};
        sa.coords = static_cast<const fk_coord *>(c->coords.p);
        sa.k = 1;
        sa.n_games = (uint32_t)n;
        sa.state = static_cast<uint32_t *>(CSET(c).state.p);
        sa.state_dw = 4;
        sa.inc = static_cast<uint4 *>(CSET(c).inc.p);
        hipLaunchKernelGGL(fk_seed_kernel, dim3((unsigned)((n + SEED_BLOCK - 1) / SEED_BLOCK)), dim3(SEED_BLOCK), 0, c->stream, sa);
        HIPCHK(c, hipGetLastError());
        d_seeds = static_cast<const uint4 *>(CSET(c).state.p);
        d_incs = sa.inc;
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
    hipLaunchKernelGGL(fk_dbg_dice_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, n, d_seeds, d_incs, d_state, n_calls,
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

int fk_game_seeds(fk_ctx *c, uint32_t purpose, uint64_t root_seed, uint64_t k, uint64_t shuffle_begin, uint64_t n_shuffles,
                  uint32_t games_per_shuffle, uint32_t *seed32) {
    if (!c) return FK_ERR_ARG;
    if (!seed32 || games_per_shuffle == 0 || n_shuffles > 0xffffffffull || n_shuffles * games_per_shuffle > ((uint64_t)1 << 33))
        return fail(c, FK_ERR_ARG, "bad arguments");
    if (n_shuffles == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    c->letters.clear();
    c->mail_used = 0;
    const size_t n = (size_t)n_shuffles * games_per_shuffle;
    int rc = ensure(c, c->dbg[5], n * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(fk_game_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, seed_prefix(purpose, root_seed, k),
                       shuffle_begin, (uint32_t)n_shuffles, games_per_shuffle, static_cast<uint32_t *>(c->dbg[5].p));
    HIPCHK(c, hipGetLastError());
    rc = post_d2h(c, seed32, c->dbg[5].p, n * 4);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    deliver_mail(c);
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

int fk_debug_dice_keys(fk_ctx *c, int64_t n, const uint64_t *state, int32_t n_calls, const int32_t *sizes, uint32_t *keys,
                       uint64_t *state_out) {
    if (!c) return FK_ERR_ARG;
    if (n < 0 || n_calls < 1 || !state || !sizes || !keys || !state_out) return fail(c, FK_ERR_ARG, "bad arguments");
    for (int32_t i = 0; i < n_calls; ++i)
        if (sizes[i] < 1 || sizes[i] > 6) return fail(c, FK_ERR_ARG, "roll sizes must be in 1..6");
    if (n == 0) return FK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->dbg[0], (size_t)n * 48))) return rc;
    if ((rc = ensure(c, c->dbg[1], (size_t)n_calls * 4))) return rc;
    if ((rc = ensure(c, c->dbg[2], (size_t)n * n_calls * 4))) return rc;
    if ((rc = ensure(c, c->dbg[4], (size_t)n * 48))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->dbg[0].p, state, (size_t)n * 48, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dbg[1].p, sizes, (size_t)n_calls * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(fk_dbg_dice_key_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, n,
                       static_cast<const uint64_t *>(c->dbg[0].p), n_calls, static_cast<const int32_t *>(c->dbg[1].p),
                       static_cast<uint32_t *>(c->dbg[2].p), static_cast<uint64_t *>(c->dbg[4].p));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(keys, c->dbg[2].p, (size_t)n * n_calls * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(state_out, c->dbg[4].p, (size_t)n * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FK_OK;
}

} // extern "C"
