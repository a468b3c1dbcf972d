/*
 * farkle_hip.h — C-ABI of the MI355X (gfx950) Farkle simulation engine.
 *
 * The reference (Isaac-McPadden/Farkle_II) is pure Python and has no FFI; its seams for
 * the simulation hot path are in-process callables.  Each entry point below replaces one
 * of those seams (citations are path:line under the reference's root):
 *
 *   fk_tournament_run  <-  _play_one_shuffle / _run_chunk / _run_chunk_metrics
 *                          src/farkle/simulation/run_tournament.py:301-393, 403-457, 473-585
 *   fk_play_games      <-  _play_game over an explicit coordinate list
 *                          src/farkle/simulation/simulation.py:576-655 (callers: simulate_many_games
 *                          :658-722, _measure_throughput run_tournament.py:593-619)
 *   fk_h2h_run         <-  _simulate_block_from_manifest attempt loop (the BlockRunner contract)
 *                          src/farkle/analysis/h2h_schedule.py:1149-1243, 1521
 *   fk_coordinate_seeds <- coordinate_seed (src/farkle/utils/random.py:190-232)
 *   fk_debug_*         <-  FarklePlayer._roll (src/farkle/game/engine.py:85-101) and
 *                          default_score / decide (src/farkle/game/scoring.py:618-693,
 *                          src/farkle/simulation/strategies.py:212-275): single-op probes of the
 *                          SAME device functions the game kernel uses, for parity tests.
 *
 * Conventions: plain C types, caller-allocated caller-owned HOST buffers, no callbacks.
 * Every function returns FK_OK (0) or a negative error code; fk_last_error() gives the
 * message (for FK_ERR_ROLL_LIMIT it names the offending game, mirroring the RuntimeError of
 * engine.py:242-243).  One context per process/GPU; calls on a context are serialised by the
 * caller.  There is NO CPU fallback: fk_init fails if no HIP device is usable.
 */
#ifndef FARKLE_HIP_H
#define FARKLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FK_OK = 0,
    FK_ERR_ROLL_LIMIT = -1,       /* a turn exceeded 1000 rolls (engine.py:36,242) */
    FK_ERR_ARG = -2,              /* invalid argument */
    FK_ERR_COUNTER_OVERFLOW = -3, /* a per-seat u16 counter left its guarded range */
    FK_ERR_HIP = -4,              /* HIP runtime failure */
    FK_ERR_NO_DEVICE = -5,
    FK_ERR_COMM = -6,             /* RCCL not loadable / communicator failure */
    FK_ERR_IO = -7                /* a row shard could not be written (fk_write_row_shards) */
};

enum { FK_COMPLETED = 0, FK_SAFETY_LIMIT = 1 };

/* ThresholdStrategy (src/farkle/simulation/strategies.py:165-194), 20 bytes */
typedef struct {
    int32_t score_threshold;
    int32_t dice_threshold;
    uint8_t smart_five, smart_one, consider_score, consider_dice;
    uint8_t require_both, auto_hot_dice, run_up_score, favor_score; /* favor_score=1: FavorDiceOrScore.SCORE */
    int32_t strategy_id; /* carried, not interpreted */
} fk_strategy;

/* Per-seat part of a game row (PlayerStats, src/farkle/game/engine.py:360-406), 28 bytes */
typedef struct {
    int32_t score;
    int32_t strategy; /* INDEX into the strategy table of the call */
    uint16_t farkles, rolls, n_turns, highest_turn;
    uint16_t smart_five_uses, n_smart_five_dice, smart_one_uses, n_smart_one_dice, hot_dice;
    uint8_t rank;           /* 1-based; 0 = null (safety limit) */
    uint8_t hit_max_rounds; /* 0/1 */
} fk_seat;

/* Row = fk_row_hdr followed by k fk_seat records: 4 + 28*k bytes (simulation.py:628-655) */
typedef struct {
    uint16_t n_rounds;
    uint8_t status;     /* FK_COMPLETED / FK_SAFETY_LIMIT */
    int8_t winner_seat; /* 0-based; -1 = none */
} fk_row_hdr;

/* max_rounds override (src/farkle/simulation/game_profile.py:24-69).
 * Tournament: a=shuffle_index, b=game_index, k_or_order=k.  H2H: a=pair_id, b=attempt_index, k_or_order=order. */
typedef struct {
    uint64_t root_seed;
    uint64_t a, b;
    uint32_t k_or_order;
    uint32_t max_rounds;
} fk_override;

/* Seat-stream coordinate of one game (src/farkle/utils/random.py:80-124); seat_index is implied */
typedef struct {
    uint32_t purpose;
    uint32_t pad;
    uint64_t root_seed, k, shuffle_index, pair_id, order, game_index, seat_index, replicate_index;
} fk_coord;

#define FK_SEAT_STAT_COLS 31 /* all-seat integer statistics per strategy, see fk_tournament_run_stats */
#define FK_LAG_COLS 11 /* lag sufficient statistics per (strategy, lag), see fk_tournament_run_lags */
#define FK_MAX_LAGS 16
#define FK_TALLY_COLS 26 /* wins, attempted, completed, safety, 11 metric sums, 11 square sums (run_tournament.py:109-121) */

typedef struct {
    char name[64];
    char arch[32];
    int32_t compute_units;
    int32_t clock_mhz;
    int32_t wavefront_size;
    int32_t lds_bytes_per_cu;
    uint64_t hbm_bytes;
} fk_device_info;

/* Timing of the last fk_tournament_run / fk_play_games / fk_h2h_run on this context, from HIP events
 * recorded on the context's stream. */
typedef struct {
    float perm_ms;    /* shuffle-permutation kernel(s) of the chunks prepared in front of their game kernel */
    float seed_ms;    /* SeedSequence -> PCG64DXSM seeding kernel(s) of the same chunks */
    float play_ms;    /* game kernel(s) */
    float total_ms;   /* first launch -> last kernel done (device time incl. memsets) */
    int32_t play_launches;
    int32_t play_block, play_grid, play_lds_bytes;
    int64_t games;
    int32_t prefetched_chunks; /* chunks whose permutations / seeding ran on the side stream behind an earlier game kernel:
                                  not in perm_ms / seed_ms */
    int32_t play_clock_mhz;    /* option "clock_stamps": shader clock of the call's last game kernel, measured inside it (median over
                                  its workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz); 0 = not measured */
    float play_block_end_p50_ms, play_block_end_max_ms; /* same option, same kernel: when the median and the last of its workgroups
                                  finished, from the first workgroup's start (100 MHz counter) — max - p50 is the launch's tail, the
                                  stretch in which the chip drains behind the longest games */
    int32_t play_mixed_flags;  /* the strategy-flag form of the last game kernel instance: 0 (every flag shared by the whole table: scalar),
                                  0xc000 (only require_both / favor_score differ) or 0xff00 (any flag may differ) */
} fk_timing;

typedef struct fk_ctx fk_ctx;

int fk_init(int device_ordinal, fk_ctx **out);
void fk_destroy(fk_ctx *ctx);
const char *fk_last_error(fk_ctx *ctx);
int fk_get_device_info(fk_ctx *ctx, fk_device_info *out);
int fk_get_timing(fk_ctx *ctx, fk_timing *out);
/* Page-locked host memory for the caller-owned output buffers (rows above all: 60 bytes per k=2 game).  A buffer from here is
 * copied to by one DMA per chunk at PCIe rate, while the next chunk plays; any other host pointer works too (staged by the HIP
 * runtime, about a third of the rate, the host thread waits).  Free with fk_host_free before fk_destroy.  The pages are anonymous
 * memory populated by the kernel and then registered with the HIP runtime (256 MB: ~15 + ~2 ms, and only the registration takes the
 * runtime's lock — hipHostMalloc, the fallback, holds it for ~45 ms); may be called from any thread. */
int fk_host_alloc(fk_ctx *ctx, size_t bytes, void **out);
int fk_host_free(fk_ctx *ctx, void *p);
/* Tunables: "chunk_bytes" (device workspace budget per chunk), "batch_threshold" (lanes that must be waiting
 * before a wave runs its game hand-over; 0, the default = auto: 8 up to eight seats, 12 at nine / ten, 16 at eleven / twelve), "use_lds_tally" (0/1/-1 auto), "block" (0 auto), "lean" (seat-record layout:
 * -1 auto, 0 full, 1 lean), "state_store" (-1 auto: seat records live in the HBM state store, with only the turn owner's staged in LDS, when
 * k of them do not fit LDS (k > 64); 0 the same; 1 always), "blocks_per_cu", "max_waves" (resident waves per SIMD the launch plan counts
 * on, default 6), "longest_first" (1 = deal
 * never-banking pairings first), "uniform_flags" (-1 auto: tables whose strategies share all flag bits run the scalar-flag
 * kernel instance, 0 never), "perm_split" (-1 auto), "columns_by_seat" (-1 auto: the column images of fk_tournament_run_columns by one thread per (game, seat) up to sixteen seats, per game beyond; 0 / 1 force), "perm_draw_wave" (-1 auto: a shuffle's Fisher-Yates draws by a whole wave in chunks of up to 32 768 shuffles, by one thread beyond; 0 / 1 force), "pipeline" (1, default: the next chunk / hinted call is prepared around the
 * current game kernel — permutations in front of it, schedule and seat seeding on a low-priority stream in its drain tail;
 * 0: every chunk is prepared on the main stream in front of its own game kernel), "hot_cold" (tournament launches of 4..12 seats on
 * the hot / cold game kernel, csrc/fk_play_hc.h: -1 auto, 0 never — the LDS-record kernel plays them), "comm_timeout_ms" (deadline of
 * fk_comm_init and of each collective, default 120 000 or the environment's FK_COMM_TIMEOUT_MS — a non-negative integer, consulted only
 * while this option has never been set; 0 = no deadline.  After a timed-out fk_comm_init the context keeps playing, a peer that joins
 * late meets a helper that aborts the orphan communicator itself, and the process should end when its work is done: a helper thread may
 * still sit inside librccl), "rows_chunk_games" (rows mode plays in chunks of about this many games,
 * default 4 000 000: chunk i's rows cross PCIe while chunk i + 1 plays), "resident_tally" (see fk_tally_resident_reduce), "clock_stamps" (1: every workgroup of a game kernel reads the
 * shader-clock and the 100 MHz reference counters at its first and last instruction; fk_timing.play_clock_mhz).  All of them
 * are scheduling / layout choices: results are identical for every setting. */
int fk_set_option(fk_ctx *ctx, const char *name, int64_t value);
/* Read back an option or a figure of the last call: "chunk_bytes", "workspace_percent" (a buffer set's workspace is at most this share of
 * the device memory the context can have — free now + its own per-chunk buffers — halved for the two sets; default 80), "last_budget" (the
 * bytes per buffer set the last tournament / H2H call planned with), "oom_replays" (how often that call met hipErrorOutOfMemory, gave
 * its workspace back and was replayed with half the budget: results are identical), "comm_timeout_ms" (the EFFECTIVE deadline: the option
 * if it was ever set, else a well-formed FK_COMM_TIMEOUT_MS, else 120 000), "rows_chunk_games". */
int fk_get_option(fk_ctx *ctx, const char *name, int64_t *value);

/* Tournament shuffles [shuffle_begin, shuffle_end) of the S-strategy table at k players.
 *   tally       int64 [n_batches][S][26], n_batches = ceil(n_shuffles / shuffles_per_batch); overwritten.
 *   rows        nullable; n_shuffles * (S/k) rows of 4+28k bytes, game-major in (shuffle, game) order.
 *   perms       nullable; int32 [n_shuffles][S] (the permutation of each shuffle; tests/diagnostics).
 * Requires S % k == 0 (run_tournament.py:274), S <= 65535, max_rounds <= 65535.  Targets above 3 200 000 points play with
 * full LDS records (lean ones carry the banked total / 50 in 16 bits): FK_ERR_ARG if k of those do not fit LDS, and for the
 * batched head-to-head entry points. */
int fk_tournament_run(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                      uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch,
                      int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov,
                      int64_t *tally, void *rows, int32_t *perms);

/* ---- row shards without a host-side conversion -------------------------------------------------------------------------------
 * The reference persists rows as ONE PARQUET FILE PER SHUFFLE (`rows_<root>_<k>p_<shuffle:012d>.parquet`, run_tournament.py:530-558,
 * schema raw_simulation_schema_for(k), utils/schema_helpers.py:23-90).  fk_tournament_run_columns is fk_tournament_run with the rows
 * delivered as per-shuffle COLUMN IMAGES: every column of that schema that depends on the games, already in its Parquet physical type
 * (strategy ids resolved on the device through `strategy_ids`, ranks / margins / loss margins computed there).  Image of one shuffle,
 * fk_row_columns_bytes(k, gps) bytes (a multiple of 64), gps = S / k:
 *     int32 planes [4 + 13 k][gps]   winner_strategy, winning_score, victory_margin, n_rounds, then for seat 1..k: score, farkles, rolls,
 *                                    highest_turn, strategy, rank, loss_margin, smart_five_uses, n_smart_five_dice, smart_one_uses,
 *                                    n_smart_one_dice, hot_dice, n_turns                                   (the schema's column order)
 *     uint8 status[gps]              1 = safety limit: the row's nullable fields are null and every hit flag is set (engine.py:485-489)
 *     uint8 winner_seat[gps]         0-based; uint8 rank_order[gps][k]: seats in rank order (the seat_ranks list)
 *   columns     n_shuffles images, shuffle-major; k <= 64.     strategy_ids   int32 [S]: the strategy_id of table row i. */
int fk_tournament_run_columns(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                              uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch,
                              int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov,
                              int64_t *tally, const int32_t *strategy_ids, void *columns);
/* fk_tournament_run_columns that also delivers what a row shard carries besides the images: shuffle_seeds[n_shuffles] = the ns-100
 * fingerprint of each shuffle (the shards' manifest records, run_tournament.py:97-105) and game_seeds[n_shuffles][S / k] = the ns-102
 * fingerprint of each game (the game_seed column, run_tournament.py:340-350), as fk_game_seeds computes them; either may be null.
 * Complete on return (not subject to "rows_async"). */
int fk_tournament_run_columns_seeds(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                                    uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                                    int32_t max_rounds, const fk_override *overrides, int32_t n_overrides, int64_t *tally,
                                    const int32_t *strategy_ids, void *columns, uint32_t *shuffle_seeds, uint32_t *game_seeds);
size_t fk_row_columns_bytes(int32_t k, int32_t games_per_shuffle);
/* Option "rows_async" = 1: a call that delivers rows (AoS or column images) returns when its last device-to-host copy is QUEUED; the
 * buffer may be read after fk_rows_wait(ctx, slot) with slot = fk_get_option("rows_event") read right after that call (a ring of 4:
 * at most four calls' images may be outstanding).
 * The next call's game kernel then runs beside the copy (a caller with two destination buffers: `farkle run`, rows mode). */
int fk_rows_wait(fk_ctx *ctx, int32_t slot);

/* Column images -> the row-shard files, on `threads` host threads (no ctx: plain host code).  Each file is a complete Parquet file
 * (one row group, uncompressed PLAIN / RLE_DICTIONARY pages, no statistics) whose Arrow schema — restored by a reader from the
 * ARROW:schema entry — and rows equal what the reference writes for the same games.  The FileMetaData parts that only depend on the
 * schema come from the footer of a file Arrow wrote for raw_simulation_schema_for(k) (farkle_ii_amd/parquet_template.py): `footer_head`
 * = its fields 1-2 (version, schema), `footer_kv` = field 5 (key-value metadata), `footer_orders` = field 7 (column orders), as raw Thrift
 * spans; `leaf_type[i]` / `leaf_paths` (components '\0'-separated, each path '\0'-terminated) = physical type and path_in_schema of leaf i.
 * Outputs per shuffle: the file's size and SHA-256 (the manifest record's byte_length / data_sha256) and — with a contract-v3 sidecar
 * template (contract_v3.SimulationContract.shard_template: the text around size, digest and name) — the written sidecar's SHA-256.
 * `atomic`: write `<name>.tmp` and rename.  Returns FK_OK, FK_ERR_ARG or FK_ERR_IO with a message in `error`. */
typedef struct {
    int32_t k, games_per_shuffle, n_shuffles, threads, atomic, rng_purpose_namespace;
    uint64_t root_seed;
    const int64_t *shuffle_index;   /* [n_shuffles] */
    const int64_t *shuffle_seed;    /* [n_shuffles] namespace-100 fingerprints */
    const int32_t *batch_id;        /* [n_shuffles] deterministic_batch_id */
    const uint32_t *game_seed;      /* [n_shuffles][gps] namespace-102 fingerprints (fk_game_seeds) */
    const void *columns;            /* [n_shuffles] images, `shard_stride` bytes apart */
    size_t shard_stride;
    const char *directory;
    const uint8_t *footer_head; size_t footer_head_len;
    const uint8_t *footer_kv; size_t footer_kv_len;
    const uint8_t *footer_orders; size_t footer_orders_len;
    const int32_t *leaf_type;       /* [n_leaves], n_leaves = 18 + 14 k */
    const char *leaf_paths;
    int32_t n_leaves;
    const char *const *side_body;   /* nullable: 4 pieces */
    const char *const *side_full;   /* nullable: 5 pieces */
    const char *side_directory;     /* results-root-relative directory of the shards, with a trailing '/' */
} fk_shard_job;
int fk_write_row_shards(const fk_shard_job *job, int64_t *byte_length, uint8_t *sha256, uint8_t *sidecar_sha256, char *error, size_t error_len);
/* SHA-256 as the shard writer computes it (SHA-NI when the CPU has it; portable = 1 forces the scalar rounds; portable = 2: the two-message
 * lockstep form — digests of data[0 .. n/2) and data[n/2 .. n) into out32[0 .. 32) and out32[32 .. 64)): parity probe. */
int fk_debug_sha256(const void *data, size_t n, uint8_t *out32, int32_t portable);

/* fk_tournament_run plus the integer sufficient statistics of ALL seats (not winners only), per batch and strategy:
 *   seat_stats  nullable; int64 [n_batches][S][FK_SEAT_STAT_COLS]; overwritten.  Columns:
 *     0 exposures, 1 completed exposures, 2 safety-limit exposures (= max-round aborts), 3 wins,
 *     4/5 sum / sum of squares of the final score, 6/7 of n_turns, 8 exposures with n_turns != n_rounds,
 *     9/10 sum / sum of squares of (n_turns - n_rounds), then (sum, sum of squares) pairs of: rank, loss_margin (both over
 *     completed exposures only), rolls, farkles, highest_turn, hot_dice, smart_five_uses, n_smart_five_dice,
 *     smart_one_uses, n_smart_one_dice.
 * These are the integer accumulators of the reference's unconditional all-player batch metrics
 * (src/farkle/analysis/all_player_metrics.py:257-340), produced on the device from the state store without
 * materialising rows; its two ratio statistics (score / n_turns, score / n_rounds) are float64 sums in row order:
 * fk_tournament_run_all_player below. */
int fk_tournament_run_stats(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                            uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch,
                            int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov,
                            int64_t *tally, void *rows, int32_t *perms, int64_t *seat_stats);

/* fk_tournament_run_stats plus the four float64 accumulators of the same table (all_player_metrics.py:308-321), per batch and strategy:
 *   seat_ratio_sums  double [n_batches][S][FK_SEAT_RATIO_COLS]; overwritten.  Columns: sum of score / n_turns over the strategy's
 *     exposures (0 where n_turns is 0), sum of its squares, sum of score / n_rounds, sum of its squares.
 * The reference adds them with np.add.at in source-row order (:174-177), i.e. one sequential float64 sum per strategy over its
 * exposures in (shuffle, game, seat) order.  A strategy sits once per shuffle: the device runs exactly that sequence — one thread
 * per (batch, strategy), ascending shuffles, one rounding per division, multiplication and addition — so the sums carry the bits the
 * reference's do whenever its curated rows are in (shuffle, game) order (pinned by tests/golden/all_player_vectors.json, the reference's
 * own _iter_batch_tables).  seat_stats is required (the digests both read are built once). */
#define FK_SEAT_RATIO_COLS 4
int fk_tournament_run_all_player(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                                 uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch,
                                 int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov,
                                 int64_t *tally, void *rows, int32_t *perms, int64_t *seat_stats, double *seat_ratio_sums);

/* Scheduling hint: the next fk_tournament_run / fk_tournament_run_stats call on this context — the one AFTER the call that
 * follows this hint — will play shuffles [shuffle_begin, shuffle_end) of the same table, k and root seed (need_state != 0: it
 * will ask for rows or seat_stats).  The call that follows the hint prepares that range around its own game kernel
 * (permutations in front of it, schedule and seat seeding on a low-priority stream in its drain tail; option "pipeline" = 0
 * ignores hints).  Results never depend on hints; a hint that turns out wrong only wastes the preparation.  (The reference's process pool keeps `window = 4 * n_jobs` chunks in flight
 * for the same reason, run_tournament.py:1576-1586.) */
int fk_tournament_hint_next(fk_ctx *ctx, uint64_t shuffle_begin, uint64_t shuffle_end, int32_t need_state);

/* fk_tournament_run + the lag sufficient statistics of the reference's RNG diagnostics for the STRATEGY family
 * (analysis/rng_diagnostics.py: observation records :1870-1905, _OnlineMetric :2031-2076, stats rows :2110-2160).  A strategy is
 * seated exactly once per shuffle, so its observation series — win indicator and n_rounds of the game it sat in, ordered by
 * (root_seed, shuffle_index) — is indexed by the shuffle; for every lag the pairs (earlier x, later y) = (series[t - lag], series[t]).
 *   lags        n_lags (1 .. FK_MAX_LAGS) strictly increasing positive shuffle distances (analysis.rng_diagnostic_lags)
 *   lag_sums    int64 [S][n_lags][FK_LAG_COLS], overwritten: 0 pair_count; win indicator: 1 sum x, 2 sum y, 3 sum x^2, 4 sum y^2,
 *               5 sum xy; n_rounds: 6 sum x, 7 sum y, 8 sum x^2, 9 sum y^2, 10 sum xy — over the pairs whose BOTH shuffles lie in
 *               [shuffle_begin, shuffle_end).  Exact integers (the reference accumulates the same integers in float64).
 *   edge_head / edge_tail   uint16 [m][S], m = min(max lag, n_shuffles): the series values (n_rounds | won << 15) of the first / last m
 *               shuffles of the range, strategy-minor.  Two ranges that follow each other (launch groups, ranks: contiguous whole
 *               batches per rank) combine on the host: sums add, plus the pairs that straddle the cut, which need exactly the
 *               tail of the earlier and the head of the later range (farkle_ii_amd/rng_lags.py: LagSummary.merge).
 * The matchup family of the same module (one group per seat TUPLE, O(games) groups) has no pre-aggregation and keeps reading rows.
 * Requires max_rounds (and every override) <= 32767. */
int fk_tournament_run_lags(fk_ctx *ctx, const fk_strategy *strategies, int32_t S, int32_t k, uint64_t root_seed,
                           uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score,
                           int32_t max_rounds, const fk_override *ov, int32_t n_ov, int64_t *tally, const int32_t *lags, int32_t n_lags,
                           int64_t *lag_sums, uint16_t *edge_head, uint16_t *edge_tail);

/* Explicit game list: game g seats strategies table[seat_strategy[g*k+i]] with streams coords[g](seat i).
 * rows: n_games * (4+28k) bytes (required). */
int fk_play_games(fk_ctx *ctx, const fk_coord *coords, int64_t n_games, const fk_strategy *table, int32_t S,
                  const int32_t *seat_strategy, int32_t k, int32_t target_score, int32_t max_rounds, void *rows);

/* H2H block: attempts [state[0], min(max_attempts, state[0]+chunk_games)) of (root, pair, order) in
 * attempt order until `target` completed games; state = {attempted, completed, safety, wins_seat1,
 * wins_seat2} in/out (h2h_schedule.py:1165-1235). */
int fk_h2h_run(fk_ctx *ctx, const fk_strategy seats[2], uint64_t root_seed, uint64_t pair_id, uint32_t order,
               uint64_t target, uint64_t max_attempts, uint64_t chunk_games, int32_t target_score,
               int32_t max_rounds, const fk_override *ov, int32_t n_ov, uint64_t state[5]);

/* One (pair, order) block of a batched H2H call: the two seated strategies, the block's coordinates and limits, and its
 * progress {attempted, completed, safety, wins_seat1, wins_seat2} in/out (h2h_schedule.py:1088-1146, 1165-1235). */
typedef struct {
    fk_strategy seats[2];
    uint64_t pair_id;
    uint32_t order;
    uint32_t pad;
    uint64_t target, max_attempts;
    uint64_t state[5];
} fk_h2h_block;

/* Many H2H blocks of one root advanced together, each by at most chunk_games attempts, each exactly as fk_h2h_run would
 * advance it alone (same in-order prefix rule per block); all blocks share the kernel launches (the production H2H
 * schedule is tens of thousands of blocks of ~2 000 games: execute_h2h_schedule's block loop, h2h_schedule.py:2038-2093). */
int fk_h2h_run_blocks(fk_ctx *ctx, fk_h2h_block *blocks, int64_t n_blocks, uint64_t root_seed, uint64_t chunk_games,
                      int32_t target_score, int32_t max_rounds, const fk_override *ov, int32_t n_ov);

/* ---- multi-GPU: one process per GPU, one context per process, ONE exchange ----
 * The (seed x shuffle x game) space partitions with no data dependency; the only collective of the path is the integer
 * SUM of the per-strategy tally to one rank — the analogue of OutcomeCounter.absorb + _reduce_metric_chunk_payloads
 * (src/farkle/simulation/run_tournament.py:197-213, 1023-1042).  It runs as ncclReduce(sum, int64) on the context's
 * stream over RCCL/xGMI; librccl is loaded on first use.
 *   rank 0:     fk_comm_unique_id(&id); ship the 128 bytes to every rank (any channel: env, file, socket)
 *   every rank: fk_comm_init(ctx, &id, rank, world)           (collective: all ranks call it)
 *               fk_reduce_tally(ctx, tally, n, root)          (collective; `tally` is overwritten with the sum on root) */
typedef struct {
    char bytes[128];
} fk_comm_id;
int fk_comm_unique_id(fk_comm_id *out);
int fk_comm_init(fk_ctx *ctx, const fk_comm_id *id, int32_t rank, int32_t world_size);
int fk_reduce_tally(fk_ctx *ctx, int64_t *tally, int64_t n, int32_t root_rank);
/* Device-resident form of the same reduction.  With fk_set_option(ctx, "resident_tally", 1) every fk_tournament_run* call also
 * adds its [n_batches][S][26] tally to an accumulator in HBM (calls of one shape; a shape change starts a new accumulator).
 * fk_tally_resident_reduce sums the accumulators of all ranks with ncclReduce on the engine's stream — the tally does not leave
 * the device before it is reduced — copies the total to `out` on `root_rank` only (out may be NULL elsewhere) and clears the
 * accumulator.  Without a communicator (one rank) it is the plain copy.  n = elements the accumulator holds.
 * Replaces: OutcomeCounter.absorb over worker results (run_tournament.py:197-213). */
int fk_tally_resident_reduce(fk_ctx *ctx, int64_t *out, int64_t n, int32_t root_rank);
/* Ranks of the context's communicator as RCCL counts them (ncclCommCount); 1 when there is none. */
int fk_comm_ranks(fk_ctx *ctx);
int fk_comm_destroy(fk_ctx *ctx);

/* SeedSequence fingerprints of n coordinates (all nine coordinate words of the record, seat_index included):
 * seed32[i] = generate_state(1, uint32)[0], seed64[i] = generate_state(1, uint64)[0]; either may be NULL.
 * Replaces coordinate_seed (src/farkle/utils/random.py:190-232): the shuffle_seed (namespace 100) and game_seed
 * (namespace 102) columns of rows and manifests (run_tournament.py:318-351), the per-game seed of
 * simulate_many_games (namespace 1, simulation.py:700-713). */
int fk_coordinate_seeds(fk_ctx *ctx, int64_t n, const fk_coord *coords, uint32_t *seed32, uint64_t *seed64);
/* The same uint32 fingerprints for the games of a shuffle range without a coordinate list: seed32[(s - shuffle_begin) * gps + g] =
 * coordinate_seed(purpose, root_seed, k, shuffle_index = s, game_index = g) — the game_seed column of the row contract
 * (run_tournament.py:340-350; purpose 102 for tournaments). */
int fk_game_seeds(fk_ctx *ctx, uint32_t purpose, uint64_t root_seed, uint64_t k, uint64_t shuffle_begin, uint64_t n_shuffles,
                  uint32_t games_per_shuffle, uint32_t *seed32);

/* Hold device memory on purpose until about leave_free bytes are free (leave_free < 0: give everything back); free_now / total as
 * hipMemGetInfo reports them afterwards (either may be NULL).  Tests of the workspace budget and of the out-of-memory replay. */
int fk_debug_hold_memory(fk_ctx *ctx, int64_t leave_free, int64_t *free_now, int64_t *total);

/* The deadline handshake of fk_comm_init with a stand-in for ncclCommInitRank that takes init_ms: out[0] = 1 when the caller received the
 * communicator in time, 0 when it gave up at timeout_ms; out[1] = communicators torn down as orphans by the helper thread.  Host code. */
int fk_debug_deadline_handshake(int32_t init_ms, int32_t timeout_ms, int64_t *out);

/* ---- single-op probes of the device functions (parity tests) ---- */
/* n rolls: roll i scores faces[i*6 .. i*6+len[i]) for strategy[i] with turn_score_pre[i];
 * out[i*5..] = score, used, reroll, d5, d1 (default_score(return_discards=True)). */
int fk_debug_score(fk_ctx *ctx, int64_t n, const uint8_t *faces, const int32_t *len, const int32_t *turn_score_pre,
                   const fk_strategy *strategy, int32_t *out);
/* n decisions: args[i*6..] = turn_score, dice_left, has_scored, final_round, score_to_beat, player_score
 * -> out[i] = FarklePlayer._should_continue (engine.py:156-205) */
int fk_debug_should_continue(fk_ctx *ctx, int64_t n, const int32_t *args, const fk_strategy *strategy, int32_t *out);
/* n streams: coordinate -> PCG64DXSM state; then n_calls dice rolls of sizes[c] each (same sizes for all
 * streams); faces out: n * sum(sizes) bytes; raw64 (nullable): first 4 raw outputs of each stream. */
int fk_debug_dice(fk_ctx *ctx, int64_t n, const fk_coord *coords, int32_t n_calls, const int32_t *sizes,
                  uint8_t *faces, uint64_t *raw64);
/* Same but from explicit generator states: state[i*6..] = state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger. */
int fk_debug_dice_state(fk_ctx *ctx, int64_t n, const uint64_t *state, int32_t n_calls, const int32_t *sizes,
                        uint8_t *faces, uint64_t *state_out);
/* The game kernels' own dice path from explicit generator states: keys[i * n_calls + c] = the 18-bit key of call c (six 3-bit
 * face counts, face 1 in the low bits — what the score table is indexed by), produced by the roll_counts<3> instantiation of
 * fk_play_kernel / fk_play_hc_kernel incl. its Lemire-rejection replay (engine.py:101 -> Generator.integers). */
int fk_debug_dice_keys(fk_ctx *ctx, int64_t n, const uint64_t *state, int32_t n_calls, const int32_t *sizes, uint32_t *keys,
                       uint64_t *state_out);

#ifdef __cplusplus
}
#endif
#endif
