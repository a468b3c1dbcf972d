/*
 * farkle_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the Farkle_II simulation hot path.  It is the parity
 * checker for the HIP path; nothing in the product (farkle_ii_amd/, the C-ABI
 * library) links, imports or calls it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.
 *
 * Parity is PINNED: the restatement is checked (tests/test_oracle_golden.py) against
 *   - the reference's own goldens EXPECTED_ROWS (tests/integration/test_raw_simulation_oracle.py:45-58),
 *     EXPECTED_H2H_BLOCKS (tests/helpers/tournament_analysis_oracle.py:65-78), the
 *     deterministic-counts KAT (tests/unit/simulation/test_simulation.py:184-199), the
 *     SeedSequence KAT (tests/unit/utils/test_random_utils.py:32-39) and the scoring CSV
 *     (tests/data/test_farkle_scores_data.csv), and
 *   - vectors produced by importing the Python reference in the build container
 *     (oracle/gen_golden.py -> tests/golden/).
 *
 * All citations are path:line under /root/reference/.
 */
#ifndef FARKLE_ORACLE_H
#define FARKLE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RandomPurpose namespaces, src/farkle/utils/random.py:18-37 */
enum {
    FKO_NS_INDEXED_SEED = 1,
    FKO_NS_PLAYER = 10,
    FKO_NS_STRATEGY = 11,
    FKO_NS_TOURNAMENT_SHUFFLE = 100,
    FKO_NS_SHUFFLE_PERMUTATION = 101,
    FKO_NS_TOURNAMENT_GAME = 102,
    FKO_NS_TOURNAMENT_PLAYER = 103,
    FKO_NS_H2H_GAME = 202,
    FKO_NS_H2H_PLAYER = 203
};

enum { FKO_OK = 0, FKO_ERR_ROLL_LIMIT = -1, FKO_ERR_ARG = -2, FKO_ERR_COUNTER_OVERFLOW = -3 };
enum { FKO_COMPLETED = 0, FKO_SAFETY_LIMIT = 1 };

/* ThresholdStrategy, src/farkle/simulation/strategies.py:165-194 */
typedef struct {
    int32_t score_threshold;
    int32_t dice_threshold;
    uint8_t smart_five, smart_one, consider_score, consider_dice;
    uint8_t require_both, auto_hot_dice, run_up_score, favor_score; /* favor_score=1 <=> FavorDiceOrScore.SCORE */
    int32_t strategy_id;
} fko_strategy; /* 20 bytes */

/* Semantic coordinate of one seat stream / fingerprint, src/farkle/utils/random.py:80-124 */
typedef struct {
    uint32_t purpose;
    uint32_t pad;
    uint64_t root_seed, k, shuffle_index, pair_id, order, game_index, seat_index, replicate_index;
} fko_coord;

typedef struct {
    int32_t score;
    int32_t strategy; /* index into the strategy table of the call */
    uint16_t farkles, rolls, n_turns, highest_turn;
    uint16_t smart_five_uses, n_smart_five_dice, smart_one_uses, n_smart_one_dice, hot_dice;
    uint8_t rank;           /* 1-based, 0 = null (safety limit) */
    uint8_t hit_max_rounds; /* 0/1 */
} fko_seat; /* 28 bytes */

typedef struct {
    uint16_t n_rounds;
    uint8_t status;     /* FKO_COMPLETED / FKO_SAFETY_LIMIT */
    int8_t winner_seat; /* 0-based, -1 = none */
} fko_row_hdr;          /* 4 bytes; followed by k fko_seat records */

/* max_rounds override on a tournament (a=shuffle_index,b=game_index) or H2H
 * (a=pair_id, b=attempt_index, order) coordinate; src/farkle/simulation/game_profile.py:24-69 */
typedef struct {
    uint64_t root_seed;
    uint64_t a, b;
    uint32_t k_or_order;
    uint32_t max_rounds;
} fko_override;

#define FKO_TALLY_COLS 26 /* wins, attempted, completed, safety, 11 sums, 11 square sums */

/* ---- RNG primitives (NumPy SeedSequence / PCG64DXSM / Generator) ---- */
typedef struct {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    uint32_t has_uint32, uinteger;
} fko_rng;

void fko_entropy_words(const fko_coord *c, uint32_t words[18]);
void fko_seedseq_pool(const uint32_t *entropy, int n, uint32_t pool[4]);
void fko_seedseq_generate(const uint32_t pool[4], uint32_t *out, int n_words32);
uint32_t fko_coordinate_seed32(const fko_coord *c);
uint64_t fko_coordinate_seed64(const fko_coord *c);
void fko_rng_init(fko_rng *r, const fko_coord *c);
uint64_t fko_next64(fko_rng *r);
uint32_t fko_next32(fko_rng *r);
/* Generator.integers(lo, hi) (hi exclusive, hi-lo <= 2^32) */
int64_t fko_integers(fko_rng *r, int64_t lo, int64_t hi);
/* Generator.permutation(n) into out[n] */
void fko_permutation(fko_rng *r, int32_t n, int32_t *out);
/* draw `n` raw uint64 from a coordinate's stream (test helper) */
void fko_stream64(const fko_coord *c, int n, uint64_t *out);
/* Generator.integers(1,7,size=sizes[i]) repeated; writes all faces consecutively */
void fko_dice_stream(const fko_coord *c, int n_calls, const int32_t *sizes, uint8_t *faces);

void fko_dice_from_state(const uint64_t st[6], int n_calls, const int32_t *sizes, uint8_t *faces, uint64_t out[6]);

/* ---- scoring ---- */
/* counts[6] -> score, used, single_fives, single_ones; src/farkle/game/scoring_lookup.py:123-172 */
void fko_evaluate(const int32_t counts[6], int32_t *score, int32_t *used, int32_t *sf, int32_t *so);
/* default_score(return_discards=True); src/farkle/game/scoring.py:618-693 */
void fko_default_score(const uint8_t *faces, int32_t n, int32_t turn_score_pre, const fko_strategy *s,
                       int32_t out[5] /* score, used, reroll, d5, d1 */);
/* ThresholdStrategy.decide; src/farkle/simulation/strategies.py:212-275 */
int32_t fko_decide(const fko_strategy *s, int32_t turn_score, int32_t dice_left, int32_t has_scored,
                   int32_t final_round, int32_t score_to_beat, int32_t running_total);

/* FarklePlayer._should_continue; src/farkle/game/engine.py:156-205 */
int32_t fko_should_continue(const fko_strategy *s, int32_t turn_score, int32_t dice_left, int32_t has_scored,
                            int32_t final_round, int32_t score_to_beat, int32_t player_score);

/* ---- game / tournament / h2h ---- */
/* One game: seat i uses strategy table[seat_strategy[i]] and stream coord(seat_index=i).
 * `row` receives 4 + 28*k bytes.  Returns FKO_OK or an error code. */
int fko_play_game(const fko_coord *game_coord, const fko_strategy *table, const int32_t *seat_strategy,
                  int32_t k, int32_t target_score, int32_t max_rounds, void *row);

/* Scripted-dice variant used to replay the reference's engine unit tests
 * (tests/unit/game/test_engine.py): faces are consumed in order instead of RNG draws. */
int fko_play_game_scripted(const uint8_t *faces, int32_t n_faces, const fko_strategy *table,
                           const int32_t *seat_strategy, int32_t k, int32_t target_score,
                           int32_t max_rounds, void *row);

/* Explicit game list (farkle time / simulate_many_games path): coords[g] gives everything
 * but seat_index; seat_strategy[g*k + i]. rows = n_games * (4+28k) bytes. */
int fko_play_games(const fko_coord *coords, int64_t n_games, const fko_strategy *table,
                   const int32_t *seat_strategy, int32_t k, int32_t target_score, int32_t max_rounds,
                   void *rows, int32_t n_threads);

/* _play_one_shuffle over [shuffle_begin, shuffle_end); src/farkle/simulation/run_tournament.py:301-393.
 * tally: int64[n_batches][S][26], batch = (shuffle - shuffle_begin) / shuffles_per_batch.
 * rows (nullable): [(shuffle_end-shuffle_begin) * (S/k)] records of 4+28k bytes.
 * perms (nullable): int32[(n_shuffles)][S]. game_seeds (nullable): uint32 per game (ns=102). */
int fko_tournament(const fko_strategy *table, int32_t S, int32_t k, uint64_t root_seed,
                   uint64_t shuffle_begin, uint64_t shuffle_end, uint32_t shuffles_per_batch,
                   int32_t target_score, int32_t max_rounds, const fko_override *ov, int32_t n_ov,
                   int64_t *tally, void *rows, int32_t *perms, uint32_t *game_seeds, int32_t n_threads);

/* _simulate_block_from_manifest attempt loop; src/farkle/analysis/h2h_schedule.py:1149-1243.
 * state = {attempted, completed, safety, wins_seat1, wins_seat2} in/out. */
int fko_h2h_block(const fko_strategy seats[2], uint64_t root_seed, uint64_t pair_id, uint32_t order,
                  uint64_t target, uint64_t max_attempts, uint64_t chunk_games, int32_t target_score,
                  int32_t max_rounds, const fko_override *ov, int32_t n_ov, uint64_t state[5]);

/* random_threshold_strategy(rng) with rng = coordinate_rng(STRATEGY, root_seed=seed, k=k, seat_index=i);
 * src/farkle/simulation/strategies.py:418-452, src/farkle/simulation/time_farkle.py:23-46 */
void fko_random_strategy(uint64_t seed, uint64_t k, uint64_t seat_index, fko_strategy *out);

#ifdef __cplusplus
}
#endif
#endif
