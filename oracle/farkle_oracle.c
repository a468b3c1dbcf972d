/*
 * farkle_oracle.c — TEST INFRASTRUCTURE ONLY (see farkle_oracle.h).
 *
 * Scalar CPU restatement of the Farkle_II simulation hot path, written to follow the
 * reference's control flow statement by statement (not optimised).  Citations are
 * path:line under /root/reference/.  The RNG layer restates NumPy's published
 * algorithms (numpy>=1.26, pyproject.toml:21; validated against NumPy 2.2.6):
 * SeedSequence (O'Neill seed_seq_fe128), PCG64DXSM, the buffered next_uint32,
 * Lemire bounded integers and the masked-rejection random_interval used by shuffle.
 */
#include "farkle_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* a1: coordinate_entropy / _uint64_words, src/farkle/utils/random.py:58-124 */
/* ------------------------------------------------------------------------- */
void fko_entropy_words(const fko_coord *c, uint32_t w[18]) {
    const uint64_t v[8] = {c->root_seed, c->k,          c->shuffle_index, c->pair_id,
                           c->order,     c->game_index, c->seat_index,    c->replicate_index};
    w[0] = 2u; /* RNG_SCHEME_VERSION, random.py:13,112 */
    w[1] = c->purpose;
    for (int i = 0; i < 8; ++i) {
        w[2 + 2 * i] = (uint32_t)(v[i] & 0xffffffffu); /* low word first, random.py:62 */
        w[3 + 2 * i] = (uint32_t)(v[i] >> 32);
    }
}

/* ------------------------------------------------------------------------- */
/* a2: numpy.random.SeedSequence (called at random.py:156)                    */
/* ------------------------------------------------------------------------- */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u

static uint32_t ss_hashmix(uint32_t value, uint32_t *hash_const) {
    value ^= *hash_const;
    *hash_const *= SS_MULT_A;
    value *= *hash_const;
    value ^= value >> 16;
    return value;
}

static uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> 16;
    return r;
}

void fko_seedseq_pool(const uint32_t *entropy, int n, uint32_t pool[4]) {
    uint32_t hc = SS_INIT_A;
    for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n ? entropy[i] : 0u, &hc);
    for (int src = 0; src < 4; ++src)
        for (int dst = 0; dst < 4; ++dst)
            if (src != dst) pool[dst] = ss_mix(pool[dst], ss_hashmix(pool[src], &hc));
    for (int src = 4; src < n; ++src)
        for (int dst = 0; dst < 4; ++dst) pool[dst] = ss_mix(pool[dst], ss_hashmix(entropy[src], &hc));
}

void fko_seedseq_generate(const uint32_t pool[4], uint32_t *out, int n_words32) {
    uint32_t hc = SS_INIT_B;
    for (int i = 0; i < n_words32; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hc;
        hc *= SS_MULT_B;
        v *= hc;
        v ^= v >> 16;
        out[i] = v;
    }
}

/* coordinate_seed(dtype=uint32 / uint64), random.py:191-225 */
uint32_t fko_coordinate_seed32(const fko_coord *c) {
    uint32_t w[18], pool[4], out[1];
    fko_entropy_words(c, w);
    fko_seedseq_pool(w, 18, pool);
    fko_seedseq_generate(pool, out, 1);
    return out[0];
}

uint64_t fko_coordinate_seed64(const fko_coord *c) {
    uint32_t w[18], pool[4], out[2];
    fko_entropy_words(c, w);
    fko_seedseq_pool(w, 18, pool);
    fko_seedseq_generate(pool, out, 2);
    return (uint64_t)out[0] | ((uint64_t)out[1] << 32);
}

/* ------------------------------------------------------------------------- */
/* a2: numpy.random.PCG64DXSM (constructed at random.py:188)                  */
/* ------------------------------------------------------------------------- */
#define PCG_CHEAP_MULT 0xda942042e4dd58b5ULL
static const u128 PCG_DEFAULT_MULT = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;

void fko_rng_init(fko_rng *r, const fko_coord *c) {
    uint32_t w[18], pool[4], g[8];
    fko_entropy_words(c, w);
    fko_seedseq_pool(w, 18, pool);
    fko_seedseq_generate(pool, g, 8); /* generate_state(4, uint64): low word first */
    uint64_t s[4];
    for (int i = 0; i < 4; ++i) s[i] = (uint64_t)g[2 * i] | ((uint64_t)g[2 * i + 1] << 32);
    u128 initstate = ((u128)s[0] << 64) | s[1];
    u128 initseq = ((u128)s[2] << 64) | s[3];
    u128 inc = (initseq << 1) | 1u;
    /* NumPy seeds PCG64DXSM through the shared pcg_setseq_128_srandom_r, i.e. with the
     * DEFAULT 128-bit multiplier for the two seeding steps (verified against NumPy). */
    u128 state = 0;
    state = state * PCG_DEFAULT_MULT + inc;
    state += initstate;
    state = state * PCG_DEFAULT_MULT + inc;
    r->state_hi = (uint64_t)(state >> 64);
    r->state_lo = (uint64_t)state;
    r->inc_hi = (uint64_t)(inc >> 64);
    r->inc_lo = (uint64_t)inc;
    r->has_uint32 = 0;
    r->uinteger = 0;
}

uint64_t fko_next64(fko_rng *r) {
    uint64_t hi = r->state_hi, lo = r->state_lo;
    /* DXSM output on the pre-advanced state */
    lo |= 1u;
    hi ^= hi >> 32;
    hi *= PCG_CHEAP_MULT;
    hi ^= hi >> 48;
    hi *= lo;
    /* cheap-multiplier LCG step */
    u128 st = ((u128)r->state_hi << 64) | r->state_lo;
    u128 inc = ((u128)r->inc_hi << 64) | r->inc_lo;
    st = st * (u128)PCG_CHEAP_MULT + inc;
    r->state_hi = (uint64_t)(st >> 64);
    r->state_lo = (uint64_t)st;
    return hi;
}

/* buffered 32-bit draw: low half first, high half kept for the next call */
uint32_t fko_next32(fko_rng *r) {
    if (r->has_uint32) {
        r->has_uint32 = 0;
        return r->uinteger;
    }
    uint64_t n = fko_next64(r);
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

/* a3/a13: Generator.integers(lo, hi): buffered_bounded_lemire_uint32 */
int64_t fko_integers(fko_rng *r, int64_t lo, int64_t hi) {
    uint64_t rng = (uint64_t)(hi - lo - 1); /* inclusive range */
    if (rng == 0) return lo;
    if (rng == 0xffffffffULL) return lo + (int64_t)fko_next32(r);
    uint32_t rng_excl = (uint32_t)rng + 1u;
    uint64_t m = (uint64_t)fko_next32(r) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (uint32_t)((0xffffffffu - (uint32_t)rng) % rng_excl);
        while (leftover < threshold) {
            m = (uint64_t)fko_next32(r) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return lo + (int64_t)(m >> 32);
}

/* a10: Generator.permutation(n) = arange + Fisher-Yates with random_interval */
static uint32_t rng_interval(fko_rng *r, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = (fko_next32(r) & mask)) > max) {
    }
    return v;
}

void fko_permutation(fko_rng *r, int32_t n, int32_t *out) {
    for (int32_t i = 0; i < n; ++i) out[i] = i;
    for (int32_t i = n - 1; i >= 1; --i) {
        uint32_t j = rng_interval(r, (uint32_t)i);
        int32_t t = out[i];
        out[i] = out[j];
        out[j] = t;
    }
}

void fko_stream64(const fko_coord *c, int n, uint64_t *out) {
    fko_rng r;
    fko_rng_init(&r, c);
    for (int i = 0; i < n; ++i) out[i] = fko_next64(&r);
}

/* dice from an explicit generator state {state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger} */
void fko_dice_from_state(const uint64_t st[6], int n_calls, const int32_t *sizes, uint8_t *faces, uint64_t out[6]) {
    fko_rng r = {st[0], st[1], st[2], st[3], (uint32_t)st[4], (uint32_t)st[5]};
    for (int i = 0; i < n_calls; ++i)
        for (int j = 0; j < sizes[i]; ++j) *faces++ = (uint8_t)fko_integers(&r, 1, 7);
    out[0] = r.state_hi;
    out[1] = r.state_lo;
    out[2] = r.inc_hi;
    out[3] = r.inc_lo;
    out[4] = r.has_uint32;
    out[5] = r.uinteger;
}

void fko_dice_stream(const fko_coord *c, int n_calls, const int32_t *sizes, uint8_t *faces) {
    fko_rng r;
    fko_rng_init(&r, c);
    for (int i = 0; i < n_calls; ++i)
        for (int j = 0; j < sizes[i]; ++j) *faces++ = (uint8_t)fko_integers(&r, 1, 7);
}

/* ------------------------------------------------------------------------- */
/* a4: _evaluate_nb, src/farkle/game/scoring_lookup.py:27-172                 */
/* ------------------------------------------------------------------------- */
void fko_evaluate(const int32_t counts[6], int32_t *score, int32_t *used, int32_t *sf, int32_t *so) {
    int32_t ctr[6];
    int ones = 0, pairs = 0, trips = 0, has4 = 0, has2 = 0;
    for (int i = 0; i < 6; ++i) {
        ctr[i] = counts[i];
        ones += (ctr[i] == 1);
        pairs += (ctr[i] == 2);
        trips += (ctr[i] == 3);
        has4 |= (ctr[i] == 4);
        has2 |= (ctr[i] == 2);
    }
    *sf = 0;
    *so = 0;
    if (ones == 6) { *score = 1500; *used = 6; return; }   /* _straight :28-38 */
    if (pairs == 3) { *score = 1500; *used = 6; return; }  /* _three_pairs :42-53 */
    if (trips == 2) { *score = 2500; *used = 6; return; }  /* _two_triplets :57-68 */
    if (has4 && has2) { *score = 1500; *used = 6; return; } /* _four_kind_plus_pair :72-82 */
    int32_t sc = 0, us = 0;
    for (int face = 0; face < 6; ++face) { /* _apply_sets :86-115 */
        int32_t n = ctr[face];
        if (n >= 3) {
            int32_t pts;
            if (n == 3) pts = (face == 0) ? 300 : (face + 1) * 100;
            else if (n == 4) pts = 1000;
            else if (n == 5) pts = 2000;
            else pts = 3000;
            sc += pts;
            us += n;
            ctr[face] = 0;
        }
    }
    int32_t lone_ones = ctr[0], lone_fives = ctr[4]; /* :168-172 */
    sc += lone_ones * 100 + lone_fives * 50;
    us += lone_ones + lone_fives;
    *score = sc;
    *used = us;
    *sf = lone_fives;
    *so = lone_ones;
}

/* a5: _must_bank, src/farkle/game/scoring.py:283-300 */
static int must_bank(int32_t score_after, int32_t dice_left_after, const fko_strategy *s) {
    int hit_score = s->consider_score ? (score_after >= s->score_threshold) : 0;
    int hit_dice = s->consider_dice ? (dice_left_after <= s->dice_threshold) : 0;
    if (s->consider_score && s->consider_dice && s->require_both) return hit_score && hit_dice;
    return hit_score || hit_dice;
}

/* a5: _decide_smart_discards_impl / generate_sequences / score_lister / _select_candidate,
 * src/farkle/game/scoring.py:197-467 — literal enumeration with re-scoring. */
static void decide_smart_discards(const int32_t counts[6], int32_t single_fives, int32_t single_ones,
                                  int32_t raw_used, int32_t len, int32_t turn_score_pre,
                                  const fko_strategy *s, int32_t *d5, int32_t *d1) {
    *d5 = 0;
    *d1 = 0;
    if (!s->smart_five || raw_used == len || (single_fives == 0 && single_ones == 0)) return; /* :433 */
    int have_best = 0;
    int32_t best_a = 0, best_b = 0, best_sf = single_fives, best_so = single_ones;
    int32_t max_fives = counts[4];
    int32_t max_ones = s->smart_one ? counts[0] : 0; /* :219-222 */
    for (int32_t drop5 = 0; drop5 <= max_fives; ++drop5) {
        for (int32_t drop1 = 0; drop1 <= max_ones; ++drop1) {
            int32_t nc[6];
            memcpy(nc, counts, sizeof(nc));
            nc[4] -= drop5;
            nc[0] -= drop1;
            int32_t cs, cu, csf, cso;
            fko_evaluate(nc, &cs, &cu, &csf, &cso);
            if (cs == 0) continue;                                        /* score_lister :262 */
            if (drop5 > single_fives || drop1 > single_ones) continue;    /* :326-329 */
            int32_t score_after = turn_score_pre + cs;                    /* :331 */
            int32_t dice_left_after = len - cu;                           /* :334 */
            if (must_bank(score_after, dice_left_after, s)) continue;     /* :337 */
            int32_t ka = s->favor_score ? score_after : dice_left_after;  /* :346-353 */
            int32_t kb = s->favor_score ? dice_left_after : score_after;
            if (!have_best || ka > best_a || (ka == best_a && kb > best_b)) { /* tuple '>' :354 */
                have_best = 1;
                best_a = ka;
                best_b = kb;
                best_sf = csf;
                best_so = cso;
            }
        }
    }
    if (!have_best) return; /* :460 */
    *d5 = single_fives - best_sf; /* :467 */
    *d1 = single_ones - best_so;
}

/* a6: default_score, src/farkle/game/scoring.py:618-693 */
void fko_default_score(const uint8_t *faces, int32_t n, int32_t turn_score_pre, const fko_strategy *s,
                       int32_t out[5]) {
    int32_t counts[6] = {0, 0, 0, 0, 0, 0};
    for (int32_t i = 0; i < n; ++i) counts[faces[i] - 1]++; /* _faces_to_counts_nb :48-64 */
    int32_t raw_score, raw_used, sf, so, d5, d1;
    fko_evaluate(counts, &raw_score, &raw_used, &sf, &so);
    decide_smart_discards(counts, sf, so, raw_used, n, turn_score_pre, s, &d5, &d1);
    int32_t final_score = raw_score - 50 * d5 - 100 * d1; /* apply_discards :575-578 */
    int32_t final_used = raw_used - d5 - d1;
    out[0] = final_score;
    out[1] = final_used;
    out[2] = n - final_used;
    out[3] = d5;
    out[4] = d1;
}

/* a7: _decide_continue + ThresholdStrategy.decide, src/farkle/simulation/strategies.py:125-162, 212-275 */
int32_t fko_decide(const fko_strategy *s, int32_t turn_score, int32_t dice_left, int32_t has_scored,
                   int32_t final_round, int32_t score_to_beat, int32_t running_total) {
    if (!has_scored && turn_score < 500) return 1; /* :249 */
    if (final_round) {                             /* :253-260 */
        if (running_total <= score_to_beat) return 1;
        if (!s->run_up_score) return 0;
    }
    int want_s = s->consider_score && turn_score < s->score_threshold; /* :154 */
    int want_d = s->consider_dice && dice_left > s->dice_threshold;    /* :155 */
    if (s->consider_score && s->consider_dice) return s->require_both ? (want_s || want_d) : (want_s && want_d);
    if (s->consider_score) return want_s;
    if (s->consider_dice) return want_d;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a7/a8: FarklePlayer / FarkleGame, src/farkle/game/engine.py                */
/* ------------------------------------------------------------------------- */
typedef struct {
    const fko_strategy *strategy;
    fko_rng rng;
    int32_t score, has_scored;
    int32_t n_turns, n_farkles, n_rolls, highest_turn;
    int32_t smart_five_uses, n_smart_five_dice, smart_one_uses, n_smart_one_dice, n_hot_dice;
} player_t;

typedef struct {
    const uint8_t *faces; /* scripted dice (NULL = use player RNGs) */
    int32_t n_faces, pos;
} script_t;

#define ROLL_LIMIT 1000 /* engine.py:36 */

static int roll_dice(player_t *p, script_t *sc, int32_t n, uint8_t *faces) { /* _roll :85-101 */
    p->n_rolls += 1;
    for (int32_t i = 0; i < n; ++i) {
        if (sc && sc->faces) {
            if (sc->pos >= sc->n_faces) return FKO_ERR_ARG;
            faces[i] = sc->faces[sc->pos++];
        } else {
            faces[i] = (uint8_t)fko_integers(&p->rng, 1, 7);
        }
    }
    return FKO_OK;
}

static int should_continue(const player_t *p, int32_t turn_score, int32_t dice_left, int32_t final_round,
                           int32_t score_to_beat) { /* _should_continue :156-205 */
    int32_t running_total = p->score + turn_score;
    if (final_round && running_total > score_to_beat && !p->strategy->run_up_score) return 0; /* :189 */
    int keep = fko_decide(p->strategy, turn_score, dice_left, p->has_scored, final_round, score_to_beat,
                          running_total);
    if (final_round && running_total <= score_to_beat) keep = 1; /* :202 */
    return keep;
}

int32_t fko_should_continue(const fko_strategy *s, int32_t turn_score, int32_t dice_left, int32_t has_scored,
                            int32_t final_round, int32_t score_to_beat, int32_t player_score) {
    player_t p;
    memset(&p, 0, sizeof(p));
    p.strategy = s;
    p.score = player_score;
    p.has_scored = has_scored;
    return should_continue(&p, turn_score, dice_left, final_round, score_to_beat);
}

static int take_turn(player_t *p, script_t *sc, int32_t final_round, int32_t score_to_beat) { /* :208-273 */
    p->n_turns += 1;
    int32_t dice = 6, turn_score = 0, rolls_this_turn = 0;
    while (dice > 0) {
        if (rolls_this_turn >= ROLL_LIMIT) return FKO_ERR_ROLL_LIMIT; /* :242 */
        uint8_t faces[6];
        int32_t n = dice;
        int rc = roll_dice(p, sc, n, faces);
        if (rc) return rc;
        rolls_this_turn += 1;
        int32_t r[5];
        fko_default_score(faces, n, turn_score, p->strategy, r); /* _score_roll :103-147 */
        int32_t pts = r[0], used = r[1], reroll = r[2], d5 = r[3], d1 = r[4];
        if (pts == 0) {
            p->n_farkles += 1;
            turn_score = 0; /* :247-249 */
            break;
        }
        if (d5 > 0) { p->smart_five_uses += 1; p->n_smart_five_dice += d5; }
        if (d1 > 0) { p->smart_one_uses += 1; p->n_smart_one_dice += d1; }
        dice = (used == n && reroll == 0) ? 6 : reroll; /* :146 */
        turn_score += pts;
        if (p->strategy->auto_hot_dice && dice == 6) { /* _apply_hot_dice :149-154 */
            p->n_hot_dice += 1;
            continue;
        }
        if (!should_continue(p, turn_score, dice, final_round, score_to_beat)) break;
    }
    if (!p->has_scored && turn_score >= 500) p->has_scored = 1; /* :267 */
    if (p->has_scored) {                                        /* :271-273 */
        p->score += turn_score;
        if (turn_score > p->highest_turn) p->highest_turn = turn_score;
    }
    return FKO_OK;
}

static int u16_ok(int32_t v) { return v >= 0 && v <= 0xffff; }

static int play_players(player_t *pl, script_t *sc, const int32_t *seat_strategy, int32_t k,
                        int32_t target_score, int32_t max_rounds, void *row) { /* FarkleGame.play :436-521 */
    int final_round = 0;
    int32_t score_to_beat = target_score;
    int32_t rounds = 0;
    while (rounds < max_rounds) {
        rounds += 1;
        for (int32_t i = 0; i < k; ++i) {
            int rc = take_turn(&pl[i], sc, final_round, score_to_beat);
            if (rc) return rc;
            if (!final_round && pl[i].score >= target_score) { /* :462-468 */
                final_round = 1;
                score_to_beat = pl[i].score;
                for (int32_t j = 0; j < k; ++j) { /* _run_final_round :523-550 */
                    if (j == i) continue;
                    rc = take_turn(&pl[j], sc, 1, score_to_beat);
                    if (rc) return rc;
                    if (pl[j].score > score_to_beat) score_to_beat = pl[j].score;
                }
                break;
            }
        }
        if (final_round) break;
    }
    int max_rounds_hit = (!final_round) && rounds >= max_rounds; /* :472 */

    fko_row_hdr *hdr = (fko_row_hdr *)row;
    fko_seat *seats = (fko_seat *)((char *)row + sizeof(fko_row_hdr));
    if (rounds > 0xffff) return FKO_ERR_COUNTER_OVERFLOW;
    hdr->n_rounds = (uint16_t)rounds;
    hdr->status = max_rounds_hit ? FKO_SAFETY_LIMIT : FKO_COMPLETED;
    hdr->winner_seat = -1;
    for (int32_t i = 0; i < k; ++i) {
        const player_t *p = &pl[i];
        if (!(u16_ok(p->n_farkles) && u16_ok(p->n_rolls) && u16_ok(p->n_turns) && u16_ok(p->highest_turn) &&
              u16_ok(p->smart_five_uses) && u16_ok(p->n_smart_five_dice) && u16_ok(p->smart_one_uses) &&
              u16_ok(p->n_smart_one_dice) && u16_ok(p->n_hot_dice)))
            return FKO_ERR_COUNTER_OVERFLOW;
        fko_seat *s = &seats[i];
        s->score = p->score;
        s->strategy = seat_strategy[i];
        s->farkles = (uint16_t)p->n_farkles;
        s->rolls = (uint16_t)p->n_rolls;
        s->n_turns = (uint16_t)p->n_turns;
        s->highest_turn = (uint16_t)p->highest_turn;
        s->smart_five_uses = (uint16_t)p->smart_five_uses;
        s->n_smart_five_dice = (uint16_t)p->n_smart_five_dice;
        s->smart_one_uses = (uint16_t)p->smart_one_uses;
        s->n_smart_one_dice = (uint16_t)p->n_smart_one_dice;
        s->hot_dice = (uint16_t)p->n_hot_dice;
        s->hit_max_rounds = (uint8_t)max_rounds_hit;
        s->rank = 0;
    }
    if (!max_rounds_hit) {
        /* stable sort on score descending: rank = 1 + #{j: score_j > score_i or (== and j < i)}, :477-483 */
        for (int32_t i = 0; i < k; ++i) {
            int32_t rank = 1;
            for (int32_t j = 0; j < k; ++j)
                if (pl[j].score > pl[i].score || (pl[j].score == pl[i].score && j < i)) rank++;
            seats[i].rank = (uint8_t)rank;
            if (rank == 1) hdr->winner_seat = (int8_t)i;
        }
    }
    return FKO_OK;
}

#define FKO_MAX_K 64

int fko_play_game(const fko_coord *gc, const fko_strategy *table, const int32_t *seat_strategy, int32_t k,
                  int32_t target_score, int32_t max_rounds, void *row) { /* _play_game simulation.py:576-655 */
    if (k < 1 || k > FKO_MAX_K) return FKO_ERR_ARG;
    player_t pl[FKO_MAX_K];
    memset(pl, 0, sizeof(player_t) * (size_t)k);
    for (int32_t i = 0; i < k; ++i) { /* _make_players simulation.py:412-447 */
        fko_coord c = *gc;
        c.seat_index = (uint64_t)i;
        pl[i].strategy = &table[seat_strategy[i]];
        fko_rng_init(&pl[i].rng, &c);
    }
    return play_players(pl, NULL, seat_strategy, k, target_score, max_rounds, row);
}

int fko_play_game_scripted(const uint8_t *faces, int32_t n_faces, const fko_strategy *table,
                           const int32_t *seat_strategy, int32_t k, int32_t target_score, int32_t max_rounds,
                           void *row) {
    if (k < 1 || k > FKO_MAX_K) return FKO_ERR_ARG;
    player_t pl[FKO_MAX_K];
    memset(pl, 0, sizeof(player_t) * (size_t)k);
    for (int32_t i = 0; i < k; ++i) pl[i].strategy = &table[seat_strategy[i]];
    script_t sc = {faces, n_faces, 0};
    return play_players(pl, &sc, seat_strategy, k, target_score, max_rounds, row);
}

int fko_play_games(const fko_coord *coords, int64_t n_games, const fko_strategy *table,
                   const int32_t *seat_strategy, int32_t k, int32_t target_score, int32_t max_rounds,
                   void *rows, int32_t n_threads) {
    size_t row_bytes = sizeof(fko_row_hdr) + sizeof(fko_seat) * (size_t)k;
    int err = FKO_OK;
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (int64_t g = 0; g < n_games; ++g) {
        int rc = fko_play_game(&coords[g], table, seat_strategy + g * k, k, target_score, max_rounds,
                               (char *)rows + (size_t)g * row_bytes);
        if (rc) {
#pragma omp critical
            err = rc;
        }
    }
    return err;
}

/* ------------------------------------------------------------------------- */
/* a10/a11: _play_one_shuffle + tally, src/farkle/simulation/run_tournament.py:301-393 */
/* ------------------------------------------------------------------------- */
static uint32_t lookup_override(const fko_override *ov, int32_t n_ov, uint64_t root, uint32_t k_or_order,
                                uint64_t a, uint64_t b, uint32_t dflt) {
    for (int32_t i = 0; i < n_ov; ++i)
        if (ov[i].root_seed == root && ov[i].k_or_order == k_or_order && ov[i].a == a && ov[i].b == b)
            return ov[i].max_rounds;
    return dflt;
}

static void tally_row(int64_t *t /* [S][26] */, const void *row, int32_t k) {
    const fko_row_hdr *hdr = (const fko_row_hdr *)row;
    const fko_seat *seats = (const fko_seat *)((const char *)row + sizeof(fko_row_hdr));
    for (int32_t i = 0; i < k; ++i) { /* OutcomeCounter.record_row :177-195 */
        int64_t *r = t + (size_t)seats[i].strategy * FKO_TALLY_COLS;
        r[1] += 1;
        if (hdr->status == FKO_COMPLETED) r[2] += 1;
        else r[3] += 1;
    }
    if (hdr->status != FKO_COMPLETED) return; /* :376-379 */
    const fko_seat *w = &seats[hdr->winner_seat];
    int64_t *r = t + (size_t)w->strategy * FKO_TALLY_COLS;
    r[0] += 1; /* wins :386 */
    const int64_t m[11] = {w->score,           hdr->n_rounds,         w->farkles,        w->rolls,
                           w->highest_turn,    w->smart_five_uses,    w->n_smart_five_dice,
                           w->smart_one_uses,  w->n_smart_one_dice,   w->hot_dice,       w->hit_max_rounds};
    for (int j = 0; j < 11; ++j) { /* METRIC_LABELS :109-121, :387-389 */
        r[4 + j] += m[j];
        r[15 + j] += m[j] * m[j];
    }
}

int fko_tournament(const fko_strategy *table, int32_t S, int32_t k, uint64_t root_seed, uint64_t shuffle_begin,
                   uint64_t shuffle_end, uint32_t shuffles_per_batch, int32_t target_score, int32_t max_rounds,
                   const fko_override *ov, int32_t n_ov, int64_t *tally, void *rows, int32_t *perms,
                   uint32_t *game_seeds, int32_t n_threads) {
    if (k < 1 || k > FKO_MAX_K || S < k || S % k != 0 || shuffles_per_batch == 0) return FKO_ERR_ARG; /* :274 */
    int64_t n_sh = (int64_t)(shuffle_end - shuffle_begin);
    int32_t gps = S / k; /* games_per_shuffle :92-94 */
    size_t row_bytes = sizeof(fko_row_hdr) + sizeof(fko_seat) * (size_t)k;
    int64_t n_batches = (n_sh + shuffles_per_batch - 1) / shuffles_per_batch;
    memset(tally, 0, sizeof(int64_t) * (size_t)n_batches * (size_t)S * FKO_TALLY_COLS);
    int err = FKO_OK;
    (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
        char *rowbuf = (char *)malloc(row_bytes);
        int64_t *local = (int64_t *)calloc((size_t)S * FKO_TALLY_COLS, sizeof(int64_t));
        int64_t local_batch = -1;
        /* one shuffle per work item; a thread sees shuffles (hence batches) in increasing order and
         * folds its private tally into the batch's slot whenever the batch changes (integer sums:
         * order independent) */
#pragma omp for schedule(dynamic, 1) nowait
        for (int64_t si = 0; si < n_sh; ++si) {
            const int64_t b = si / shuffles_per_batch;
            if (b != local_batch) {
                if (local_batch >= 0) {
#pragma omp critical(fko_tally)
                    for (size_t i = 0; i < (size_t)S * FKO_TALLY_COLS; ++i)
                        tally[(size_t)local_batch * S * FKO_TALLY_COLS + i] += local[i];
                }
                memset(local, 0, sizeof(int64_t) * (size_t)S * FKO_TALLY_COLS);
                local_batch = b;
            }
            uint64_t shuffle = shuffle_begin + (uint64_t)si;
            fko_coord pc = {FKO_NS_SHUFFLE_PERMUTATION, 0, root_seed, (uint64_t)k, shuffle, 0, 0, 0, 0, 0};
            fko_rng prng;
            fko_rng_init(&prng, &pc);        /* :312-317 */
            fko_permutation(&prng, S, perm); /* :318 */
            if (perms) memcpy(perms + (size_t)si * S, perm, sizeof(int32_t) * (size_t)S);
            for (int32_t g = 0; g < gps; ++g) {
                fko_coord gc = {FKO_NS_TOURNAMENT_PLAYER, 0, root_seed, (uint64_t)k, shuffle, 0, 0,
                                (uint64_t)g,              0, 0}; /* :366-372 */
                if (game_seeds) { /* :319-329 */
                    fko_coord fc = gc;
                    fc.purpose = FKO_NS_TOURNAMENT_GAME;
                    game_seeds[(size_t)si * gps + g] = fko_coordinate_seed32(&fc);
                }
                uint32_t mr = lookup_override(ov, n_ov, root_seed, (uint32_t)k, shuffle, (uint64_t)g,
                                              (uint32_t)max_rounds);
                char *row = rows ? (char *)rows + ((size_t)si * gps + g) * row_bytes : rowbuf;
                int rc = fko_play_game(&gc, table, perm + (size_t)g * k, k, target_score, (int32_t)mr, row);
                if (rc) {
#pragma omp critical(fko_err)
                    err = rc;
                    continue;
                }
                tally_row(local, row, k);
            }
        }
        if (local_batch >= 0) {
#pragma omp critical(fko_tally)
            for (size_t i = 0; i < (size_t)S * FKO_TALLY_COLS; ++i)
                tally[(size_t)local_batch * S * FKO_TALLY_COLS + i] += local[i];
        }
        free(perm);
        free(rowbuf);
        free(local);
    }
    return err;
}

/* ------------------------------------------------------------------------- */
/* a14: _simulate_block_from_manifest, src/farkle/analysis/h2h_schedule.py:1149-1243 */
/* ------------------------------------------------------------------------- */
int fko_h2h_block(const fko_strategy seats[2], uint64_t root_seed, uint64_t pair_id, uint32_t order,
                  uint64_t target, uint64_t max_attempts, uint64_t chunk_games, int32_t target_score,
                  int32_t max_rounds, const fko_override *ov, int32_t n_ov, uint64_t state[5]) {
    uint64_t attempted = state[0], completed = state[1], safety = state[2], w1 = state[3], w2 = state[4];
    uint64_t stop = attempted + chunk_games;
    if (stop > max_attempts) stop = max_attempts; /* :1172 */
    const int32_t seat_strategy[2] = {0, 1};
    char row[sizeof(fko_row_hdr) + 2 * sizeof(fko_seat)];
    for (uint64_t attempt = state[0]; attempt < stop; ++attempt) {
        if (completed >= target) break; /* :1174 */
        fko_coord gc = {FKO_NS_H2H_PLAYER, 0, root_seed, 2, 0, pair_id, order, attempt, 0, 0}; /* :1199-1206 */
        uint32_t mr = lookup_override(ov, n_ov, root_seed, order, pair_id, attempt, (uint32_t)max_rounds);
        int rc = fko_play_game(&gc, seats, seat_strategy, 2, target_score, (int32_t)mr, row);
        if (rc) return rc;
        const fko_row_hdr *hdr = (const fko_row_hdr *)row;
        attempted += 1; /* :1220-1235 */
        if (hdr->status == FKO_COMPLETED) {
            completed += 1;
            if (hdr->winner_seat == 0) w1 += 1;
            else w2 += 1;
        } else {
            safety += 1;
        }
    }
    state[0] = attempted;
    state[1] = completed;
    state[2] = safety;
    state[3] = w1;
    state[4] = w2;
    return FKO_OK;
}

/* ------------------------------------------------------------------------- */
/* a13: random_threshold_strategy, src/farkle/simulation/strategies.py:399-452 */
/* make_random_strategies, src/farkle/simulation/time_farkle.py:23-46          */
/* ------------------------------------------------------------------------- */
void fko_random_strategy(uint64_t seed, uint64_t k, uint64_t seat_index, fko_strategy *out) {
    fko_coord c = {FKO_NS_STRATEGY, 0, seed, k, 0, 0, 0, 0, seat_index, 0};
    fko_rng r;
    fko_rng_init(&r, &c);
    int sf = (int)fko_integers(&r, 0, 2);
    int so = sf ? (int)fko_integers(&r, 0, 2) : 0;
    int cs = (int)fko_integers(&r, 0, 2);
    int cd = (int)fko_integers(&r, 0, 2);
    int rb = (cs && cd) ? (int)fko_integers(&r, 0, 2) : 0;
    int favor_score;
    if (cs == cd) favor_score = (fko_integers(&r, 0, 2) == 0); /* _sample_favor_score :413-415 */
    else favor_score = cs;
    out->score_threshold = (int32_t)fko_integers(&r, 1, 20) * 50;
    out->dice_threshold = (int32_t)fko_integers(&r, 0, 5);
    out->smart_five = (uint8_t)sf;
    out->smart_one = (uint8_t)so;
    out->consider_score = (uint8_t)cs;
    out->consider_dice = (uint8_t)cd;
    out->require_both = (uint8_t)rb;
    out->auto_hot_dice = 0;
    out->run_up_score = 0;
    out->favor_score = (uint8_t)favor_score;
    out->strategy_id = -1;
}
