// fk_play_hc.h — the hot / cold game kernel: tournament launches of four and more seats, whose k seat records do not leave
// LDS room for more than two or three waves per SIMD (included by fk_kernels.h; same rules, same turn registers, same
// hand-over as fk_play_kernel).
//
// fk_play_kernel keeps the ten dwords a turn mutates of EVERY seat in LDS: 40 k bytes per lane, i.e. 4 waves per SIMD at
// k = 4, 2.5 at k = 6 and 2 at k = 8 — and a wave alone issues one vector instruction per 4 cycles where the SIMD could
// take one per 2 (MI355X_MICROARCH.md): PMC puts the k = 8 launch at 6.2 cycles per vector instruction against 3.3 at six
// waves.  Round 2 measured the alternative that keeps ONE record per lane in LDS and the rest in HBM (state-store
// instances): 2x slower.  Round 3 found out why, and what the budget really is (profiles/r03_*):
//   * the texture addresser takes one cycle per lane of a scattered access: fk_play_kernel keeps it 49 % busy (two table
//     gathers per roll, two loads per turn), six more 16-byte accesses per turn saturate it;
//   * a CU has 128 KB of L2 to itself (4 MB per XCD / 32 CUs) — LESS than its LDS.  The increment plane's live lines are
//     16 k bytes per resident game; anything spilled beside them overflows that share from three waves per SIMD on (first
//     version of this kernel at k = 8: 0.8 L2 misses per turn, 105 GB fetched + 107 GB written back per 1.5 x 10^7 games);
//   * what such a launch has to spare is REGISTERS: at three waves per SIMD a lane may hold 168 VGPRs, the game needs ~85.
// So the seat record is split by how often it is touched, and each part goes where there is room:
//   HOT   per roll: generator state (16 B) per seat, in LDS for every seat (k = 8: 128 B per lane; three 256-thread blocks per CU with
//         the tables below).  The buffered half word (4 B) was hot until round 5; it now rides in the spare dword of the cold slot (BP);
//   COLD  per turn: the eight behaviour counters, the banked total and has_scored, packed into three dwords per seat
//         (fk_device.h: every counter field ends in a guard bit).  The turn owner's live in three registers; at a turn
//         hand-over they go to a PLANE indexed by (resident lane, seat) — 16 k bytes per lane, 25 MB for the whole chip at
//         k = 8, L2-resident once the increments no longer compete for it — and the next owner's come in.  One store + one
//         load per turn; the load is issued at the hand-over and first read in the middle of the next roll, so its latency
//         is not on the roll's dependency chain;
//   COLD IN LDS (CL instances; the launch plan's choice at k = 4, an option at k = 3 and 5): with three dwords the cold record
//         fits LDS beside the hot part — 32 bytes per seat instead of fk_play_kernel's 40: five waves per SIMD at k = 4 where
//         ten-dword records seat four (+5 % games/s), six at k = 3 and four at k = 5 (both +-0) — with no plane, no select trees
//         and the global tables (the texture addresser is not saturated there).  k = 4 launches four 320-thread blocks per CU:
//         five 256-thread blocks of exactly 32 KB do not fit 160 KB of LDS once each is rounded up to the allocation granule;
//   READ-ONLY per seat (KI instances, k <= 12): the PCG increment (4 dwords; and in the three-wave instances up to six seats the packed
//         strategy, 2 dwords) of EVERY seat stay in registers for the whole game and are picked by a select tree on the seat index at a
//         turn start (k - 1 v_cndmask per dword): no increment / strategy request per turn, no increment lines in L2.  Round 5 measured
//         the alternative — increments in 32-byte cold-plane slots, the next owner's fetched one turn ahead (IP) — 15 - 47 % slower: a
//         third scattered access per turn is more than the texture addresser has to spare;
//   TABLES (LT instances): the score / discard tables are read from a 12.4-KB LDS image (fk_device.h) instead of the 1 MiB /
//         64 KiB global tables — two gathers per roll less for the texture addresser (-14 % kernel time here; in
//         fk_play_kernel, which does not saturate it, the same change measured +1.7 % and was dropped);
//   has_buf of all seats is one bit mask per lane; the seats' strategy indices are eight 16-bit fields in four registers.
//   FOUR WAVES per SIMD for k = 5 .. 7 (WPE = 4, PKR_I = false): with the packed strategies loaded per turn instead, 128
//         registers hold the increments of six or seven seats without a spill in the roll loop, and the hot planes + table
//         image fit as 4 x 256 threads (k = 5), 2 x 512 (k = 6) or 1 x 1 024 (k = 7) per CU.  k = 8 stays at three waves: at four its
//         cold plane (1 024 lanes x 8 x 16 B = 131 KB per CU) no longer fits the CU's 128 KB of L2 (measured +8 %, round 5).
//   NINE TO TWELVE seats (round 5): one 768-thread block per CU (three waves per SIMD, 168 VGPRs, KI = 10 / 12); the hot planes are
//         16 k bytes per lane (147 KB at twelve seats) beside the table image.
// Measured against fk_play_kernel in the same process on the 5 160-strategy grid (tools/exp_hc2.py, tools/exp_hc3.py):
// k = 8 +27 %, k = 7 +25 %, k = 6 +23 %, k = 5 +10 % games/s (three-wave instances: +27 / +20 / +13 / +-0 %); k = 4 +-0 at five
// waves (spilling) and -2 % at four, k = 3 -7 %: the launch plan picks the register instances from k = 5 and the cold-in-LDS
// instance at k = 4.
// Cold record: fk_device.h (HC_*).  The 12-bit total needs target / 50 + one turn (<= 1310) < 4096; the launch plan keeps other
// tables on fk_play_kernel, and a count that reaches its guard bit (2 048 rolls, 256 farkles, 512 smart-discard uses, 1 024
// discarded dice, 256 hot-dice turns of one seat in one game) is FK_ERR_COUNTER_OVERFLOW like every other guarded counter
// (the host then replays the call on fk_play_kernel).
#pragma once

constexpr uint32_t HC_MAX_K = 12; // (round 5: 9 .. 12 seats — the reference's production list is n_players_list [2, 3, 4, 5, 6, 8, 10, 12])

// entry s (< N) of an N-entry register array, picked by a select tree on the bits of s: N - 1 v_cndmask.  get(t) must return the t-th entry
// for a COMPILE-TIME t (the arrays live in registers).  Written as a recursion over scalars: a select between two elements of a local
// ARRAY is turned by the compiler into an indexed load from scratch.
template <int N, int LO, int CNT, typename F>
__device__ __forceinline__ uint32_t hc_pick_r(uint32_t s, F get) {
    if constexpr (CNT == 1) {
        return get(LO < N ? LO : N - 1);
    } else {
        constexpr int H = CNT / 2;
        if constexpr (LO + H >= N) { // s < N: the upper half cannot be meant
            return hc_pick_r<N, LO, H>(s, get);
        } else {
            const uint32_t lo = hc_pick_r<N, LO, H>(s, get), hi = hc_pick_r<N, LO + H, H>(s, get);
            return (s & (uint32_t)H) ? hi : lo;
        }
    }
}
template <int N, typename F>
__device__ __forceinline__ uint32_t hc_pick(uint32_t s, F get) {
    static_assert(N >= 1 && N <= 16, "select tree over at most sixteen entries");
    return hc_pick_r<N, 0, 16>(s, get);
}

// KI: seats whose PCG increments (and, up to KI = 6 and unless PKR_I is off, packed strategies) stay in registers for the
// whole game (0: both are loaded at every turn start); LT: tables from the LDS image; WPE: waves per SIMD the register
// budget is cut for (0: 4 for KI = 4, 3 for the other KI instances); CL: cold records in LDS beside the buffered half words
// (no plane); NS: the most seats a launch of the instance has (strategy-index select).  See the file comment.
// (Rejected variants — increments in the cold-plane slots fetched one turn ahead, cold records in registers, increments loaded per turn —
// were template paths of this kernel until round 6; their A/B logs are profiles/r04_cold_records_in_registers.log and
// profiles/r05_increments_in_the_plane.log, their code is in the repository's history.)
template <int HC_BLOCK_I, uint32_t MIXED, bool LT, int KI = 0, int WPE = 0, bool PKR_I = true, bool CL = false, int NS = 8>
__global__ __launch_bounds__(HC_BLOCK_I) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : KI == 4 ? 4 : KI ? 3 : (HC_BLOCK_I == 256 ? 5 : HC_BLOCK_I / 256)))) void fk_play_hc_kernel(PlayArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t HC_BLOCK = (uint32_t)HC_BLOCK_I;
    const uint32_t tid = threadIdx.x;
    const uint32_t K = a.k;
    clock_stamp(a.clk, 0u);
    static_assert(!CL || (KI == 0 && !LT), "cold-in-LDS instances load increments / strategies per turn and gather from the global tables");
    // BP: the buffered half word of a seat that is not the turn owner rides in the fourth dword of its cold-plane slot (round 5): it is
    // touched once per turn like the rest of the cold record, the owner's lives in a register — 16 instead of 20 bytes of LDS per seat
    // and lane (twelve seats: 768 instead of 576 lanes per CU), two LDS instructions per roll less.  Instances without a plane keep it in LDS.
    constexpr bool BP = !CL;
    constexpr uint32_t HOT_DW = BP ? 4u : 5u;
    uint4 *const lds_state = reinterpret_cast<uint4 *>(lds) + tid;   // [seat][lane] generator state
    uint32_t *const lds_buf = lds + 4u * K * HC_BLOCK + tid;          // [seat][lane] buffered half word (!BP)
    // CL: [seat][lane] cold x, y (one 8-byte plane) and z — planes of the access width, so that a wave's lanes fall on distinct banks
    uint2 *const lds_xy = reinterpret_cast<uint2 *>(lds + 5u * K * HC_BLOCK) + tid;
    uint32_t *const lds_z = lds + 7u * K * HC_BLOCK + tid;
    // seat s's offset in a [seat][lane] plane (a 24-bit multiply where the block size is not a power of two)
    auto SB = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t {
        return (HC_BLOCK & (HC_BLOCK - 1u)) ? __umul24(s, HC_BLOCK) : s * HC_BLOCK;
    };
    const uint8_t *const lt_img = reinterpret_cast<const uint8_t *>(lds + HOT_DW * K * HC_BLOCK); // LT: the two tables
    if (LT) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds + HOT_DW * K * HC_BLOCK);
        const uint4 *src = reinterpret_cast<const uint4 *>(a.lds_tables);
        for (uint32_t i = tid; i < LT_BYTES / 16u; i += HC_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    uint4 *const cold = CL ? nullptr : a.cold + (size_t)(blockIdx.x * HC_BLOCK + tid) * K; // [resident lane][seat] slots
    // (buffered half word when BP, x, y, z)
    auto cold_load = [&](uint32_t s) __attribute__((always_inline)) -> uint4 {
        if (CL) {
            const uint2 xy = lds_xy[SB(s)];
            return make_uint4(0u, xy.x, xy.y, lds_z[SB(s)]);
        }
        const uint4 c = cold[s];
        return make_uint4(c.w, c.x, c.y, c.z);
    };

    enum : uint32_t { ST_FRESH = 0, ST_ACTIVE = 1, ST_ENDED = 2, ST_DONE = 3 };
    uint32_t st = ST_FRESH;
    uint32_t pool_next = 0, pool_end = 0, exhausted = 0; // wave-uniform ticket pool

    // game registers
    uint32_t game_id = 0, seat = 0, rounds = 0, max_rounds = 0, trigger = 0, seed_slot = 0;
    uint32_t final_round = 0, safety = 0;
    int32_t score_to_beat = 0;
    uint32_t hasbuf = 0;                          // bit s: seat s holds a buffered half word
    constexpr int NIX = (NS + 1) / 2;
    uint32_t ix0 = 0, ix1 = 0, ix2 = 0, ix3 = 0, ix4 = 0, ix5 = 0; // strategy indices of the seats, 16 bits each (scalars, not an array:
                                                  // the compiler turns a select between two ARRAY elements into an indexed load from scratch)
    // turn registers
    uint32_t dice = 6, rolls_this_turn = 0;
    int32_t turn_score = 0;
    uint64_t own_inc_lo = 0, own_inc_hi = 0;
    int32_t own_thr = 0;
    uint32_t own_bits = 0;
    uint32_t cX = 0, cY = 0, cZ = 0;              // the owner's cold record
    uint32_t own_buf = 0;                         // BP: the owner's buffered half word
    constexpr bool IR = KI != 0;                  // increments in registers
    uint32_t inc_r[IR ? KI : 1][4] = {};          // KI: every seat's increment (constant indices only: registers)
    constexpr bool PKR = PKR_I && IR && KI <= 6;      // ... and packed strategy, while 168 registers hold both without spilling
    uint32_t pk_r[PKR ? KI : 1][2] = {};

    auto seat_index = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t {
        const uint32_t w = hc_pick<NIX>(s >> 1, [&](int t) __attribute__((always_inline)) {
            return t == 0 ? ix0 : t == 1 ? ix1 : t == 2 ? ix2 : t == 3 ? ix3 : t == 4 ? ix4 : ix5;
        });
        return (s & 1u) ? (w >> 16) : (w & 0xffffu);
    };

    auto G = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t * {
        return a.state + ((size_t)seed_slot * K + s) * a.state_dw;
    };

    // turn owner := seat s (engine.py:236-240).  Three loads, none of them read before the next roll: the increment at
    // its first generator step, the strategy and the cold record behind the score-table gather.
    auto begin_turn = [&](uint32_t s) __attribute__((always_inline)) {
        uint4 inc;
        if (IR) { // select tree on the bits of s (entries beyond k are never selected)
            constexpr int NI = IR ? KI : 1;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = hc_pick<NI>(s, [&](int t) __attribute__((always_inline)) { return inc_r[t][j]; });
            inc = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            inc = a.inc[(size_t)seed_slot * K + s];
        }
        uint2 pk;
        if (PKR) {
            constexpr int NP = PKR ? KI : 1;
            pk = make_uint2(hc_pick<NP>(s, [&](int t) __attribute__((always_inline)) { return pk_r[t][0]; }), hc_pick<NP>(s, [&](int t) __attribute__((always_inline)) { return pk_r[t][1]; }));
        } else {
            pk = a.strat[seat_index(s)];
        }
        const uint4 c = cold_load(s);
        own_inc_lo = (uint64_t)inc.x | ((uint64_t)inc.y << 32);
        own_inc_hi = (uint64_t)inc.z | ((uint64_t)inc.w << 32);
        own_thr = (int32_t)pk.x;
        own_bits = pk.y;
        cX = c.y, cY = c.z, cZ = c.w;
        if (BP) own_buf = c.x;
        dice = 6;
        turn_score = 0;
        rolls_this_turn = 0;
    };

    auto raise = [&](int32_t code) {
        if (atomicCAS(&a.err[0], 0, code) == 0) a.err[1] = (int32_t)game_id;
        st = ST_DONE;
    };

    auto seat_turns = [&](uint32_t s) -> uint32_t { return rounds + ((final_round != 0u && s < trigger) ? 1u : 0u); };

    // ---- finished game -> one result record (run_tournament.py:375-391), final seat records on request ----
    auto finish_game = [&]() {
        const bool completed = (safety == 0u);
        // the owner's cold record is in registers (its store may still be on its way), the others come from the plane / LDS
        uint32_t w = 0;
        int32_t best = -1;
        uint4 wrec = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t s = 0; s < K; ++s) { // stable sort on score desc: first maximum wins (engine.py:477)
            const uint4 c = (s == seat) ? make_uint4(own_buf, cX, cY, cZ) : cold_load(s);
            const int32_t sc = (int32_t)((c.w >> HC_SCORE_SHIFT) & HC_SCORE_MASK);
            if (sc > best) {
                best = sc;
                w = s;
                wrec = c;
            }
            if (a.gs_out) { // the state store's format (R_*): score, n_turns and hot dice spelled out
                uint4 *g = reinterpret_cast<uint4 *>(G(s));
                g[0] = lds_state[SB(s)];
                g[1] = make_uint4(BP ? c.x : lds_buf[SB(s)], (uint32_t)sc,
                                  (c.y & HC_ROLLS_MASK) | (((c.y >> HC_FARKLE_SHIFT) & HC_FARKLE_MASK) << 16), (c.w & HC_HI_MASK) | (seat_turns(s) << 16));
                g[2] = make_uint4(((c.y >> HC_S5U_SHIFT) & HC_USES_MASK) | ((c.z & HC_DICE_MASK) << 16),
                                  ((c.z >> HC_S1U_SHIFT) & HC_USES_MASK) | (((c.z >> HC_D1_SHIFT) & HC_DICE_MASK) << 16),
                                  ((c.w >> HC_HOT_SHIFT) & HC_HOT_MASK) | ((c.y & HC_HAS_SCORED) ? CE_HAS_SCORED : 0u) | (((hasbuf >> s) & 1u) ? CE_HAS_BUF : 0u),
                                  seat_index(s));
            }
        }
        if (!a.rec0) return;
        const uint32_t widx = completed ? seat_index(w) : 0u;
        const uint32_t d0 = widx | (completed ? (w << 24) : REC_SAFETY);
        a.rec0[game_id] = d0;
        if (a.recs) {
            if (!completed) wrec = make_uint4(0u, 0u, 0u, 0u);
            uint4 *r = reinterpret_cast<uint4 *>(a.recs + (size_t)game_id * REC_DW);
            r[0] = make_uint4(d0, completed ? (uint32_t)best * 50u : 0u, rounds | (((wrec.y >> HC_FARKLE_SHIFT) & HC_FARKLE_MASK) << 16),
                              (wrec.y & HC_ROLLS_MASK) | (((wrec.w & HC_HI_MASK) * 50u) << 16));
            r[1] = make_uint4(((wrec.y >> HC_S5U_SHIFT) & HC_USES_MASK) | ((wrec.z & HC_DICE_MASK) << 16),
                              ((wrec.z >> HC_S1U_SHIFT) & HC_USES_MASK) | (((wrec.z >> HC_D1_SHIFT) & HC_DICE_MASK) << 16),
                              (wrec.w >> HC_HOT_SHIFT) & HC_HOT_MASK, 0u);
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
        uint32_t slot = a.sched ? ticket : id;
        if (!a.sched) { // the seed kernel's walk order (shuffle-minor)
            const uint32_t sh = id / a.gps, g = id - sh * a.gps;
            slot = g * a.n_sh + sh;
        }
        seed_slot = slot;
        uint32_t iw0 = 0, iw1 = 0, iw2 = 0, iw3 = 0, iw4 = 0, iw5 = 0;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)NS; ++s) { // (NS: the most seats a launch of the instance has)
            if (s < K) {
                const uint32_t *src = G(s);
                lds_state[SB(s)] = *reinterpret_cast<const uint4 *>(src);
                if (!BP) lds_buf[SB(s)] = 0u;
                if (CL) {
                    lds_xy[SB(s)] = make_uint2(0u, 0u);
                    lds_z[SB(s)] = 0u;
                } else {
                    cold[s] = make_uint4(0u, 0u, 0u, 0u); // the previous game's record of this lane
                }
                const uint32_t idx = (a.state_dw == STATE_DW) ? src[R_IDX] : (uint32_t)a.seat_idx[(size_t)slot * K + s];
                const uint32_t field = idx << (16u * (s & 1u));
                if ((s >> 1) == 0u) iw0 |= field;
                else if ((s >> 1) == 1u) iw1 |= field;
                else if ((s >> 1) == 2u) iw2 |= field;
                else if ((s >> 1) == 3u) iw3 |= field;
                else if ((s >> 1) == 4u) iw4 |= field;
                else iw5 |= field;
                if (IR && s < (uint32_t)(IR ? KI : 1)) {
                    const uint4 q = a.inc[(size_t)slot * K + s];
                    inc_r[s < (uint32_t)(IR ? KI : 1) ? s : 0][0] = q.x, inc_r[s < (uint32_t)(IR ? KI : 1) ? s : 0][1] = q.y;
                    inc_r[s < (uint32_t)(IR ? KI : 1) ? s : 0][2] = q.z, inc_r[s < (uint32_t)(IR ? KI : 1) ? s : 0][3] = q.w;
                    if (PKR) {
                        const uint2 pk = a.strat[idx];
                        pk_r[s < (uint32_t)(PKR ? KI : 1) ? s : 0][0] = pk.x, pk_r[s < (uint32_t)(PKR ? KI : 1) ? s : 0][1] = pk.y;
                    }
                }
            }
        }
        ix0 = iw0, ix1 = iw1, ix2 = iw2, ix3 = iw3, ix4 = iw4, ix5 = iw5;
        hasbuf = 0;
        cX = cY = cZ = 0u;
        own_buf = 0u;
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

    // ---- after a turn: advance the table (engine.py:453-472, 523-550), as selects ----
    auto advance = [&](int32_t score) __attribute__((always_inline)) {
        const bool fr = final_round != 0u;
        const bool trig = !fr & (score >= a.target50);
        const bool normal = !fr & !trig;
        const uint32_t n1 = seat + 1u;
        const bool wrap = n1 == K;
        const bool last = normal & wrap & (rounds >= max_rounds);
        uint32_t next_fr = n1 + ((n1 == trigger) ? 1u : 0u);
        uint32_t next_tr = (seat == 0u) ? 1u : 0u, next_nm = wrap ? 0u : n1;
        asm volatile("" : "+v"(next_fr), "+v"(next_tr), "+v"(next_nm));
        const uint32_t next = fr ? next_fr : trig ? next_tr : next_nm;
        rounds += (normal & wrap & !last) ? 1u : 0u;
        safety = last ? 1u : safety;
        score_to_beat = trig ? score : (fr & (score > score_to_beat)) ? score : score_to_beat;
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

    // ---- one roll of the current turn (engine.py:241-273) ----
    auto roll_step = [&]() __attribute__((always_inline)) {
        const bool roll_limit = rolls_this_turn >= 1000u; // ROLL_LIMIT, engine.py:36,242
        const uint32_t s = seat;
        const uint4 sv = lds_state[SB(s)];
        const uint32_t buf0 = BP ? own_buf : lds_buf[SB(s)];
        Rng rng{(uint64_t)sv.z | ((uint64_t)sv.w << 32), (uint64_t)sv.x | ((uint64_t)sv.y << 32), own_inc_hi, own_inc_lo, buf0, (hasbuf >> s) & 1u};
        const uint32_t n = dice;
        bool detour;
        uint32_t key = roll_counts_fast<3>(rng, n, detour);
        if (detour) { // a Lemire rejection (once in ~2^30 dice): the generator comes back from the hot planes, still the roll's input
            asm volatile("" ::: "memory"); // really re-read them: values forwarded from the loads above would stay live across the whole roll
            const uint4 sv0 = lds_state[SB(s)];
            rng.lo = (uint64_t)sv0.x | ((uint64_t)sv0.y << 32);
            rng.hi = (uint64_t)sv0.z | ((uint64_t)sv0.w << 32);
            rng.buf = buf0;
            rng.has_buf = (hasbuf >> s) & 1u;
            key = roll_counts_sequential<3>(rng, n, nullptr);
        }
        rolls_this_turn += 1u;
        int32_t dthr = (int32_t)(int8_t)(own_bits & 0xffu);
        asm volatile("" : "+v"(dthr));
        const Strat50 sp{own_thr, (own_bits & (0xffu | MIXED)) | (a.uflags & (0xff00u & ~MIXED)), dthr};
        const Roll50 rr = LT ? default_score_lds50(lt_img, key, (int32_t)n, turn_score, sp)
                             : default_score_lut50(a.score_lut, a.discard_lut, key, (int32_t)n, turn_score, sp);
        const bool farkle = rr.score50 == 0;
        const int32_t score = (int32_t)((cZ >> HC_SCORE_SHIFT) & HC_SCORE_MASK);
        cX += 1u + (farkle ? (1u << HC_FARKLE_SHIFT) : 0u) + ((rr.d5 > 0) ? (1u << HC_S5U_SHIFT) : 0u);
        cY += (uint32_t)rr.d5 + ((uint32_t)rr.d1 << HC_D1_SHIFT) + ((rr.d1 > 0) ? (1u << HC_S1U_SHIFT) : 0u);
        dice = (rr.used == (int32_t)n) ? 6u : (n - (uint32_t)rr.used);
        turn_score = farkle ? 0 : (turn_score + rr.score50);
        const bool hot = !farkle & sp.has(SF_AUTO_HOT) & (dice == 6u);
        cZ += hot ? (1u << HC_HOT_SHIFT) : 0u;
        const bool keep = should_continue50(sp, turn_score, (int32_t)dice, (cX & HC_HAS_SCORED) != 0u, final_round != 0u, score_to_beat, score);
        const bool over = farkle | (!hot & !keep);
        const uint32_t ts = over ? (uint32_t)turn_score : 0u;
        cX |= (ts >= 10u) ? HC_HAS_SCORED : 0u; // 500 points
        const uint32_t banked = (cX & HC_HAS_SCORED) ? ts : 0u;
        cZ += banked << HC_SCORE_SHIFT;
        cZ = (banked > (cZ & HC_HI_MASK)) ? ((cZ & ~HC_HI_MASK) | banked) : cZ;
        const bool overflow = (turn_score > 1310) | (((cX & HC_X_GUARD) | (cY & HC_Y_GUARD) | (cZ & HC_Z_GUARD)) != 0u);
        if (roll_limit | overflow) {
            raise(roll_limit ? FK_ERR_ROLL_LIMIT : FK_ERR_COUNTER_OVERFLOW);
            return;
        }
        lds_state[SB(s)] = make_uint4((uint32_t)rng.lo, (uint32_t)(rng.lo >> 32), (uint32_t)rng.hi, (uint32_t)(rng.hi >> 32));
        hasbuf = (hasbuf & ~(1u << s)) | (rng.has_buf << s);
        if (BP) own_buf = rng.buf;
        else lds_buf[SB(s)] = rng.buf;
        if (over) {
            if (CL) {
                lds_xy[SB(s)] = make_uint2(cX, cY);
                lds_z[SB(s)] = cZ;
            } else {
                cold[s] = make_uint4(cX, cY, cZ, own_buf);
            }
            advance(score + (int32_t)banked);
        }
    };

    // ---- wave-level hand-over (as fk_play_kernel) ----
    auto handover = [&](uint64_t waiting) {
        const bool mine = (st == ST_FRESH || st == ST_ENDED);
        if (st == ST_ENDED) finish_game();
        const uint32_t n = (uint32_t)__popcll(waiting);
        const uint32_t avail = pool_end - pool_next;
        uint32_t new_base = 0, new_avail = 0;
        if (avail < n && !exhausted) {
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

    auto handover_due = [&](uint64_t waiting, uint64_t active) -> bool {
        return waiting && (!active || (uint32_t)__popcll(waiting) >= a.batch_threshold || exhausted);
    };
    while (true) {
        uint64_t waiting = __ballot(st == ST_FRESH || st == ST_ENDED);
        uint64_t active = __ballot(st == ST_ACTIVE);
        if (!(waiting | active)) break;
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
    clock_stamp(a.clk, 1u);
}
