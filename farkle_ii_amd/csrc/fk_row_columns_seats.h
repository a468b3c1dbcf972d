// fk_row_columns_seats.h — the column images of a launch with one thread per (game, SEAT) (round 6).
//
// fk_row_columns_kernel (fk_kernels.h) gives a game to one thread: 4 + 13 k plane stores and k^2 score reads per thread, and a 256-MB
// launch group of twelve-seat games is only 4 x 10^5 threads — 0.97 ms, 0.27 TB/s written (rows mode's groups are cut by image BYTES, so
// the more seats, the fewer games).  Here a workgroup is k waves over the same 64 games: wave s writes seat s's thirteen planes (lane =
// game, so every plane store is still 64 consecutive int32) and its byte of the rank order; wave 0 also writes the four game-level planes
// and the status / winner bytes.  The k waves read the same 48 k bytes of state per game at the same time (one fetch from L2, the rest
// L1 / TA hits).  Same images, byte for byte (tests/test_shard_writer.py compares both forms with the NumPy restatement).
#pragma once

__global__ __launch_bounds__(1024) void fk_row_columns_seats_kernel(const uint32_t *state, const uint32_t *recs, const uint32_t *inv_sched,
                                                                    uint32_t n_games, uint32_t gps, uint32_t n_sh, uint32_t k, uint32_t perm_mode,
                                                                    const int32_t *ids, uint8_t *out, size_t stride) {
    const uint32_t lane = threadIdx.x & 63u, s = threadIdx.x >> 6; // (blockDim.x = 64 k)
    const uint32_t id = blockIdx.x * 64u + lane;
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
    const uint32_t *x = gs + (size_t)s * STATE_DW;
    const int32_t sc = (int32_t)x[R_SCORE];
    const int32_t win50 = (int32_t)gs[(size_t)w * STATE_DW + R_SCORE];
    // one pass over the game's scores: this seat's rank (stable: ties go to the lower seat) and, for wave 0, the runner-up
    uint32_t rank = 1;
    int32_t second50 = 0;
    bool any = false;
    for (uint32_t j = 0; j < k; ++j) {
        const int32_t o = (int32_t)gs[(size_t)j * STATE_DW + R_SCORE];
        rank += (o > sc || (o == sc && j < s)) ? 1u : 0u;
        if (j != w && (!any || o > second50)) {
            second50 = o;
            any = true;
        }
    }
    if (!completed) rank = 0;
    uint8_t *order = bytes + (size_t)2 * gps + (size_t)g * k;
    order[completed ? rank - 1u : s] = completed ? (uint8_t)s : (uint8_t)0;
    if (s == 0u) {
        plane[0] = completed ? ids[gs[(size_t)w * STATE_DW + R_IDX]] : 0;
        plane[(size_t)gps] = completed ? win50 * 50 : 0;
        plane[(size_t)2 * gps] = completed ? (win50 - (any ? second50 : 0)) * 50 : 0;
        plane[(size_t)3 * gps] = (int32_t)(q0.z & 0xffffu);
        bytes[g] = completed ? 0u : 1u;
        bytes[gps + g] = (uint8_t)w;
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
