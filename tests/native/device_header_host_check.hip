// Host-side check of the __host__ __device__ scoring code in farkle_ii_amd/csrc/fk_device.h (no GPU calls):
//   1. SWAR default_score == the readable loop form default_score_loops on every multiset of 1..6 dice, every flag
//      combination a ThresholdStrategy can have, a grid of thresholds and turn scores;
//   2. the score-table entry of every key decodes to score_counts of that multiset;
//   3. the discard-table entry, applied through discard_query/discard_key, equals the loop form as well.
// Built and run by tests/test_device_header_host.py with hipcc (host compilation only).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../farkle_ii_amd/csrc/fk_device.h"

using namespace fk;

int main() {
    std::vector<uint8_t> dlut(DISCARD_LUT_KEYS);
    for (uint32_t k = 0; k < DISCARD_LUT_KEYS; ++k) dlut[k] = discard_lut_entry(k);
    std::vector<uint32_t> lut32(SCORE_LUT_KEYS);
    for (uint32_t k = 0; k < SCORE_LUT_KEYS; ++k) lut32[k] = score_lut_entry32(k);
    std::vector<uint8_t> lds_image(LT_BYTES, 0); // the hot / cold kernels' LDS image of both tables
    lt_build_image(lds_image.data());
    long cases = 0, bad_swar = 0, bad_table = 0, bad_lut = 0, multisets = 0, bad_fast = 0, fast_cases = 0;
    for (uint32_t key = 1; key < SCORE_LUT_KEYS; ++key) {
        uint32_t n = 0;
        bool ok = true;
        for (int f = 0; f < 6; ++f) {
            const uint32_t c = (key >> (3 * f)) & 7u;
            ok &= c <= 6u;
            n += c;
        }
        if (!ok || n == 0 || n > 6) continue;
        ++multisets;
        const uint32_t nib = lut_key_to_nibbles(key);
        if (nibbles_to_lut_key(nib) != key) ++bad_lut;
        const RawScore raw = score_counts(nib), dec = raw_from_lut(score_lut_entry(key));
        if ((lut32[key] & 0x7fffu) != ((uint32_t)score_lut_entry(key) & 0x7fffu)) ++bad_lut;
        if (raw.score != dec.score || raw.used != dec.used || raw.sf != dec.sf || raw.so != dec.so) ++bad_lut;
        for (uint32_t flags = 0; flags < 256; ++flags) {
            // the two combinations ThresholdStrategy.__post_init__ rejects (strategies.py:196-207) never reach a kernel:
            // fk_* entry points return FK_ERR_ARG for them (validate_strategies)
            const uint32_t bits = flags << 8;
            if ((bits & SF_SMART_ONE) && !(bits & SF_SMART_FIVE)) continue;
            if ((bits & SF_REQUIRE_BOTH) && !((bits & SF_CONSIDER_SCORE) && (bits & SF_CONSIDER_DICE))) continue;
            for (int32_t thr : {0, 200, 300, 350, 500, 1000}) {
                for (int32_t dthr : {-1, 0, 1, 2, 3, 4, 6}) {
                    const Strat s{thr, ((uint32_t)(uint8_t)(int8_t)dthr) | (flags << 8)};
                    for (int32_t pre : {0, 50, 250, 300, 450, 950, 3000}) {
                        const RollResult want = default_score_loops(nib, (int32_t)n, pre, s);
                        const RollResult swar = default_score(nib, (int32_t)n, pre, s);
                        const DiscardQuery q = discard_query(raw, (int32_t)n, pre, s);
                        const RollResult tab = apply_discards(raw, q.eligible ? (uint32_t)dlut[discard_key(q)] : 0u);
                        ++cases;
                        if (swar.score != want.score || swar.used != want.used || swar.d5 != want.d5 || swar.d1 != want.d1) ++bad_swar;
                        if (tab.score != want.score || tab.used != want.used || tab.d5 != want.d5 || tab.d1 != want.d1) ++bad_table;
                        // 4. the game kernels' path (units of 50, 32-bit score entries carrying the roll's share of the discard key, the
                        //    strategy's share = three flag bits) and its readable twin, for thresholds and turn scores that are multiples of 50
                        if (pre % 50 == 0) {
                            const Strat50 s50 = to_units50(s);
                            const Roll50 fast = default_score_lut50(lut32.data(), dlut.data(), key, (int32_t)n, pre / 50, s50);
                            const Roll50 slow = default_score_lut50_decoded(lut32.data(), dlut.data(), key, (int32_t)n, pre / 50, s50);
                            ++fast_cases;
                            if (fast.score50 * 50 != want.score || fast.used != want.used || fast.d5 != want.d5 || fast.d1 != want.d1) ++bad_fast;
                            if (slow.score50 * 50 != want.score || slow.used != want.used || slow.d5 != want.d5 || slow.d1 != want.d1) ++bad_fast;
                            // 5. the same roll through the LDS image (pair table -> dense multiset index -> 32-bit entry -> additive discard index)
                            const Roll50 lds = default_score_lds50(lds_image.data(), key, (int32_t)n, pre / 50, s50);
                            if (lds.score50 * 50 != want.score || lds.used != want.used || lds.d5 != want.d5 || lds.d1 != want.d1) ++bad_fast;
                        }
                    }
                }
            }
        }
    }
    printf("multisets %ld cases %ld bad_swar %ld bad_table %ld bad_lut %ld fast_cases %ld bad_fast %ld\n", multisets, cases, bad_swar, bad_table, bad_lut,
           fast_cases, bad_fast);
    return (multisets == 923 && bad_swar == 0 && bad_table == 0 && bad_lut == 0 && bad_fast == 0 && fast_cases > 0) ? 0 : 1;
}
