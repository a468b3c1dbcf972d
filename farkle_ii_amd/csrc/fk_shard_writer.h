// Row shards straight from the device's column images: a Parquet writer for ONE schema family, on host threads.
//
// The reference persists game rows as one Parquet file per shuffle (`rows_<root>_<k>p_<shuffle:012d>.parquet`, schema
// `raw_simulation_schema_for(k)`, src/farkle/utils/schema_helpers.py:23-90; written per shuffle at
// src/farkle/simulation/run_tournament.py:530-558).  Through Arrow that costs 3 - 5 ms of host time per shard (18 + 14k column
// chunks, each with its own encoder set-up) — 7.2 of the 7.5 seconds of the production sweep with rows on, against 0.3 s of GPU.
// Here the GPU emits, per shuffle, the VALUES of every column already in their Parquet physical type (`fk_row_columns_kernel`:
// int32 planes in (column, game) order + three byte planes), and this writer only frames them: page headers, definition /
// repetition levels (one bitmap per shard: every nullable field of a row is null exactly when the game hit the safety limit),
// bit-packed booleans and dictionary indices, the footer.  Uncompressed PLAIN / RLE_DICTIONARY pages, one row group, no
// statistics — the encodings the Parquet specification calls mandatory for readers.  The schema elements, the ARROW:schema
// key-value entry (so that a reader restores int16 / int8 / list<item: string> exactly) and the column orders are taken verbatim
// from the footer of a file Arrow wrote for the same schema (farkle_ii_amd/parquet_template.py), so the Arrow schema a reader
// sees — and the contract-v3 schema fingerprint — is identical to the Arrow-written shard's.
//
// Also here, because the same threads have the bytes in hand: SHA-256 of every shard (manifest / completion identities) and,
// for artifact-contract version 3, the shard's sidecar from the constant template text (contract_v3.SimulationContract.shard_template).
//
// Host code only: callable without a GPU (the CPU test-suite drives it with column images built by NumPy).
#pragma once

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include <immintrin.h>

namespace fksw {

// ---- SHA-256 (FIPS 180-4): SHA-NI when the CPU has it, portable rounds otherwise ----
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_blocks_portable(uint32_t st[8], const uint8_t *p, size_t n_blocks) {
    for (; n_blocks; --n_blocks, p += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
    }
}

__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_blocks_ni(uint32_t st[8], const uint8_t *p, size_t n_blocks) {
    const __m128i mask = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[0]));
    __m128i s1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[4]));
    tmp = _mm_shuffle_epi32(tmp, 0xB1);       // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);         // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8); // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);      // CDGH
    for (; n_blocks; --n_blocks, p += 64) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
        for (int i = 0; i < 4; ++i) m[i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16 * i)), mask);
        for (int r = 0; r < 16; ++r) {
            __m128i msg = _mm_add_epi32(m[r & 3], _mm_loadu_si128(reinterpret_cast<const __m128i *>(&K256[4 * r])));
            s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
            if (r < 12) { // schedule words 4 (r + 4) .. 4 (r + 4) + 3 into the slot of words 4 r ..
                __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
            }
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
    }
    tmp = _mm_shuffle_epi32(s0, 0x1B);       // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);        // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);     // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);        // HGFE
    _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[0]), s0);
    _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[4]), s1);
}

// Two independent messages in lockstep: one message's sha256rnds2 chain is latency-bound (every round needs the previous one's state), a
// second chain in the same loop fills the other issue slots — 1.6 - 1.9 x the single-stream rate on the hosts this ran on.  Both advance
// by the same number of blocks.
__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_blocks_ni_x2(uint32_t sta[8], const uint8_t *pa, uint32_t stb[8], const uint8_t *pb, size_t n_blocks) {
    const __m128i mask = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    auto load = [](const uint32_t st[8], __m128i &s0, __m128i &s1) __attribute__((target("sha,sse4.1,ssse3"))) {
        __m128i tmp = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[0])), 0xB1);
        s1 = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[4])), 0x1B);
        s0 = _mm_alignr_epi8(tmp, s1, 8);
        s1 = _mm_blend_epi16(s1, tmp, 0xF0);
    };
    auto store = [](uint32_t st[8], __m128i s0, __m128i s1) __attribute__((target("sha,sse4.1,ssse3"))) {
        const __m128i tmp = _mm_shuffle_epi32(s0, 0x1B);
        s1 = _mm_shuffle_epi32(s1, 0xB1);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[0]), _mm_blend_epi16(tmp, s1, 0xF0));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[4]), _mm_alignr_epi8(s1, tmp, 8));
    };
    __m128i a0, a1, b0, b1;
    load(sta, a0, a1);
    load(stb, b0, b1);
    for (; n_blocks; --n_blocks, pa += 64, pb += 64) {
        const __m128i sa0 = a0, sa1 = a1, sb0 = b0, sb1 = b1;
        __m128i ma[4], mb[4];
        for (int i = 0; i < 4; ++i) {
            ma[i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(pa + 16 * i)), mask);
            mb[i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(pb + 16 * i)), mask);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const __m128i kk = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&K256[4 * r]));
            __m128i xa = _mm_add_epi32(ma[r & 3], kk), xb = _mm_add_epi32(mb[r & 3], kk);
            a1 = _mm_sha256rnds2_epu32(a1, a0, xa);
            b1 = _mm_sha256rnds2_epu32(b1, b0, xb);
            xa = _mm_shuffle_epi32(xa, 0x0E);
            xb = _mm_shuffle_epi32(xb, 0x0E);
            a0 = _mm_sha256rnds2_epu32(a0, a1, xa);
            b0 = _mm_sha256rnds2_epu32(b0, b1, xb);
            if (r < 12) {
                __m128i ya = _mm_sha256msg1_epu32(ma[r & 3], ma[(r + 1) & 3]), yb = _mm_sha256msg1_epu32(mb[r & 3], mb[(r + 1) & 3]);
                ya = _mm_add_epi32(ya, _mm_alignr_epi8(ma[(r + 3) & 3], ma[(r + 2) & 3], 4));
                yb = _mm_add_epi32(yb, _mm_alignr_epi8(mb[(r + 3) & 3], mb[(r + 2) & 3], 4));
                ma[r & 3] = _mm_sha256msg2_epu32(ya, ma[(r + 3) & 3]);
                mb[r & 3] = _mm_sha256msg2_epu32(yb, mb[(r + 3) & 3]);
            }
        }
        a0 = _mm_add_epi32(a0, sa0);
        a1 = _mm_add_epi32(a1, sa1);
        b0 = _mm_add_epi32(b0, sb0);
        b1 = _mm_add_epi32(b1, sb1);
    }
    store(sta, a0, a1);
    store(stb, b0, b1);
}

static bool have_sha_ni() {
    static const bool yes = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1");
    return yes;
}

static void sha256(const uint8_t *data, size_t len, uint8_t out[32], bool force_portable = false) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const bool ni = !force_portable && have_sha_ni();
    const size_t whole = len / 64;
    if (whole) (ni ? sha256_blocks_ni : sha256_blocks_portable)(st, data, whole);
    uint8_t tail[128] = {0};
    const size_t rest = len - whole * 64;
    std::memcpy(tail, data + whole * 64, rest);
    tail[rest] = 0x80;
    const size_t tail_blocks = rest + 9 <= 64 ? 1 : 2;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) tail[tail_blocks * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
    (ni ? sha256_blocks_ni : sha256_blocks_portable)(st, tail, tail_blocks);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

static void sha256_finish(uint32_t st[8], const uint8_t *rest_data, size_t rest, size_t total_len, uint8_t out[32], bool ni) {
    uint8_t tail[128] = {0};
    std::memcpy(tail, rest_data, rest);
    tail[rest] = 0x80;
    const size_t tail_blocks = rest + 9 <= 64 ? 1 : 2;
    const uint64_t bits = (uint64_t)total_len * 8;
    for (int i = 0; i < 8; ++i) tail[tail_blocks * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
    (ni ? sha256_blocks_ni : sha256_blocks_portable)(st, tail, tail_blocks);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

// SHA-256 of two buffers at once (the shard writer hashes the two files a thread has just built): the common prefix of whole blocks in
// lockstep on SHA-NI hosts, the rest of the longer one alone.
static void sha256_pair(const uint8_t *da, size_t la, uint8_t outa[32], const uint8_t *db, size_t lb, uint8_t outb[32], bool force_portable = false) {
    const bool ni = !force_portable && have_sha_ni();
    if (!ni) {
        sha256(da, la, outa, true);
        sha256(db, lb, outb, true);
        return;
    }
    uint32_t sa[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}, sb[8];
    std::memcpy(sb, sa, sizeof sa);
    const size_t wa = la / 64, wb = lb / 64, common = wa < wb ? wa : wb;
    if (common) sha256_blocks_ni_x2(sa, da, sb, db, common);
    if (wa > common) sha256_blocks_ni(sa, da + common * 64, wa - common);
    if (wb > common) sha256_blocks_ni(sb, db + common * 64, wb - common);
    sha256_finish(sa, da + wa * 64, la - wa * 64, la, outa, true);
    sha256_finish(sb, db + wb * 64, lb - wb * 64, lb, outb, true);
}

static void hex32(const uint8_t d[32], char out[65]) {
    static const char *digits = "0123456789abcdef";
    for (int i = 0; i < 32; ++i) { out[2 * i] = digits[d[i] >> 4]; out[2 * i + 1] = digits[d[i] & 15]; }
    out[64] = 0;
}

// ---- byte sink with Thrift compact-protocol primitives (the subset Parquet metadata needs) ----
struct Sink {
    std::vector<uint8_t> b;
    size_t size() const { return b.size(); }
    void put(uint8_t v) { b.push_back(v); }
    void raw(const void *p, size_t n) { const uint8_t *q = static_cast<const uint8_t *>(p); b.insert(b.end(), q, q + n); }
    void le32(uint32_t v) { raw(&v, 4); }
    void varint(uint64_t v) { while (v >= 0x80) { put((uint8_t)(v | 0x80)); v >>= 7; } put((uint8_t)v); }
    void zz(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
    // field header with a delta of 1..15 from the previous field id
    void field(int delta, int type) { put((uint8_t)(delta << 4 | type)); }
    void i32f(int delta, int64_t v) { field(delta, 5); zz(v); }
    void i64f(int delta, int64_t v) { field(delta, 6); zz(v); }
    void list(int n, int elem_type) { if (n < 15) put((uint8_t)(n << 4 | elem_type)); else { put((uint8_t)(0xF0 | elem_type)); varint((uint64_t)n); } }
    void binary(const void *p, size_t n) { varint(n); raw(p, n); }
};
enum { T_I32 = 5, T_I64 = 6, T_BINARY = 8, T_LIST = 9, T_STRUCT = 12 };
enum { ENC_PLAIN = 0, ENC_RLE = 3, ENC_DELTA_BINARY_PACKED = 5, ENC_RLE_DICTIONARY = 8 };
enum { PT_BOOLEAN = 0, PT_INT32 = 1, PT_INT64 = 2, PT_BYTE_ARRAY = 6 };

static void page_header(Sink &s, bool dictionary, uint32_t payload, uint32_t num_values, int encoding) {
    s.i32f(1, dictionary ? 2 : 0); // 1: type (DATA_PAGE = 0, DICTIONARY_PAGE = 2)
    s.i32f(1, payload);            // 2: uncompressed_page_size
    s.i32f(1, payload);            // 3: compressed_page_size (codec UNCOMPRESSED)
    if (dictionary) {
        s.field(4, T_STRUCT);      // 7: dictionary_page_header
        s.i32f(1, num_values);
        s.i32f(1, ENC_PLAIN);
        s.put(0);
    } else {
        s.field(2, T_STRUCT);      // 5: data_page_header
        s.i32f(1, num_values);
        s.i32f(1, encoding);
        s.i32f(1, ENC_RLE);        // definition levels
        s.i32f(1, ENC_RLE);        // repetition levels
        s.put(0);
    }
    s.put(0);
}

static size_t varint_len(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }

// LSB-first bit packer (the bit-packed runs of Parquet's RLE / bit-packing hybrid; also PLAIN booleans)
struct BitPacker {
    std::vector<uint8_t> &out;
    uint64_t acc = 0;
    int fill = 0;
    explicit BitPacker(std::vector<uint8_t> &o) : out(o) {}
    void add(uint32_t v, int width) {
        acc |= (uint64_t)v << fill;
        fill += width;
        while (fill >= 8) { out.push_back((uint8_t)acc); acc >>= 8; fill -= 8; }
    }
    void flush() { if (fill) { out.push_back((uint8_t)acc); acc = 0; fill = 0; } }
};

struct ColumnMeta { // what the footer records of a column chunk
    int64_t num_values, total_size, data_page_offset, dictionary_page_offset; // dictionary_page_offset < 0: none
    bool dictionary, delta;
};

// ---- DELTA_BINARY_PACKED (Parquet encoding 5) of int32 values ----
// Header: block size (128 values), miniblocks per block (4), value count, first value; then per block of 128 deltas: the smallest delta,
// four miniblock bit widths, and each miniblock's 32 (delta - smallest) values bit-packed LSB-first.  The widths used here are 0, 8, 16
// and 32 — whole bytes, so "bit-packing" is a narrowing store — which already brings the counters of a row shard (farkles, rolls, turns,
// ranks: < 256) to one byte each and points and strategy ids to two: about a third of PLAIN's bytes to hash and to write.
// Arithmetic wraps modulo 2^32 on both sides (the reader adds the same way), so every int32 sequence round-trips.
static inline uint32_t zigzag32(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }

#define FK_DELTA_BODY                                                                                                                   \
    uint8_t head[24];                                                                                                                   \
    size_t h = 0;                                                                                                                       \
    auto varint = [&](uint64_t x) { while (x >= 0x80) { head[h++] = (uint8_t)(x | 0x80); x >>= 7; } head[h++] = (uint8_t)x; };          \
    varint(128); varint(4); varint(n); varint(zigzag32(n ? v[0] : 0));                                                                  \
    out.insert(out.end(), head, head + h);                                                                                              \
    const uint32_t n_deltas = n ? n - 1 : 0;                                                                                            \
    for (uint32_t first = 0; first < n_deltas; first += 128) {                                                                          \
        const uint32_t cnt = n_deltas - first < 128 ? n_deltas - first : 128;                                                           \
        const int32_t *src = v + first;                                                                                                 \
        int32_t d[128];                                                                                                                 \
        uint32_t a[128];                                                                                                                \
        for (uint32_t j = 0; j < cnt; ++j) d[j] = (int32_t)((uint32_t)src[j + 1] - (uint32_t)src[j]);                                   \
        int32_t mn = d[0];                                                                                                              \
        for (uint32_t j = 1; j < cnt; ++j) mn = d[j] < mn ? d[j] : mn;                                                                  \
        for (uint32_t j = 0; j < cnt; ++j) a[j] = (uint32_t)d[j] - (uint32_t)mn;                                                        \
        for (uint32_t j = cnt; j < 128; ++j) a[j] = 0;                                                                                  \
        uint8_t block[5 + 4 + 512];                                                                                                     \
        size_t b = 0;                                                                                                                   \
        for (uint32_t x = zigzag32(mn);; x >>= 7) { if (x >= 0x80) block[b++] = (uint8_t)(x | 0x80); else { block[b++] = (uint8_t)x; break; } } \
        uint8_t *widths = block + b;                                                                                                    \
        b += 4;                                                                                                                         \
        for (uint32_t m = 0; m < 4; ++m) {                                                                                              \
            const uint32_t *am = a + 32 * m;                                                                                            \
            if (32 * m >= cnt) { widths[m] = 0; continue; }                                                                             \
            uint32_t any = 0;                                                                                                           \
            for (int j = 0; j < 32; ++j) any |= am[j];                                                                                  \
            if (any == 0) widths[m] = 0;                                                                                                \
            else if (any < 0x100u) { widths[m] = 8; for (int j = 0; j < 32; ++j) block[b + j] = (uint8_t)am[j]; b += 32; }              \
            else if (any < 0x10000u) { widths[m] = 16; uint16_t t[32]; for (int j = 0; j < 32; ++j) t[j] = (uint16_t)am[j]; std::memcpy(block + b, t, 64); b += 64; } \
            else { widths[m] = 32; std::memcpy(block + b, am, 128); b += 128; }                                                         \
        }                                                                                                                               \
        out.insert(out.end(), block, block + b);                                                                                        \
    }

__attribute__((target("avx2"))) static void delta_pack_int32_avx2(std::vector<uint8_t> &out, const int32_t *v, uint32_t n) { FK_DELTA_BODY }
static void delta_pack_int32_base(std::vector<uint8_t> &out, const int32_t *v, uint32_t n) { FK_DELTA_BODY }
#undef FK_DELTA_BODY

// appends the encoding of v[0 .. n) to `out`
static void delta_pack_int32(std::vector<uint8_t> &out, const int32_t *v, uint32_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    (avx2 ? delta_pack_int32_avx2 : delta_pack_int32_base)(out, v, n);
}

struct Job {
    int32_t k, gps, n_shuffles, threads, atomic;
    uint64_t root_seed;
    int32_t rng_purpose_namespace;
    const int64_t *shuffle_index, *shuffle_seed;
    const int32_t *batch_id;
    const uint32_t *game_seed;
    const uint8_t *columns;
    size_t shard_stride;
    const char *directory;
    // from the Arrow-written template's footer: FileMetaData fields 1-2 (version, schema), 5 (key-value metadata), 7 (column orders) as
    // raw Thrift spans, and per leaf column its physical type and path_in_schema (components separated by '\0', terminated by "\0\0")
    const uint8_t *footer_head; size_t footer_head_len;
    const uint8_t *footer_kv; size_t footer_kv_len;
    const uint8_t *footer_orders; size_t footer_orders_len;
    const int32_t *leaf_type;
    const char *leaf_paths;
    int32_t n_leaves;
    // contract-v3 sidecar template (null: none)
    const char *const *side_body; // 4 pieces
    const char *const *side_full; // 5 pieces
    const char *side_directory;
};

static inline size_t int_planes(int k) { return 4 + 13 * (size_t)k; }
static inline size_t shard_image_bytes(int k, int gps) { return ((int_planes(k) * 4 + 2 + (size_t)k) * (size_t)gps + 63) & ~(size_t)63; }
static inline int index_width(int k) { int w = 1; while ((1 << w) < k) ++w; return w; }

struct ShardBuilder {
    const Job &job;
    Sink file;
    std::vector<uint8_t> tmp, hit_bits, def_bits, rep_lv, def_lv2, idx, enc;
    std::vector<int32_t> packed;
    std::vector<ColumnMeta> meta;
    std::vector<std::vector<std::string>> paths;
    std::vector<uint8_t> iota_plain;      // game_index: the same PLAIN payload in every shard
    std::vector<uint8_t> seat_dict;       // PLAIN dictionary page payload: "P1" .. "Pk"
    std::vector<uint8_t> status_dict;     // "completed", "safety_limit"
    explicit ShardBuilder(const Job &j) : job(j) {
        const char *p = j.leaf_paths;
        for (int i = 0; i < j.n_leaves; ++i) {
            std::vector<std::string> parts;
            while (*p) { parts.emplace_back(p); p += parts.back().size() + 1; }
            ++p;
            paths.push_back(std::move(parts));
        }
        iota_plain.resize((size_t)j.gps * 4);
        for (int32_t g = 0; g < j.gps; ++g) std::memcpy(&iota_plain[(size_t)g * 4], &g, 4);
        auto add_str = [](std::vector<uint8_t> &d, const std::string &s) {
            const uint32_t n = (uint32_t)s.size();
            d.insert(d.end(), reinterpret_cast<const uint8_t *>(&n), reinterpret_cast<const uint8_t *>(&n) + 4);
            d.insert(d.end(), s.begin(), s.end());
        };
        for (int s = 1; s <= j.k; ++s) add_str(seat_dict, "P" + std::to_string(s));
        add_str(status_dict, "completed");
        add_str(status_dict, "safety_limit");
        // repetition levels of seat_ranks: 0 at a row's first item, 1 at the others — the same for every shard
        BitPacker bp(rep_lv);
        for (int g = 0; g < j.gps; ++g)
            for (int s = 0; s < j.k; ++s) bp.add(s ? 1u : 0u, 1);
        bp.flush();
    }

    void begin_column(bool dictionary) {
        ColumnMeta m{};
        m.dictionary = dictionary;
        m.delta = false;
        m.dictionary_page_offset = dictionary ? (int64_t)file.size() : -1;
        m.data_page_offset = (int64_t)file.size();
        m.total_size = (int64_t)file.size(); // start; turned into a size by end_column
        meta.push_back(m);
    }
    void end_column(int64_t num_values) {
        ColumnMeta &m = meta.back();
        m.num_values = num_values;
        m.total_size = (int64_t)file.size() - m.total_size;
    }
    void dictionary_page(const void *payload, size_t n, uint32_t entries) {
        page_header(file, true, (uint32_t)n, entries, ENC_PLAIN);
        file.raw(payload, n);
        meta.back().data_page_offset = (int64_t)file.size();
    }
    // hybrid section of one bit-packed run: [4-byte length] header bytes — the V1 level framing — or bare (dictionary indices)
    void hybrid_bitpacked(Sink &s, const uint8_t *bytes, size_t n_bytes, uint64_t n_values, int width, bool length_prefix) {
        const uint64_t groups = (n_values + 7) / 8;
        const size_t want = (size_t)groups * (size_t)width; // bytes of `groups` groups of 8 values
        const size_t total = varint_len(groups << 1 | 1) + want;
        if (length_prefix) s.le32((uint32_t)total);
        s.varint(groups << 1 | 1);
        s.raw(bytes, n_bytes < want ? n_bytes : want);
        for (size_t i = n_bytes; i < want; ++i) s.put(0);
    }
    static size_t hybrid_bitpacked_size(uint64_t n_values, int width, bool length_prefix) {
        const uint64_t groups = (n_values + 7) / 8;
        return (length_prefix ? 4 : 0) + varint_len(groups << 1 | 1) + (size_t)groups * (size_t)width;
    }

    // -- column kinds --
    void plain_required(const void *values, size_t n_bytes, uint32_t n_values) {
        begin_column(false);
        page_header(file, false, (uint32_t)n_bytes, n_values, ENC_PLAIN);
        file.raw(values, n_bytes);
        end_column(n_values);
    }
    void constant(const void *value, size_t width, uint32_t n_values) { // one-entry dictionary + one RLE run of index 0
        begin_column(true);
        dictionary_page(value, width, 1);
        const size_t payload = 1 + varint_len((uint64_t)n_values << 1) + 1;
        page_header(file, false, (uint32_t)payload, n_values, ENC_RLE_DICTIONARY);
        file.put(1);                              // bit width
        file.varint((uint64_t)n_values << 1);     // RLE run of n_values ...
        file.put(0);                              // ... times index 0
        end_column(n_values);
    }
    void boolean_plain(const std::vector<uint8_t> &bits, uint32_t n_values) {
        plain_required(bits.data(), (n_values + 7) / 8, n_values);
    }
    // int32 columns of a row shard: DELTA_BINARY_PACKED (see delta_pack_int32).  `status` != null: rows with status[g] != 0 are null.
    void delta_int32(const int32_t *dense, const uint8_t *status, uint32_t n, uint32_t n_valid) {
        enc.clear();
        if (!status || n_valid == n) delta_pack_int32(enc, dense, n);
        else {
            packed.resize(n_valid);
            uint32_t at = 0;
            for (uint32_t g = 0; g < n; ++g)
                if (!status[g]) packed[at++] = dense[g];
            delta_pack_int32(enc, packed.data(), n_valid);
        }
        begin_column(false);
        meta.back().delta = true;
        const size_t payload = (status ? hybrid_bitpacked_size(n, 1, true) : 0) + enc.size();
        page_header(file, false, (uint32_t)payload, n, ENC_DELTA_BINARY_PACKED);
        if (status) hybrid_bitpacked(file, def_bits.data(), def_bits.size(), n, 1, true);
        file.raw(enc.data(), enc.size());
        end_column(n);
    }
    void status_column(uint32_t n) { // termination_status: dictionary {completed, safety_limit}, index = the hit bit
        begin_column(true);
        dictionary_page(status_dict.data(), status_dict.size(), 2);
        const size_t payload = 1 + hybrid_bitpacked_size(n, 1, false);
        page_header(file, false, (uint32_t)payload, n, ENC_RLE_DICTIONARY);
        file.put(1);
        hybrid_bitpacked(file, hit_bits.data(), hit_bits.size(), n, 1, false);
        end_column(n);
    }
    void winner_seat_column(const uint8_t *winner, const uint8_t *status, uint32_t n, uint32_t n_valid) {
        const int w = index_width(job.k);
        idx.clear();
        BitPacker bp(idx);
        for (uint32_t g = 0; g < n; ++g)
            if (!status[g]) bp.add(winner[g], w);
        bp.flush();
        begin_column(true);
        dictionary_page(seat_dict.data(), seat_dict.size(), (uint32_t)job.k);
        const size_t payload = hybrid_bitpacked_size(n, 1, true) + 1 + (n_valid ? hybrid_bitpacked_size(n_valid, w, false) : 0);
        page_header(file, false, (uint32_t)payload, n, ENC_RLE_DICTIONARY);
        hybrid_bitpacked(file, def_bits.data(), def_bits.size(), n, 1, true);
        file.put((uint8_t)w);
        if (n_valid) hybrid_bitpacked(file, idx.data(), idx.size(), n_valid, w, false);
        end_column(n);
    }
    void seat_ranks_column(const uint8_t *order, const uint8_t *status, uint32_t n, uint32_t n_valid) {
        const int w = index_width(job.k), k = job.k;
        const uint64_t items = (uint64_t)n * (uint64_t)k, valid_items = (uint64_t)n_valid * (uint64_t)k;
        def_lv2.clear();
        idx.clear();
        {
            BitPacker lv(def_lv2), ix(idx);
            for (uint32_t g = 0; g < n; ++g) {
                const uint32_t d = status[g] ? 1u : 2u; // list present; item null (1) or present (2)
                for (int s = 0; s < k; ++s) lv.add(d, 2);
                if (!status[g])
                    for (int s = 0; s < k; ++s) ix.add(order[(size_t)g * k + s], w);
            }
            lv.flush();
            ix.flush();
        }
        begin_column(true);
        dictionary_page(seat_dict.data(), seat_dict.size(), (uint32_t)k);
        const size_t payload = hybrid_bitpacked_size(items, 1, true) + hybrid_bitpacked_size(items, 2, true) + 1 +
                               (valid_items ? hybrid_bitpacked_size(valid_items, w, false) : 0);
        page_header(file, false, (uint32_t)payload, (uint32_t)items, ENC_RLE_DICTIONARY);
        hybrid_bitpacked(file, rep_lv.data(), rep_lv.size(), items, 1, true);
        hybrid_bitpacked(file, def_lv2.data(), def_lv2.size(), items, 2, true);
        file.put((uint8_t)w);
        if (valid_items) hybrid_bitpacked(file, idx.data(), idx.size(), valid_items, w, false);
        end_column((int64_t)items);
    }

    // the whole file of shard `i` into `file`
    void build(int32_t i) {
        const int k = job.k;
        const uint32_t n = (uint32_t)job.gps;
        const uint8_t *image = job.columns + (size_t)i * job.shard_stride;
        const int32_t *planes = reinterpret_cast<const int32_t *>(image);
        const uint8_t *status = image + int_planes(k) * 4 * (size_t)n, *winner = status + n, *order = winner + n;
        auto plane = [&](size_t c) { return planes + c * (size_t)n; };
        file.b.clear();
        meta.clear();
        file.raw("PAR1", 4);
        // one bitmap per shard: hit (1 = safety limit) and its complement (1 = every nullable field present)
        hit_bits.assign((n + 7) / 8, 0);
        uint32_t n_hit = 0;
        for (uint32_t g = 0; g < n; ++g)
            if (status[g]) { hit_bits[g >> 3] |= (uint8_t)(1u << (g & 7)); ++n_hit; }
        def_bits.resize(hit_bits.size());
        for (size_t b = 0; b < hit_bits.size(); ++b) def_bits[b] = (uint8_t)~hit_bits[b];
        if (n & 7) def_bits.back() &= (uint8_t)((1u << (n & 7)) - 1u);
        const uint32_t n_valid = n - n_hit;
        const int64_t root = (int64_t)job.root_seed, sh = job.shuffle_index[i], seed = job.shuffle_seed[i];
        const int32_t kk = k, batch = job.batch_id[i], two = 2, ns = job.rng_purpose_namespace;
        constant(&root, 8, n);                 // root_seed
        constant(&kk, 4, n);                   // k
        constant(&sh, 8, n);                   // shuffle_index
        delta_int32(reinterpret_cast<const int32_t *>(iota_plain.data()), nullptr, n, n); // game_index
        constant(&batch, 4, n);                // deterministic_batch_id
        constant(&seed, 8, n);                 // shuffle_seed
        status_column(n);                      // termination_status
        boolean_plain(hit_bits, n);            // hit_safety_limit
        constant(&two, 4, n);                  // outcome_schema_version
        winner_seat_column(winner, status, n, n_valid);
        delta_int32(plane(0), status, n, n_valid); // winner_strategy
        tmp.resize((size_t)n * 8);             // game_seed: uint32 fingerprints as int64
        for (uint32_t g = 0; g < n; ++g) { const int64_t v = job.game_seed[(size_t)i * n + g]; std::memcpy(&tmp[(size_t)g * 8], &v, 8); }
        plain_required(tmp.data(), tmp.size(), n);
        constant(&two, 4, n);                  // rng_scheme_version
        constant(&ns, 4, n);                   // rng_purpose_namespace
        seat_ranks_column(order, status, n, n_valid);
        delta_int32(plane(1), status, n, n_valid); // winning_score
        delta_int32(plane(2), status, n, n_valid); // victory_margin
        delta_int32(plane(3), nullptr, n, n);      // n_rounds
        for (int s = 0; s < k; ++s) {
            const size_t base = 4 + 13 * (size_t)s;
            for (int f = 0; f < 13; ++f) { // score farkles rolls highest_turn strategy rank loss_margin sf_uses sf_dice so_uses so_dice hot_dice n_turns
                delta_int32(plane(base + f), f == 5 || f == 6 ? status : nullptr, n, n_valid);
            }
            boolean_plain(hit_bits, n);        // P#_hit_max_rounds: every seat of a safety-limit game is flagged (engine.py:485-489)
        }
        footer(n);
    }

    void footer(uint32_t n_rows) {
        const size_t start = file.size();
        file.raw(job.footer_head, job.footer_head_len); // 1: version, 2: schema
        file.i64f(1, n_rows);                           // 3: num_rows
        file.field(1, T_LIST);                          // 4: row_groups
        file.list(1, T_STRUCT);
        file.field(1, T_LIST);                          //   1: columns
        file.list((int)meta.size(), T_STRUCT);
        int64_t total = 0;
        for (size_t c = 0; c < meta.size(); ++c) {
            const ColumnMeta &m = meta[c];
            total += m.total_size;
            file.i64f(2, 0);                            //     2: file_offset (deprecated; Arrow writes 0)
            file.field(1, T_STRUCT);                    //     3: meta_data
            file.i32f(1, job.leaf_type[c]);             //       1: type
            file.field(1, T_LIST);                      //       2: encodings
            if (m.dictionary) { file.list(3, T_I32); file.zz(ENC_PLAIN); file.zz(ENC_RLE); file.zz(ENC_RLE_DICTIONARY); }
            else if (m.delta) { file.list(2, T_I32); file.zz(ENC_RLE); file.zz(ENC_DELTA_BINARY_PACKED); }
            else { file.list(2, T_I32); file.zz(ENC_PLAIN); file.zz(ENC_RLE); }
            file.field(1, T_LIST);                      //       3: path_in_schema
            file.list((int)paths[c].size(), T_BINARY);
            for (const std::string &part : paths[c]) file.binary(part.data(), part.size());
            file.i32f(1, 0);                            //       4: codec UNCOMPRESSED
            file.i64f(1, m.num_values);                 //       5
            file.i64f(1, m.total_size);                 //       6: total_uncompressed_size
            file.i64f(1, m.total_size);                 //       7: total_compressed_size
            file.i64f(2, m.data_page_offset);           //       9
            if (m.dictionary) file.i64f(2, m.dictionary_page_offset); // 11
            file.put(0);
            file.put(0);
        }
        file.i64f(1, total);                            //   2: total_byte_size
        file.i64f(1, n_rows);                           //   3: num_rows
        file.put(0);
        file.raw(job.footer_kv, job.footer_kv_len);     // 5: key_value_metadata (ARROW:schema)
        static const char created_by[] = "farkle_ii_amd shard writer version 7 (build 0)";
        file.field(1, T_BINARY);                        // 6: created_by
        file.binary(created_by, sizeof(created_by) - 1);
        file.raw(job.footer_orders, job.footer_orders_len); // 7: column_orders
        file.put(0);
        file.le32((uint32_t)(file.size() - start));
        file.raw("PAR1", 4);
    }
};

static bool write_all(int fd, const uint8_t *data, size_t n) {
    size_t done = 0;
    while (done < n) {
        const ssize_t w = ::write(fd, data + done, n - done);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        done += (size_t)w;
    }
    return true;
}

// A file that appears under its name complete or not at all.  First choice: an unnamed file in the target directory (O_TMPFILE) linked
// into place — ONE operation on the directory, whose lock the writer's threads all contend for, instead of create + rename.  Where the
// file system has no O_TMPFILE, /proc is not mounted, or the name exists already (a replayed batch): "<name>.tmp" + rename.
static bool write_file(const std::string &path, const uint8_t *data, size_t n, bool atomic, std::string &err) {
    static std::atomic<bool> unnamed_files{true};
    if (atomic && unnamed_files.load(std::memory_order_relaxed)) {
        const size_t slash = path.rfind('/');
        const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash ? slash : 1);
        const int fd = ::open(dir.c_str(), O_TMPFILE | O_WRONLY | O_CLOEXEC, 0644);
        if (fd < 0) {
            if (errno != ENOENT && errno != EACCES) unnamed_files.store(false); // not supported here (a missing directory is reported below)
        } else {
            bool linked = false;
            if (write_all(fd, data, n)) {
                char proc[64];
                std::snprintf(proc, sizeof proc, "/proc/self/fd/%d", fd);
                linked = ::linkat(AT_FDCWD, proc, AT_FDCWD, path.c_str(), AT_SYMLINK_FOLLOW) == 0;
                if (!linked && errno != EEXIST) unnamed_files.store(false);
            }
            ::close(fd);
            if (linked) return true;
        }
    }
    const std::string staging = atomic ? path + ".tmp" : path;
    const int fd = ::open(staging.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { err = "open " + staging + ": " + std::strerror(errno); return false; }
    size_t done = 0;
    while (done < n) {
        const ssize_t w = ::write(fd, data + done, n - done);
        if (w < 0) {
            if (errno == EINTR) continue;
            err = "write " + staging + ": " + std::strerror(errno);
            ::close(fd);
            return false;
        }
        done += (size_t)w;
    }
    if (::close(fd) != 0) { err = "close " + staging + ": " + std::strerror(errno); return false; }
    if (atomic && ::rename(staging.c_str(), path.c_str()) != 0) { err = "rename " + staging + ": " + std::strerror(errno); return false; }
    return true;
}

// The writer's host threads, kept between calls (a production sweep makes 41 calls of ~10 ms: creating and joining 15 threads for each was
// 1.5 ms of it).  Never destroyed: the threads are detached and sleep on the condition variable until the process ends.
class WorkerPool {
    std::mutex run_lock, m;
    std::condition_variable cv_work, cv_done;
    const std::function<void()> *fn = nullptr;
    int alive = 0, want = 0, started = 0, finished = 0;
    uint64_t generation = 0;
    pid_t owner = 0;

    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return generation != seen; });
            seen = generation;
            if (started >= want) continue; // enough threads have taken this job
            ++started;
            const std::function<void()> *f = fn;
            lk.unlock();
            (*f)();
            lk.lock();
            if (++finished == want) cv_done.notify_all();
        }
    }

public:
    // f() on n_threads threads at once (the caller's included); returns when all have returned.  One job at a time.
    void run(int n_threads, const std::function<void()> &f) {
        std::lock_guard<std::mutex> one(run_lock);
        {
            std::unique_lock<std::mutex> lk(m);
            if (owner != getpid()) { alive = 0; owner = getpid(); } // (a forked child has none of the parent's threads)
            while (alive < n_threads - 1) {
                std::thread([this] { worker(); }).detach();
                ++alive;
            }
            fn = &f;
            want = n_threads - 1;
            started = finished = 0;
            ++generation;
        }
        cv_work.notify_all();
        f();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return finished == want; });
        fn = nullptr;
    }
    static WorkerPool &instance() {
        static WorkerPool *pool = new WorkerPool; // (leaked on purpose, see above)
        return *pool;
    }
};

// returns 0, or -1 with `error` set
static int write_shards(const Job &job, int64_t *byte_length, uint8_t *sha, uint8_t *side_sha, std::string &error) {
    if (job.k < 1 || job.gps < 1 || job.n_shuffles < 0 || job.n_leaves != 18 + 14 * job.k) { error = "bad shard job"; return -1; }
    if (job.shard_stride < shard_image_bytes(job.k, job.gps)) { error = "shard_stride is smaller than a shard image"; return -1; }
    std::atomic<int32_t> next{0};
    std::atomic<bool> failed{false};
    std::string first_error;
    std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
    const int n_threads = std::max(1, std::min<int>(job.threads, job.n_shuffles));
    const bool timing = getenv("FK_SHARD_WRITER_TIMING") != nullptr; // diagnostics: where a writer thread's time goes (stderr, per call)
    std::atomic<long long> ns_build{0}, ns_sha{0}, ns_write{0};
    auto now = []() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto work = [&]() {
        ShardBuilder sb[2] = {ShardBuilder(job), ShardBuilder(job)}; // two files per turn: their SHA-256 chains run in lockstep (sha256_pair)
        std::string side, err;
        char name[96], hex[65];
        auto publish = [&](int32_t i, const Sink &file) -> bool { // the file, and (contract v3) its sidecar from the template text
            std::snprintf(name, sizeof name, "rows_%llu_%dp_%012lld.parquet", (unsigned long long)job.root_seed, job.k, (long long)job.shuffle_index[i]);
            const std::string path = std::string(job.directory) + "/" + name;
            if (!write_file(path, file.b.data(), file.size(), job.atomic != 0, err)) return false;
            if (!job.side_body) return true;
            hex32(sha + (size_t)i * 32, hex);
            const std::string len = std::to_string(file.size()), digest = std::string("\"") + hex + "\"", rel = std::string("\"") + job.side_directory + name + "\"";
            side.clear();
            side.append(job.side_body[0]).append(len).append(job.side_body[1]).append(digest).append(job.side_body[2]).append(rel).append(job.side_body[3]);
            uint8_t d[32];
            sha256(reinterpret_cast<const uint8_t *>(side.data()), side.size(), d);
            hex32(d, hex);
            side.clear();
            side.append(job.side_full[0]).append(len).append(job.side_full[1]).append(digest).append(job.side_full[2]).append(rel).append(job.side_full[3])
                .append("\"").append(hex).append("\"").append(job.side_full[4]).append("\n");
            sha256(reinterpret_cast<const uint8_t *>(side.data()), side.size(), side_sha + (size_t)i * 32);
            return write_file(path + ".sidecar.json", reinterpret_cast<const uint8_t *>(side.data()), side.size(), false, err);
        };
        for (;;) {
            const int32_t first = next.fetch_add(2);
            if (first >= job.n_shuffles || failed.load()) return;
            const int32_t n_here = first + 1 < job.n_shuffles ? 2 : 1;
            const long long t0 = timing ? now() : 0;
            for (int32_t j = 0; j < n_here; ++j) {
                sb[j].build(first + j);
                byte_length[first + j] = (int64_t)sb[j].file.size();
            }
            const long long t1 = timing ? now() : 0;
            if (n_here == 2)
                sha256_pair(sb[0].file.b.data(), sb[0].file.size(), sha + (size_t)first * 32, sb[1].file.b.data(), sb[1].file.size(), sha + (size_t)(first + 1) * 32);
            else
                sha256(sb[0].file.b.data(), sb[0].file.size(), sha + (size_t)first * 32);
            const long long t2 = timing ? now() : 0;
            bool ok = true;
            for (int32_t j = 0; j < n_here && ok; ++j) ok = publish(first + j, sb[j].file);
            if (timing) {
                ns_build += t1 - t0;
                ns_sha += t2 - t1;
                ns_write += now() - t2;
            }
            if (!ok) {
                failed.store(true);
                while (err_lock.test_and_set()) {}
                if (first_error.empty()) first_error = err;
                err_lock.clear();
                return;
            }
        }
    };
    const long long wall0 = timing ? now() : 0;
    WorkerPool::instance().run(n_threads, work);
    if (timing)
        std::fprintf(stderr, "[fk shards] %d shards, %d threads: build %.1f ms, sha256 %.1f ms, write %.1f ms (thread time, summed); wall %.1f ms, began at %.1f ms\n",
                     job.n_shuffles, n_threads, ns_build.load() / 1e6, ns_sha.load() / 1e6, ns_write.load() / 1e6, (now() - wall0) / 1e6, (wall0 % 100000000000ll) / 1e6);
    if (failed.load()) { error = first_error; return -1; }
    return 0;
}

} // namespace fksw
