// rans_host.cpp -- host range coder of the hot path (product code).
//
// Replaces compressai.ans.BufferedRansEncoder / RansDecoder as called from
// image_model.py:221,253-254,273-274,288 (and inside EntropyBottleneck.compress/decompress,
// image_model.py:206-207,260).  Bitstream format = compressai 1.2.1 rans_interface.cpp over
// ryg_rans rans64.h: 64-bit state, L = 2^31, 32-bit renormalisation words,
// 16-bit probabilities, out-of-range symbols escaped through 4-bit bypass nibbles; symbols are
// pushed in order and coded in reverse; the stream is the little-endian word sequence.
//
// Implementation notes (differs from the straightforward form on purpose; bytes are identical by construction
// and checked against the oracle's plain restatement in tests/test_cabi_host.py):
//   * the pending list stores one packed 32-bit entry per coded item (start:16 | (range-1):16,
//     bypass items tagged by an out-of-band bitmap) -- half the memory traffic of a struct list;
//   * ENCODE never divides: x / freq is a 64x64->128 multiply by a per-frequency reciprocal (the ryg_rans
//     Rans64EncSymbolInit construction, exact for every state the coder can reach), table built once per process;
//   * DECODE finds the symbol through a per-row 256-entry lookup table indexed by the top 8 bits of the 16-bit
//     cumulative frequency (first candidate symbol), then a short forward scan -- rows reach 3133 entries for
//     sigma = 256 but the probability mass sits in a few symbols, so the scan is ~1 step; the LUT belongs to the
//     caller's dsvc_cdf_table (dsvc_cdf_build_lut); without one the decoder binary-searches.
#include <cstdint>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <vector>
#include <algorithm>
#include "common.h"

namespace {
constexpr uint32_t kPrecision = 16;
constexpr uint32_t kBypassBits = 4;
constexpr int32_t kBypassMax = 15;
constexpr uint64_t kRansL = 1ull << 31;

// ---- division-free encode: q = mulhi(x, rcp) >> shift == x / freq for all x < 2^63 (ryg_rans rans64.h) ----------
struct Rcp { uint64_t rcp; uint32_t shift; };
static Rcp g_rcp[65537];
static bool g_rcp_ready = false;
static void init_rcp() {
    if (g_rcp_ready) return;
    g_rcp[0] = Rcp{0, 0};
    g_rcp[1] = Rcp{~0ull, 0};               // freq 1: handled separately (q = x)
    for (uint32_t freq = 2; freq <= 65536; ++freq) {
        uint32_t shift = 0;
        while (freq > (1u << shift)) ++shift;
        // ((2^(shift+63)) + freq-1) / freq by two 64-bit divides
        uint64_t x0 = freq - 1, x1 = 1ull << (shift + 31);
        const uint64_t t1 = x1 / freq;
        x0 += (x1 % freq) << 32;
        const uint64_t t0 = x0 / freq;
        g_rcp[freq] = Rcp{t0 + (t1 << 32), shift - 1};
    }
    g_rcp_ready = true;
}
static const bool g_rcp_init_at_load = (init_rcp(), true);   // built before any thread can encode
static inline uint64_t div_by(uint64_t x, uint32_t freq) {
    if (freq == 1) return x;
    const Rcp& r = g_rcp[freq];
    return (uint64_t)(((unsigned __int128)x * r.rcp) >> 64) >> r.shift;
}

struct Encoder {
    std::vector<uint32_t> items;       // start | (range-1) << 16; only the first `count` entries are live
    size_t count = 0;
    std::vector<uint64_t> bypass_at;   // ascending positions of the items that are bypass nibbles (escapes are rare)
    std::vector<uint32_t> words;       // scratch for flush()
    int64_t nbytes = -1;               // valid after a size query
    int64_t first = 0;
};

struct Decoder {
    std::vector<uint32_t> words;
    size_t pos = 2;
    uint64_t x = 0;
    inline void renorm() {
        if (x < kRansL) {
            const uint32_t w = pos < words.size() ? words[pos] : 0u;
            ++pos;
            x = (x << 32) | w;
        }
    }
    inline uint32_t bits() {
        const uint32_t v = (uint32_t)(x & kBypassMax);
        x >>= kBypassBits;
        renorm();
        return v;
    }
};

inline void push_bypass(Encoder* e, uint32_t nibble) {      // caller guarantees capacity
    e->bypass_at.push_back((uint64_t)e->count);
    e->items[e->count++] = nibble;
}

bool table_ok(const dsvc_cdf_table* t) {
    return t && t->cdf && t->cdf_length && t->offset && t->rows > 0 && t->stride > 1;
}

// Every row must be a proper 16-bit CDF over its cdf_length entries: row[0] = 0, strictly increasing (no zero-frequency
// symbol: the encoder's (freq-1)<<16 packing and the decoder's forward scan both rely on it), row[len-1] = 65536.  Tables
// come from checkpoint buffers (entropy_models.host_tables), so they are validated, not trusted; a few hundred thousand
// integer compares per call against >= 10^5 coded symbols.
int rows_valid(const dsvc_cdf_table* t, const char* who) {
    for (int r = 0; r < t->rows; ++r) {
        const int32_t len = t->cdf_length[r];
        DSVC_REQUIRE(len >= 2 && len <= t->stride, "%s: cdf_length[%d] = %d invalid (stride %d)", who, r, len, (int)t->stride);
        const int32_t* row = t->cdf + (int64_t)r * t->stride;
        DSVC_REQUIRE(row[0] == 0 && row[len - 1] == (1 << kPrecision), "%s: cdf row %d does not span [0, 65536]", who, r);
        int bad = 0;
        for (int32_t i = 1; i < len; ++i) bad |= row[i] <= row[i - 1];
        DSVC_REQUIRE(!bad, "%s: cdf row %d is not strictly increasing (zero-frequency symbol)", who, r);
    }
    return DSVC_OK;
}
}  // namespace

extern "C" int dsvc_rans_encoder_create(void** enc) {
    DSVC_REQUIRE(enc != nullptr, "rans_encoder_create: null out pointer");
    init_rcp();
    *enc = new Encoder();
    return DSVC_OK;
}

// ---- which LUTs came from dsvc_cdf_build_lut ------------------------------------------------------------------------------
// A table that carries a LUT skips the per-call row validation -- but only if THIS library built that LUT for THIS table: the
// builder registers (lut pointer -> fingerprint of the table it validated); the coder looks it up (one mutex-free read of a small
// array per call) and falls back to validating the rows when the pointer is unknown or the table behind it has changed.
namespace {
struct LutReg { const uint16_t* lut; uint64_t fp; };
constexpr int kMaxLuts = 256;        // 4 GOP lanes x (2 P-frame codecs + 1 I-frame codec) x 2 entropy models per rank = 24 live tables
LutReg g_luts[kMaxLuts];
int g_nluts = 0;
std::shared_mutex g_lut_mu;          // readers: one shared lock per coder call; writers: build / forget
// Fingerprint of the table a LUT was built for: its address and geometry, every row length, and EIGHT evenly spaced entries of every row
// (+ the last one).  Contract (include/dsvc.h): a table is not modified while a LUT built for it is in use -- rebuild the LUT after
// any change (the Python side does: entropy_models._EntropyModel.tables_changed drops the CdfTables object); the samples catch a buffer
// that was recycled for another table.  Rows whose length is out of range make the table untrusted (nothing is read through them).
bool table_fp(const dsvc_cdf_table* t, uint64_t* out) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)(uintptr_t)t->cdf); mix((uint64_t)t->rows); mix((uint64_t)t->stride);
    mix((uint64_t)(uintptr_t)t->cdf_length); mix((uint64_t)(uintptr_t)t->offset);
    for (int r = 0; r < t->rows; ++r) {
        const int32_t len = t->cdf_length[r];
        if (len < 2 || len > t->stride) return false;
        const int32_t* row = t->cdf + (int64_t)r * t->stride;
        mix((uint64_t)(uint32_t)len); mix((uint64_t)(uint32_t)t->offset[r]);
        for (int k = 0; k < 8; ++k) mix((uint64_t)(uint32_t)row[(int64_t)(len - 1) * k / 8]);
        mix((uint64_t)(uint32_t)row[len - 1]);
    }
    *out = h;
    return true;
}
bool lut_trusted(const dsvc_cdf_table* t) {
    if (!t->lut) return false;
    uint64_t fp;
    if (!table_fp(t, &fp)) return false;
    std::shared_lock<std::shared_mutex> lk(g_lut_mu);
    for (int i = 0; i < g_nluts; ++i)
        if (g_luts[i].lut == t->lut) return g_luts[i].fp == fp;
    return false;
}
}  // namespace

long long g_dsvc_lut_overflow = 0;      // LUTs that found the registry full (dsvc_get_counter("lut_registry_overflow")): their tables are validated per call

extern "C" int dsvc_cdf_forget_lut(const uint16_t* lut) {
    std::unique_lock<std::shared_mutex> lk(g_lut_mu);
    for (int i = 0; i < g_nluts; ++i)
        if (g_luts[i].lut == lut) { g_luts[i] = g_luts[--g_nluts]; return 1; }
    return 0;
}

extern "C" void dsvc_rans_encoder_destroy(void* enc) { delete (Encoder*)enc; }

extern "C" int dsvc_rans_encode_with_indexes(void* enc, const int32_t* symbols, const int32_t* indexes, int64_t n,
                                             const dsvc_cdf_table* t) {
    DSVC_REQUIRE(enc && symbols && indexes && n >= 0, "rans_encode: null argument");
    DSVC_REQUIRE(table_ok(t), "rans_encode: bad cdf table");
    Encoder* e = (Encoder*)enc;
    e->nbytes = -1;
    // validation hoisted out of the symbol loop: the table once, the indexes in one vectorisable pass
    // (a table that carries a LUT was validated row by row when dsvc_cdf_build_lut made it)
    if (!lut_trusted(t)) { if (int rc = rows_valid(t, "rans_encode")) return rc; }
    {
        uint32_t worst = 0;
        for (int64_t i = 0; i < n; ++i) worst |= (uint32_t)indexes[i] >= (uint32_t)t->rows ? 1u : 0u;
        if (worst)
            for (int64_t i = 0; i < n; ++i)
                DSVC_REQUIRE(indexes[i] >= 0 && indexes[i] < t->rows, "rans_encode: cdf index %d out of range at %lld", indexes[i], (long long)i);
    }
    constexpr size_t kSlack = 32;     // an escape appends at most 1 + 2 + 8 bypass items after its symbol
    if (e->items.size() < e->count + (size_t)n + kSlack) e->items.resize(e->count + (size_t)n + (size_t)n / 8 + 4 * kSlack);
    const int32_t* cdf = t->cdf;
    const int64_t stride = t->stride;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t* row = cdf + (int64_t)ci * stride;
        const int32_t max_value = t->cdf_length[ci] - 2;
        int32_t v = (int32_t)((uint32_t)symbols[i] - (uint32_t)t->offset[ci]);      // (wraps like the reference's int32 arithmetic, without UB)
        if (__builtin_expect((uint32_t)v < (uint32_t)max_value, 1)) {       // in-range symbol: one item, no escape
            const uint32_t start = (uint32_t)row[v];
            e->items[e->count++] = (start & 0xFFFFu) | (((uint32_t)row[v + 1] - start - 1u) << 16);
            continue;
        }
        uint32_t raw;
        if (v < 0) raw = (uint32_t)(-2 * (int64_t)v - 1);
        else raw = (uint32_t)(2 * (int64_t)(v - max_value));
        v = max_value;
        if (e->items.size() < e->count + (size_t)(n - i) + kSlack) e->items.resize(e->items.size() + (size_t)n / 8 + 4 * kSlack);
        {
            const uint32_t start = (uint32_t)row[v];
            e->items[e->count++] = (start & 0xFFFFu) | (((uint32_t)row[v + 1] - start - 1u) << 16);
        }
        int32_t nb = 0;
        while (nb < 8 && (raw >> (nb * kBypassBits)) != 0) ++nb;
        int32_t val = nb;
        while (val >= kBypassMax) { push_bypass(e, kBypassMax); val -= kBypassMax; }
        push_bypass(e, (uint32_t)val);
        for (int32_t j = 0; j < nb; ++j) push_bypass(e, (raw >> (j * kBypassBits)) & kBypassMax);
    }
    return DSVC_OK;
}

extern "C" int64_t dsvc_rans_flush(void* enc, uint8_t* out, int64_t cap) {
    if (!enc) { dsvc_set_error("rans_flush: null encoder"); return DSVC_E_ARG; }
    Encoder* e = (Encoder*)enc;
    if (e->nbytes < 0) {
        const size_t m = e->count;
        e->words.resize(m + 4);      // every item emits at most one word; +2 state words (+1: the speculative store below)
        uint32_t* words = e->words.data();
        const uint32_t* items = e->items.data();
        size_t p = e->words.size();
        uint64_t x = kRansL;
        size_t nbp = e->bypass_at.size();
        uint64_t next_bp = nbp ? e->bypass_at[nbp - 1] : ~0ull;
        for (size_t k = m; k-- > 0;) {
            const uint32_t it = items[k];
            if (__builtin_expect((uint64_t)k != next_bp, 1)) {
                const uint32_t start = it & 0xFFFFu, freq = (it >> 16) + 1;
                const uint64_t x_max = ((kRansL >> kPrecision) << 32) * (uint64_t)freq;
                // renormalise without a branch (a word leaves every ~32/bits symbols: badly predictable): the word is stored
                // speculatively below the write pointer and the pointer moves only when it counts
                const bool emit = x >= x_max;
                words[p - 1] = (uint32_t)x;
                p -= emit ? 1 : 0;
                x = emit ? x >> 32 : x;
                const uint64_t q = div_by(x, freq);                           // x / freq without a divide
                x = x + start + q * (uint64_t)((1u << kPrecision) - freq);     // == (q << 16) + x % freq + start
            } else {
                --nbp;
                next_bp = nbp ? e->bypass_at[nbp - 1] : ~0ull;
                const uint64_t x_max = ((kRansL >> 16) << 32) * (1ull << (16 - kBypassBits));
                if (x >= x_max) { words[--p] = (uint32_t)x; x >>= 32; }
                x = (x << kBypassBits) | (it & 0xFFFFu);
            }
        }
        p -= 2;
        words[p] = (uint32_t)x;
        words[p + 1] = (uint32_t)(x >> 32);
        e->first = (int64_t)p;
        e->nbytes = (int64_t)(e->words.size() - p) * 4;
    }
    if (!out) return e->nbytes;
    if (cap < e->nbytes) { dsvc_set_error("rans_flush: buffer %lld < %lld", (long long)cap, (long long)e->nbytes); return DSVC_E_RANGE; }
    std::memcpy(out, e->words.data() + e->first, (size_t)e->nbytes);  // x86-64: little endian == stream order
    const int64_t nb = e->nbytes;
    e->count = 0; e->bypass_at.clear(); e->nbytes = -1;
    return nb;
}

extern "C" int dsvc_cdf_build_lut(const dsvc_cdf_table* t, uint16_t* lut_out) {
    DSVC_REQUIRE(table_ok(t) && lut_out, "cdf_build_lut: bad argument");
    if (int rc = rows_valid(t, "cdf_build_lut")) return rc;
    {
        uint64_t fp = 0;
        const bool have_fp = table_fp(t, &fp);            // (rows_valid passed: always true)
        std::unique_lock<std::shared_mutex> lk(g_lut_mu);
        int slot = -1;
        for (int i = 0; i < g_nluts; ++i) if (g_luts[i].lut == lut_out) slot = i;
        if (slot < 0 && g_nluts < kMaxLuts) slot = g_nluts++;
        if (slot >= 0 && have_fp) g_luts[slot] = LutReg{lut_out, fp};
        else __atomic_fetch_add(&g_dsvc_lut_overflow, 1, __ATOMIC_RELAXED);     // not trusted: its table is validated on every call
    }
    for (int r = 0; r < t->rows; ++r) {
        const int32_t* row = t->cdf + (int64_t)r * t->stride;
        const int32_t len = t->cdf_length[r];
        int32_t s = 0;
        for (int b = 0; b < 256; ++b) {
            const uint32_t cf = (uint32_t)b << 8;
            while (s + 1 < len - 1 && (uint32_t)row[s + 1] <= cf) ++s;   // largest s with row[s] <= cf
            lut_out[(size_t)r * 256 + b] = (uint16_t)s;
        }
    }
    return DSVC_OK;
}

extern "C" int dsvc_rans_decoder_create(void** dec, const uint8_t* data, int64_t nbytes) {
    DSVC_REQUIRE(dec && data && nbytes >= 8 && nbytes % 4 == 0, "rans_decoder_create: need >= 8 bytes, multiple of 4 (got %lld)", (long long)nbytes);
    Decoder* d = new Decoder();
    d->words.resize((size_t)nbytes / 4);
    std::memcpy(d->words.data(), data, (size_t)nbytes);
    d->x = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    *dec = d;
    return DSVC_OK;
}

extern "C" void dsvc_rans_decoder_destroy(void* dec) { delete (Decoder*)dec; }

extern "C" int dsvc_rans_decode_stream(void* dec, const int32_t* indexes, int64_t n, const dsvc_cdf_table* t, int32_t* out) {
    DSVC_REQUIRE(dec && indexes && out && n >= 0, "rans_decode: null argument");
    DSVC_REQUIRE(table_ok(t), "rans_decode: bad cdf table");
    Decoder* d = (Decoder*)dec;
    {   // validation hoisted out of the symbol loop
        uint32_t worst = 0;
        for (int64_t i = 0; i < n; ++i) worst |= (uint32_t)indexes[i] >= (uint32_t)t->rows ? 1u : 0u;
        if (worst)
            for (int64_t i = 0; i < n; ++i)
                DSVC_REQUIRE(indexes[i] >= 0 && indexes[i] < t->rows, "rans_decode: cdf index %d out of range at %lld", indexes[i], (long long)i);
    }
    // a LUT that dsvc_cdf_build_lut did not make for THIS table is not used at all (its entries index the rows): binary search instead
    const bool trusted = lut_trusted(t);
    if (!trusted) { if (int rc = rows_valid(t, "rans_decode")) return rc; }
    const uint16_t* lut = trusted ? t->lut : nullptr;
    const int32_t* cdf = t->cdf;
    const int64_t stride = t->stride;
    const uint32_t* words = d->words.data();
    const size_t nw = d->words.size();
    uint64_t x = d->x;
    size_t pos = d->pos;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t* row = cdf + (int64_t)ci * stride;
        const int32_t max_value = t->cdf_length[ci] - 2;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        // largest s in [0, len-2] with row[s] <= cf  (row[0] = 0, row[len-1] = 65536 > cf):
        // start from the LUT's answer for the bucket's first value, scan forward (usually 0-1 steps)
        int32_t s;
        if (lut) {
            s = lut[(size_t)ci * 256 + (cf >> 8)];
            while (s < max_value && (uint32_t)row[s + 1] <= cf) ++s;     // bounded: never reads past row[len-1]
        } else {
            int32_t lo = 0, hi = max_value + 1;
            while (hi - lo > 1) {
                const int32_t mid = (lo + hi) >> 1;
                if ((uint32_t)row[mid] <= cf) lo = mid; else hi = mid;
            }
            s = lo;
        }
        const uint32_t start = (uint32_t)row[s], freq = (uint32_t)(row[s + 1] - row[s]);
        x = (uint64_t)freq * (x >> kPrecision) + cf - start;
        if (x < kRansL) {
            const uint32_t w = pos < nw ? words[pos] : 0u;
            ++pos;
            x = (x << 32) | w;
        }
        int32_t v = s;
        if (__builtin_expect(v == max_value, 0)) {          // escape: Exp-Golomb-like nibbles through the bypass path
            d->x = x; d->pos = pos;
            int32_t val = (int32_t)d->bits(), nb = val;
            while (val == kBypassMax) { val = (int32_t)d->bits(); nb += val; }
            uint32_t raw = 0;
            for (int32_t j = 0; j < nb && j < 8; ++j) raw |= d->bits() << (j * kBypassBits);
            v = (int32_t)(raw >> 1);
            v = (raw & 1u) ? -v - 1 : (int32_t)((uint32_t)v + (uint32_t)max_value);       // (a forged escape may wrap; never UB)
            x = d->x; pos = d->pos;
            if (pos > nw + 2) { dsvc_set_error("rans_decode: stream exhausted at symbol %lld", (long long)i); return DSVC_E_RANGE; }
        }
        out[i] = (int32_t)((uint32_t)v + (uint32_t)t->offset[ci]);
    }
    d->x = x; d->pos = pos;
    if (pos > nw + 2) { dsvc_set_error("rans_decode: stream exhausted within %lld symbols", (long long)n); return DSVC_E_RANGE; }
    return DSVC_OK;
}
