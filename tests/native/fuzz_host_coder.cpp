// fuzz_host_coder.cpp -- sanitizer harness for libdsvc's HOST functions (the range coder parses untrusted byte strings
// and takes tables from checkpoint buffers).  Test infrastructure: tests/test_cabi_host.py compiles this file TOGETHER WITH the product
// sources deepsvc_amd/csrc/rans_host.cpp + host_misc.cpp (-DDSVC_HOST_ONLY: no HIP runtime, no device entry points) under
// -fsanitize=address,undefined and runs it on the CPU.  Every case must end in a return code -- never in a sanitizer report:
//   1. round trips on random tables / symbols incl. escapes far outside the support (both signs, INT32 extremes);
//   2. TRUNCATED streams (every prefix length of a short stream, random cuts of long ones) and BIT-FLIPPED streams through the decoder;
//   3. FORGED tables: bad row lengths, non-monotone rows, rows not spanning [0, 65536], a LUT that was not built for the table
//      (random entries, up to 65535), a table edited after its LUT was built, out-of-range CDF indexes;
//   4. pmf_to_quantized_cdf on NaN / negative / all-zero / single-spike / denormal pmfs; weight packing of ragged segment lists.
// Deterministic (xorshift seed from argv[1], default 1); prints a one-line summary.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../include/dsvc.h"

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static uint32_t rnd_below(uint32_t n) { return (uint32_t)(rnd() % n); }

struct Table {
    std::vector<int32_t> cdf, len, off;
    std::vector<uint16_t> lut;
    int rows = 0, stride = 0;
    dsvc_cdf_table c{};
    void bind(bool with_lut) {
        c.cdf = cdf.data(); c.rows = rows; c.stride = stride; c.cdf_length = len.data(); c.offset = off.data();
        c.lut = with_lut ? lut.data() : nullptr;
    }
};

// a valid random table: rows x (support sizes 1 .. max_support), built through the product's own pmf_to_quantized_cdf
static bool make_table(Table& t, int rows, int max_support) {
    t.rows = rows; t.stride = max_support + 2;
    t.cdf.assign((size_t)rows * t.stride, 0); t.len.assign(rows, 0); t.off.assign(rows, 0);
    for (int r = 0; r < rows; ++r) {
        const int n = 1 + (int)rnd_below((uint32_t)max_support);         // pmf entries incl. the tail bin
        std::vector<float> pmf(n + 1);
        for (int i = 0; i <= n; ++i) pmf[i] = (float)(1 + rnd_below(1000)) * ((rnd() & 7) == 0 ? 50.f : 1.f);
        float sum = 0; for (float v : pmf) sum += v;
        for (float& v : pmf) v /= sum;
        std::vector<int32_t> row(n + 2);
        if (dsvc_pmf_to_quantized_cdf(pmf.data(), n + 1, 16, row.data()) != DSVC_OK) return false;
        memcpy(&t.cdf[(size_t)r * t.stride], row.data(), sizeof(int32_t) * (n + 2));
        t.len[r] = n + 2;
        t.off[r] = -(int32_t)rnd_below((uint32_t)n + 1);
    }
    t.lut.assign((size_t)rows * 256, 0);
    t.bind(false);
    if (dsvc_cdf_build_lut(&t.c, t.lut.data()) != DSVC_OK) return false;
    return true;
}

static int encode(const Table& t, bool with_lut, const std::vector<int32_t>& sym, const std::vector<int32_t>& idx, std::vector<uint8_t>& out) {
    Table u = t; u.bind(with_lut);
    if (with_lut) { u.c.lut = t.lut.data(); }       // the registered LUT belongs to t's buffers: bind to those
    dsvc_cdf_table c = t.c; c.lut = with_lut ? t.lut.data() : nullptr;
    void* e = nullptr;
    if (dsvc_rans_encoder_create(&e) != DSVC_OK) return -100;
    static const int32_t dummy = 0;          // n == 0: the API still wants non-null arrays
    int rc = dsvc_rans_encode_with_indexes(e, sym.empty() ? &dummy : sym.data(), idx.empty() ? &dummy : idx.data(), (int64_t)sym.size(), &c);
    if (rc == DSVC_OK) {
        const int64_t n = dsvc_rans_flush(e, nullptr, 0);
        if (n < 0) rc = (int)n;
        else { out.resize((size_t)n); const int64_t m = dsvc_rans_flush(e, out.data(), n); if (m != n) rc = -101; }
    }
    dsvc_rans_encoder_destroy(e);
    return rc;
}

static int decode(const dsvc_cdf_table& c, const uint8_t* data, size_t nbytes, const std::vector<int32_t>& idx, std::vector<int32_t>& out) {
    void* d = nullptr;
    int rc = dsvc_rans_decoder_create(&d, data, (int64_t)nbytes);
    if (rc != DSVC_OK) return rc;
    out.assign(idx.size(), 0);
    static int32_t dummy = 0;
    rc = dsvc_rans_decode_stream(d, idx.empty() ? &dummy : idx.data(), (int64_t)idx.size(), &c, out.empty() ? &dummy : out.data());
    dsvc_rans_decoder_destroy(d);
    return rc;
}

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: %s (%s)\n", __LINE__, msg, dsvc_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc > 1) g_s ^= strtoull(argv[1], nullptr, 10) * 0xD6E8FEB86659FD93ull;
    long n_round = 0, n_trunc = 0, n_flip = 0, n_forged = 0, n_rejected = 0;
    // ---- 1. round trips -------------------------------------------------------------------------------------
    for (int iter = 0; iter < 40; ++iter) {
        Table t;
        CHECK(make_table(t, 1 + (int)rnd_below(12), 2 + (int)rnd_below(iter < 30 ? 40 : 3000)), "make_table");
        const size_t n = 1 + rnd_below(iter < 30 ? 3000 : 40000);
        std::vector<int32_t> sym(n), idx(n);
        for (size_t i = 0; i < n; ++i) {
            idx[i] = (int32_t)rnd_below((uint32_t)t.rows);
            const int32_t support = t.len[idx[i]] - 2;
            const uint32_t k = rnd_below(100);
            if (k < 90) sym[i] = t.off[idx[i]] + (int32_t)rnd_below((uint32_t)support + 1);            // in range (incl. the first escape value)
            else if (k < 96) sym[i] = t.off[idx[i]] + (int32_t)rnd_below(2000) - 1000;                 // moderate escapes, both signs
            else if (k < 98) sym[i] = (int32_t)(rnd() & 0x3FFFFFFF) * ((rnd() & 1) ? 1 : -1);          // huge escapes
            else sym[i] = (rnd() & 1) ? INT32_MAX : INT32_MIN;                                         // arithmetic extremes: must not be UB
        }
        for (int with_lut = 0; with_lut < 2; ++with_lut) {
            std::vector<uint8_t> bytes;
            CHECK(encode(t, with_lut != 0, sym, idx, bytes) == DSVC_OK, "encode");
            dsvc_cdf_table c = t.c; c.lut = with_lut ? t.lut.data() : nullptr;
            std::vector<int32_t> out;
            CHECK(decode(c, bytes.data(), bytes.size(), idx, out) == DSVC_OK, "decode");
            for (size_t i = 0; i < n; ++i) {
                // escapes whose magnitude exceeds what the format's 8 nibbles carry wrap (as compressai's int32 arithmetic does): compare
                // only symbols within +-2^28 of the support
                const int64_t v = (int64_t)sym[i] - t.off[idx[i]];
                if (v > -(1ll << 28) && v < (1ll << 28)) CHECK(out[i] == sym[i], "round trip symbol");
            }
            ++n_round;
            // ---- 2. truncated and bit-flipped streams ----
            const int cuts = bytes.size() <= 64 ? (int)bytes.size() / 4 : 24;
            for (int k = 0; k < cuts; ++k) {
                size_t keep = bytes.size() <= 64 ? (size_t)k * 4 : (size_t)rnd_below((uint32_t)bytes.size() / 4) * 4;
                std::vector<uint8_t> cut(bytes.begin(), bytes.begin() + keep);     // exact-size heap block: ASan sees any over-read
                std::vector<int32_t> o2;
                const int rc = decode(c, cut.data(), cut.size(), idx, o2);
                CHECK(rc == DSVC_OK || rc == DSVC_E_ARG || rc == DSVC_E_RANGE, "truncated stream: unexpected return code");
                n_rejected += rc != DSVC_OK; ++n_trunc;
            }
            {   // odd lengths are refused
                std::vector<uint8_t> odd(bytes.begin(), bytes.end()); odd.push_back(0);
                std::vector<int32_t> o2;
                CHECK(decode(c, odd.data(), odd.size(), idx, o2) == DSVC_E_ARG, "stream length not a multiple of 4 accepted");
            }
            for (int k = 0; k < 24; ++k) {
                std::vector<uint8_t> bad(bytes);
                const int flips = 1 + (int)rnd_below(8);
                for (int f = 0; f < flips; ++f) bad[rnd_below((uint32_t)bad.size())] ^= (uint8_t)(1u << rnd_below(8));
                if (k % 6 == 5) for (auto& b : bad) b = 0xFF;              // all-ones: endless 15-nibbles in the escape path
                std::vector<int32_t> o2;
                const int rc = decode(c, bad.data(), bad.size(), idx, o2);
                CHECK(rc == DSVC_OK || rc == DSVC_E_RANGE, "corrupted stream: unexpected return code");
                ++n_flip;
            }
        }
        // ---- 3. forged tables ----
        std::vector<uint8_t> bytes;
        CHECK(encode(t, true, sym, idx, bytes) == DSVC_OK, "encode (for the forged-table cases)");
        std::vector<int32_t> out;
        {   // a LUT the library did not build: random entries.  Must be ignored (binary search), result still correct
            std::vector<uint16_t> fake((size_t)t.rows * 256);
            for (auto& v : fake) v = (uint16_t)rnd();
            dsvc_cdf_table c = t.c; c.lut = fake.data();
            CHECK(decode(c, bytes.data(), bytes.size(), idx, out) == DSVC_OK, "decode with a foreign LUT");
            for (size_t i = 0; i < n; ++i) { const int64_t v = (int64_t)sym[i] - t.off[idx[i]]; if (v > -(1ll << 28) && v < (1ll << 28)) CHECK(out[i] == sym[i], "foreign LUT changed the result"); }
            ++n_forged;
        }
        {   // table edited after its LUT was built (same buffers): trust is revoked, the broken row is refused
            Table u = t; u.bind(false);
            CHECK(dsvc_cdf_build_lut(&u.c, u.lut.data()) == DSVC_OK, "build_lut(copy)");
            const int r = (int)rnd_below((uint32_t)u.rows);
            const int32_t L = u.len[r];
            int32_t* row = &u.cdf[(size_t)r * u.stride];
            const int mode = (int)rnd_below(5);
            if (mode == 0) row[0] = 1;
            else if (mode == 1) row[L - 1] = 65535;
            else if (mode == 2 && L > 2) row[1 + rnd_below((uint32_t)L - 2)] = 70000;
            else if (mode == 3) u.len[r] = u.stride + 1 + (int32_t)rnd_below(1000);          // length beyond the stride (also read by the fingerprint)
            else u.len[r] = -(int32_t)rnd_below(5);
            dsvc_cdf_table c = u.c; c.lut = u.lut.data();
            // first / last entry and the row lengths are part of the fingerprint: such an edit revokes the LUT's trust and the row
            // validation refuses the table.  An edit of an interior entry of a long row may escape the sampled fingerprint (the header's
            // contract: rebuild the LUT after ANY change) -- then the result is garbage or DSVC_E_RANGE, but never an out-of-bounds access.
            const bool must_refuse = mode != 2 || L <= 9;
            const int rc = decode(c, bytes.data(), bytes.size(), idx, out);
            CHECK(must_refuse ? rc == DSVC_E_ARG : (rc == DSVC_E_ARG || rc == DSVC_E_RANGE || rc == DSVC_OK), "edited table: unexpected decoder return code");
            void* e = nullptr;
            CHECK(dsvc_rans_encoder_create(&e) == DSVC_OK, "encoder_create");
            const int rc2 = dsvc_rans_encode_with_indexes(e, sym.data(), idx.data(), (int64_t)n, &c);
            CHECK(must_refuse ? rc2 == DSVC_E_ARG : (rc2 == DSVC_E_ARG || rc2 == DSVC_OK), "edited table: unexpected encoder return code");
            if (rc2 == DSVC_OK) { const int64_t nb = dsvc_rans_flush(e, nullptr, 0); CHECK(nb >= 8, "flush after a garbage table"); }
            dsvc_rans_encoder_destroy(e);
            std::vector<uint16_t> l2((size_t)u.rows * 256);
            CHECK(dsvc_cdf_build_lut(&u.c, l2.data()) == DSVC_E_ARG, "edited table was not refused by build_lut");
            dsvc_cdf_forget_lut(u.lut.data());
            ++n_forged;
        }
        {   // out-of-range CDF indexes
            std::vector<int32_t> bad_idx(idx);
            bad_idx[rnd_below((uint32_t)n)] = (rnd() & 1) ? t.rows + (int32_t)rnd_below(1000) : -1 - (int32_t)rnd_below(1000);
            dsvc_cdf_table c = t.c; c.lut = t.lut.data();
            CHECK(decode(c, bytes.data(), bytes.size(), bad_idx, out) == DSVC_E_ARG, "out-of-range index accepted by the decoder");
            void* e = nullptr;
            CHECK(dsvc_rans_encoder_create(&e) == DSVC_OK, "encoder_create");
            CHECK(dsvc_rans_encode_with_indexes(e, sym.data(), bad_idx.data(), (int64_t)n, &c) == DSVC_E_ARG, "out-of-range index accepted by the encoder");
            dsvc_rans_encoder_destroy(e);
            ++n_forged;
        }
        dsvc_cdf_forget_lut(t.lut.data());
    }
    {   // empty input: the 8-byte stream of the initial state (known answer), decodes to nothing
        Table t; CHECK(make_table(t, 2, 5), "make_table");
        std::vector<uint8_t> bytes;
        CHECK(encode(t, true, {}, {}, bytes) == DSVC_OK && bytes.size() == 8, "empty encode");
        const uint8_t want[8] = {0, 0, 0, 0x80, 0, 0, 0, 0};
        CHECK(memcmp(bytes.data(), want, 8) == 0, "empty stream bytes");
        std::vector<int32_t> out;
        CHECK(decode(t.c, bytes.data(), 8, {}, out) == DSVC_OK, "empty decode");
        CHECK(dsvc_rans_flush(nullptr, nullptr, 0) == DSVC_E_ARG, "flush(NULL)");
        void* d = nullptr;
        CHECK(dsvc_rans_decoder_create(&d, bytes.data(), 4) == DSVC_E_ARG, "4-byte stream accepted");
        dsvc_cdf_forget_lut(t.lut.data());
    }
    {   // LUT registry: more live LUTs than slots -> counted, tables still coded correctly (per-call validation)
        std::vector<Table> many(300);
        for (auto& t : many) CHECK(make_table(t, 1, 4), "make_table(many)");
        CHECK(dsvc_get_counter("lut_registry_overflow") >= 300 - 256, "registry overflow not counted");
        std::vector<int32_t> sym = {many[299].off[0]}, idx = {0}, out;
        std::vector<uint8_t> bytes;
        CHECK(encode(many[299], true, sym, idx, bytes) == DSVC_OK, "encode with an unregistered LUT");
        dsvc_cdf_table c = many[299].c; c.lut = many[299].lut.data();
        CHECK(decode(c, bytes.data(), bytes.size(), idx, out) == DSVC_OK && out[0] == sym[0], "decode with an unregistered LUT");
        for (auto& t : many) dsvc_cdf_forget_lut(t.lut.data());
    }
    // ---- 4. pmf_to_quantized_cdf / weight packing --------------------------------------------------------------
    {
        int32_t cdf[16];
        const float nanv = std::nanf("");
        float a[4] = {0.5f, nanv, 0.25f, 0.25f}; CHECK(dsvc_pmf_to_quantized_cdf(a, 4, 16, cdf) == DSVC_E_PMF, "NaN pmf");
        float b[4] = {0.5f, -0.1f, 0.3f, 0.3f}; CHECK(dsvc_pmf_to_quantized_cdf(b, 4, 16, cdf) == DSVC_E_PMF, "negative pmf");
        float z[4] = {0, 0, 0, 0}; CHECK(dsvc_pmf_to_quantized_cdf(z, 4, 16, cdf) == DSVC_E_PMF, "zero pmf");
        float i4[4] = {INFINITY, 0, 0, 0}; CHECK(dsvc_pmf_to_quantized_cdf(i4, 4, 16, cdf) == DSVC_E_PMF, "infinite pmf");
        float s[8] = {1.f, 0, 0, 0, 0, 0, 0, 1e-30f}; CHECK(dsvc_pmf_to_quantized_cdf(s, 8, 16, cdf) == DSVC_OK, "spike pmf");
        for (int i = 0; i < 8; ++i) CHECK(cdf[i + 1] > cdf[i], "spike pmf: zero-frequency bin");
        CHECK(cdf[0] == 0 && cdf[8] == 65536, "spike pmf: span");
        CHECK(dsvc_pmf_to_quantized_cdf(s, 0, 16, cdf) == DSVC_E_ARG && dsvc_pmf_to_quantized_cdf(s, 8, 17, cdf) == DSVC_E_ARG, "bad pmf arguments");
        for (int iter = 0; iter < 50; ++iter) {
            const int n = 1 + (int)rnd_below(3000);
            std::vector<float> p(n);
            for (auto& v : p) v = (rnd() & 3) ? 0.f : (float)rnd_below(1000) * 1e-3f;
            std::vector<int32_t> c(n + 1);
            const int rc = dsvc_pmf_to_quantized_cdf(p.data(), n, 16, c.data());
            CHECK(rc == DSVC_OK || rc == DSVC_E_PMF, "random sparse pmf");
            if (rc == DSVC_OK) for (int i = 0; i < n; ++i) CHECK(c[i + 1] > c[i], "random sparse pmf: zero-frequency bin");
        }
        for (int iter = 0; iter < 30; ++iter) {
            const int nseg = 1 + (int)rnd_below(3), cout = 1 + (int)rnd_below(70), k = 1 + 2 * (int)rnd_below(4), compute = (int)rnd_below(2);
            int32_t seg[3]; int cin = 0;
            for (int s2 = 0; s2 < nseg; ++s2) { seg[s2] = 1 + (int32_t)rnd_below(70); cin += seg[s2]; }
            std::vector<float> w((size_t)cout * cin * k * k);
            for (auto& v : w) v = (float)((int)rnd_below(2001) - 1000) * ((rnd() & 15) ? 1e-3f : 1e3f);      // incl. values beyond the f16 range
            const int64_t nb = dsvc_conv_packed_bytes(cout, nseg, seg, k, k, compute);
            CHECK(nb > 0, "conv_packed_bytes");
            std::vector<uint8_t> packed((size_t)nb);       // exact size: an over-write is an ASan report
            CHECK(dsvc_conv_pack_weights(w.data(), cout, nseg, seg, k, k, compute, packed.data()) == DSVC_OK, "conv_pack_weights");
        }
        std::vector<float> wt((size_t)5 * 3 * 25), bias(3), wo((size_t)12 * 5 * 9), b4(12);
        CHECK(dsvc_deconv5x5s2_to_conv3x3(wt.data(), bias.data(), 5, 3, wo.data(), b4.data()) == DSVC_OK, "deconv rewrite");
    }
    printf("fuzz_host_coder: %ld round trips, %ld truncated streams (%ld refused), %ld corrupted streams, %ld forged-table cases: no sanitizer report\n",
           n_round, n_trunc, n_rejected, n_flip, n_forged);
    return 0;
}
