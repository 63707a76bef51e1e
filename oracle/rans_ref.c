/* Oracle: C restatement of the range coder the reference calls through compressai 1.2.1
 * (rans_interface.cpp over ryg_rans rans64.h) and of pmf_to_quantized_cdf.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product library.
 * "parity unpinned": compressai is not vendored under /root/reference; the algorithm is
 * restated from its published form.  Reference call sites:
 * image_model.py:221,253-254 (BufferedRansEncoder.encode_with_indexes + flush),
 * image_model.py:273-274,288 (RansDecoder.set_stream / decode_stream).
 *
 * Written as a plain, obviously-correct scalar coder (linear CDF search, explicit symbol list);
 * the product's coder (deepsvc_amd/csrc/rans_host.cpp) is a different, optimised implementation
 * that must produce the same bytes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define PRECISION 16u
#define BYPASS_BITS 4u
#define BYPASS_MAX 15
#define RANS_L (1ull << 31)

typedef struct { uint16_t start; uint32_t range; uint8_t bypass; } sym_t;

/* Returns number of bytes written at the END of out (out has cap_words 32-bit words);
 * *first_word receives the word index where the stream begins.  -1 on overflow. */
long orc_rans_encode(const int32_t* symbols, const int32_t* indexes, long n,
                     const int32_t* cdf, long cdf_stride, const int32_t* cdf_length,
                     const int32_t* offset, uint32_t* out, long cap_words, long* first_word) {
    long cap = 4 * n + 16, m = 0;
    sym_t* s = (sym_t*)malloc(sizeof(sym_t) * (size_t)cap);
    if (!s) return -1;
    for (long i = 0; i < n; ++i) {
        if (m + 40 > cap) { cap *= 2; s = (sym_t*)realloc(s, sizeof(sym_t) * (size_t)cap); if (!s) return -1; }
        const int32_t ci = indexes[i];
        const int32_t* row = cdf + (long)ci * cdf_stride;
        const int32_t max_value = cdf_length[ci] - 2;
        int32_t v = symbols[i] - offset[ci];
        uint32_t raw = 0;
        if (v < 0) { raw = (uint32_t)(-2 * v - 1); v = max_value; }
        else if (v >= max_value) { raw = (uint32_t)(2 * (v - max_value)); v = max_value; }
        s[m].start = (uint16_t)row[v]; s[m].range = (uint32_t)(row[v + 1] - row[v]); s[m].bypass = 0; ++m;
        if (v == max_value) {
            int32_t nb = 0;
            while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb;
            int32_t val = nb;
            while (val >= BYPASS_MAX) { s[m].start = BYPASS_MAX; s[m].range = BYPASS_MAX + 1; s[m].bypass = 1; ++m; val -= BYPASS_MAX; }
            s[m].start = (uint16_t)val; s[m].range = (uint32_t)val + 1; s[m].bypass = 1; ++m;
            for (int32_t j = 0; j < nb; ++j) {
                uint32_t nib = (raw >> (j * BYPASS_BITS)) & BYPASS_MAX;
                s[m].start = (uint16_t)nib; s[m].range = nib + 1; s[m].bypass = 1; ++m;
            }
        }
    }
    uint64_t x = RANS_L;
    long p = cap_words;
    for (long k = m - 1; k >= 0; --k) {
        if (p < 3) { free(s); return -1; }
        if (!s[k].bypass) {
            const uint64_t freq = s[k].range;
            const uint64_t x_max = ((RANS_L >> PRECISION) << 32) * freq;
            if (x >= x_max) { out[--p] = (uint32_t)x; x >>= 32; }
            x = ((x / freq) << PRECISION) + (x % freq) + s[k].start;
        } else {
            const uint64_t freq = 1ull << (16 - BYPASS_BITS);
            const uint64_t x_max = ((RANS_L >> 16) << 32) * freq;
            if (x >= x_max) { out[--p] = (uint32_t)x; x >>= 32; }
            x = (x << BYPASS_BITS) | s[k].start;
        }
    }
    free(s);
    if (p < 2) return -1;
    p -= 2;
    out[p] = (uint32_t)x;
    out[p + 1] = (uint32_t)(x >> 32);
    *first_word = p;
    return (cap_words - p) * 4;
}

typedef struct { const uint32_t* w; long n, pos; uint64_t x; uint32_t* own; } dec_t;

void* orc_rans_dec_open(const uint8_t* data, long nbytes) {
    dec_t* d = (dec_t*)calloc(1, sizeof(dec_t));
    d->own = (uint32_t*)malloc((size_t)nbytes + 16);
    memset(d->own, 0, (size_t)nbytes + 16);
    memcpy(d->own, data, (size_t)nbytes);
    d->w = d->own; d->n = nbytes / 4;
    d->x = (uint64_t)d->w[0] | ((uint64_t)d->w[1] << 32);
    d->pos = 2;
    return d;
}

void orc_rans_dec_close(void* h) { dec_t* d = (dec_t*)h; if (d) { free(d->own); free(d); } }

static inline uint32_t dec_bits(dec_t* d) {
    uint32_t val = (uint32_t)(d->x & BYPASS_MAX);
    d->x >>= BYPASS_BITS;
    if (d->x < RANS_L) { d->x = (d->x << 32) | d->w[d->pos < d->n ? d->pos : d->n]; d->pos++; }
    return val;
}

long orc_rans_dec_run(void* h, const int32_t* indexes, long n, const int32_t* cdf, long cdf_stride,
                      const int32_t* cdf_length, const int32_t* offset, int32_t* out) {
    dec_t* d = (dec_t*)h;
    for (long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t* row = cdf + (long)ci * cdf_stride;
        const int32_t len = cdf_length[ci];
        const int32_t max_value = len - 2;
        const uint32_t cf = (uint32_t)(d->x & 0xFFFF);
        int32_t s = 0;
        while (s + 1 < len && (uint32_t)row[s + 1] <= cf) ++s;
        const uint32_t start = (uint32_t)row[s], freq = (uint32_t)(row[s + 1] - row[s]);
        d->x = (uint64_t)freq * (d->x >> PRECISION) + cf - start;
        if (d->x < RANS_L) { d->x = (d->x << 32) | d->w[d->pos < d->n ? d->pos : d->n]; d->pos++; }
        int32_t v = s;
        if (v == max_value) {
            int32_t val = (int32_t)dec_bits(d), nb = val;
            while (val == BYPASS_MAX) { val = (int32_t)dec_bits(d); nb += val; }
            uint32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) raw |= dec_bits(d) << (j * BYPASS_BITS);
            v = (int32_t)(raw >> 1);
            v = (raw & 1) ? -v - 1 : v + max_value;
        }
        out[i] = v + offset[ci];
    }
    return 0;
}

/* pmf (float[n]) -> cdf (int32[n+1]); 0 ok, -1 invalid pmf, -2 zero total, -3 cannot steal. */
int orc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf) {
    for (int i = 0; i < n; ++i) if (!(pmf[i] >= 0.0f) || !isfinite(pmf[i])) return -1;
    uint32_t* c = (uint32_t*)cdf;
    c[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) { c[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision)); }
    for (int i = 0; i <= n; ++i) total += c[i];
    if (total == 0) return -2;
    for (int i = 0; i <= n; ++i) c[i] = (uint32_t)((((uint64_t)1 << precision) * c[i]) / total);
    for (int i = 1; i <= n; ++i) c[i] += c[i - 1];
    c[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (c[i] == c[i + 1]) {
            uint32_t best_freq = ~0u; int best = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t f = c[j + 1] - c[j];
                if (f > 1 && f < best_freq) { best_freq = f; best = j; }
            }
            if (best < 0) return -3;
            if (best < i) { for (int j = best + 1; j <= i; ++j) c[j]--; }
            else { for (int j = i + 1; j <= best; ++j) c[j]++; }
        }
    }
    return 0;
}
