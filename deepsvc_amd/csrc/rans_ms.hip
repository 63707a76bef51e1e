// rans_ms.hip -- the GPU-decodable MULTI-STREAM range coder (the optional half: "a multi-stream GPU-decodable rANS
// variant with measured bpp overhead"; SURVEY section 7.2 "Decode latency chain").
//
// NOT the reference's bitstream: compressai's coder (image_model.py:221,253-254,273-274,288) keeps ONE 64-bit rANS state for a whole y string,
// so decoding is one dependent chain per symbol by format -- on the host, with a device -> host -> device round trip and a stream
// synchronisation at every dependency level of the slice scan (6 per codec and frame).  This is a LABELLED alternative mode (container flag
// FLAG_MULTISTREAM, deepsvc_amd/bitstream.py) that codes the same symbols with the same integer CDF tables:
//   * a level's n symbols (plane order [C][h][w], as the scan kernels write them) are dealt round-robin to n_streams = ceil(n / S)
//     independent streams: symbol j belongs to stream j mod n_streams, so the lanes of a wave read / write consecutive plane entries;
//   * each stream is a 32-bit rANS with 16-bit renormalisation words (state in [2^16, 2^32), 16-bit probabilities: the tables'
//     precision), one GPU thread per stream; the encoder walks its symbols backwards and writes words down from the end of a fixed slot;
//   * out-of-support values are coded as compressai codes them (rans_interface.cpp), inside the stream: the row's escape symbol, then the
//     zig-zag magnitude's nibble count and its nibbles, least significant first, each a uniform 4-bit item (frequency 2^12 of 2^16: exactly
//     4 bits) -- with this weight set 3-5 % of the symbols escape, almost all with one nibble;
//   * per stream the string carries a 4-byte final state and a 2-byte word count: 6 bytes per S symbols of overhead (S = 256: ~5 %
//     at 3.5 bits per symbol; measured per frame in DESIGN.md).
// Decoding a level is then ONE kernel launch between the scan kernels -- no host round trip, no synchronisation.
#include "common.h"

namespace {

struct MsTab { const int32_t* cdf; const int32_t* len; const int32_t* off; const uint16_t* lut; int rows, stride; };

__global__ void rans_ms_encode_kernel(const int32_t* __restrict__ sym, const int32_t* __restrict__ idx, int64_t n, int n_streams, MsTab T,
                                      uint16_t* __restrict__ slots, int slot_words, uint32_t* __restrict__ out_state, uint16_t* __restrict__ out_len) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_streams) return;
    uint16_t* base = slots + (size_t)t * slot_words;
    int w = slot_words;                                  // words are written downwards: the stream is base[w .. slot_words)
    uint32_t x = 1u << 16;
    auto put = [&](uint32_t start, uint32_t freq) {      // freq in [1, 65535]
        if ((x >> 16) >= freq) { base[--w] = (uint16_t)x; x >>= 16; }
        x = ((x / freq) << 16) + (x % freq) + start;
    };
    auto put_nib = [&](uint32_t v4) {                    // a uniform 4-bit item: start = v4 << 12, freq = 1 << 12
        if ((x >> 16) >= 4096u) { base[--w] = (uint16_t)x; x >>= 16; }
        x = ((x >> 12) << 16) + (x & 0xFFFu) + (v4 << 12);
    };
    const int64_t n_t = (n - t + n_streams - 1) / n_streams;
    for (int64_t i = n_t - 1; i >= 0; --i) {
        const int64_t j = (int64_t)t + i * n_streams;
        const int ci = idx[j];
        const int32_t* row = T.cdf + (int64_t)ci * T.stride;
        const int32_t maxv = T.len[ci] - 2;
        int32_t v = (int32_t)((uint32_t)sym[j] - (uint32_t)T.off[ci]);
        if ((uint32_t)v >= (uint32_t)maxv) {             // escape: the decoder reads [escape symbol, nibble count, nibbles LSB first] -> pushed in reverse
            const uint32_t raw = v < 0 ? (uint32_t)(-2 * (int64_t)v - 1) : (uint32_t)(2 * (int64_t)(v - maxv));
            int nb = 0;
            while (nb < 8 && (raw >> (4 * nb)) != 0) ++nb;
            for (int k = nb - 1; k >= 0; --k) put_nib((raw >> (4 * k)) & 15u);
            put_nib((uint32_t)nb);
            v = maxv;
        }
        const uint32_t start = (uint32_t)row[v], freq = (uint32_t)row[v + 1] - start;
        if (freq < 65536u) put(start, freq);             // a one-symbol row (probability 1) costs nothing and changes nothing
    }
    out_state[t] = x;
    out_len[t] = (uint16_t)(slot_words - w);
}

__global__ void rans_ms_decode_kernel(const int32_t* __restrict__ idx, int64_t n, int n_streams, MsTab T, const uint32_t* __restrict__ state,
                                      const uint32_t* __restrict__ word_off, const uint16_t* __restrict__ payload, int64_t payload_words,
                                      int32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_streams) return;
    uint32_t x = state[t];
    int64_t pos = word_off[t];
    const int64_t end = word_off[t + 1];                 // a corrupt stream reads zeros past its end instead of its neighbour's words
    auto renorm = [&]() {
        if (x < (1u << 16)) { const uint32_t wv = (pos < end && pos < payload_words) ? payload[pos] : 0u; ++pos; x = (x << 16) | wv; }
    };
    const int64_t n_t = (n - t + n_streams - 1) / n_streams;
    for (int64_t i = 0; i < n_t; ++i) {
        const int64_t j = (int64_t)t + i * n_streams;
        const int ci = idx[j];
        const int32_t* row = T.cdf + (int64_t)ci * T.stride;
        const int32_t maxv = T.len[ci] - 2;
        const uint32_t cf = x & 0xFFFFu;
        int32_t s = T.lut[(size_t)ci * 256 + (cf >> 8)];              // largest symbol whose start is <= the bucket's first value, then forward
        while (s < maxv && (uint32_t)row[s + 1] <= cf) ++s;
        const uint32_t start = (uint32_t)row[s], freq = (uint32_t)row[s + 1] - start;
        if (freq < 65536u) { x = freq * (x >> 16) + cf - start; renorm(); }
        int32_t v = s;
        if (s == maxv) {
            auto get_nib = [&]() { const uint32_t c = x & 0xFFFFu; x = 4096u * (x >> 16) + (c & 0xFFFu); renorm(); return c >> 12; };
            const uint32_t nb = get_nib();
            uint32_t raw = 0;
            for (uint32_t k = 0; k < nb && k < 8; ++k) raw |= get_nib() << (4 * k);
            v = (int32_t)(raw >> 1);
            v = (raw & 1u) ? -v - 1 : (int32_t)((uint32_t)v + (uint32_t)maxv);
        }
        out[j] = (int32_t)((uint32_t)v + (uint32_t)T.off[ci]);
    }
}

int check_tab(const dsvc_cdf_table* t, const char* who) {
    DSVC_REQUIRE(t && t->cdf && t->cdf_length && t->offset && t->lut && t->rows > 0 && t->stride > 1, "%s: device CDF table (cdf, cdf_length, offset, lut) incomplete", who);
    return DSVC_OK;
}

}  // namespace

extern "C" int64_t dsvc_rans_ms_slot_words(int32_t symbols_per_stream) {
    // a symbol costs at most 16 bits (probability >= 2^-16) + 4 bits of nibble count + 8 x 4 bits of magnitude = 52 bits -> 4 words
    return symbols_per_stream > 0 ? 4 * (int64_t)symbols_per_stream + 2 : DSVC_E_ARG;
}

extern "C" int dsvc_rans_ms_encode(void* stream, const int32_t* symbols, const int32_t* indexes, int64_t n, int32_t n_streams, const dsvc_cdf_table* dev_table,
                                   uint16_t* slots, int32_t slot_words, uint32_t* states, uint16_t* lens) {
    DSVC_REQUIRE(symbols && indexes && slots && states && lens && n >= 0 && n_streams > 0, "rans_ms_encode: bad argument");
    if (int rc = check_tab(dev_table, "rans_ms_encode")) return rc;
    const int64_t per = (n + n_streams - 1) / n_streams;
    DSVC_REQUIRE(slot_words >= 4 * per + 2 && slot_words < 65536, "rans_ms_encode: slot of %d words for %lld symbols per stream (need 4 n + 2, < 65536)", slot_words, (long long)per);
    const MsTab T{dev_table->cdf, dev_table->cdf_length, dev_table->offset, dev_table->lut, dev_table->rows, dev_table->stride};
    hipLaunchKernelGGL(rans_ms_encode_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, (hipStream_t)stream, symbols, indexes, n, n_streams, T, slots, slot_words, states, lens);
    DSVC_CHECK_LAUNCH("rans_ms_encode_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_rans_ms_decode(void* stream, const int32_t* indexes, int64_t n, int32_t n_streams, const dsvc_cdf_table* dev_table,
                                   const uint32_t* states, const uint32_t* word_offsets, const uint16_t* payload, int64_t payload_words, int32_t* symbols_out) {
    DSVC_REQUIRE(indexes && states && word_offsets && payload && symbols_out && n >= 0 && n_streams > 0 && payload_words >= 0, "rans_ms_decode: bad argument");
    if (int rc = check_tab(dev_table, "rans_ms_decode")) return rc;
    const MsTab T{dev_table->cdf, dev_table->cdf_length, dev_table->offset, dev_table->lut, dev_table->rows, dev_table->stride};
    hipLaunchKernelGGL(rans_ms_decode_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, (hipStream_t)stream, indexes, n, n_streams, T, states, word_offsets, payload,
                       payload_words, symbols_out);
    DSVC_CHECK_LAUNCH("rans_ms_decode_kernel");
    return DSVC_OK;
}
