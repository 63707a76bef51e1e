"""Plain-Python restatement of the multi-stream range coder (csrc/rans_ms.hip; string layout: deepsvc_amd/entropy_models.ms_pack) -- test
infrastructure that pins the FORMAT: a level's symbols dealt round-robin to n_streams 32-bit rANS streams with 16-bit words, 16-bit
probabilities from the codec's integer CDF tables, escapes as the row's last symbol + the zig-zag magnitude's nibble count and nibbles (uniform
4-bit items, least significant first: compressai's escape code inside the stream)."""
import numpy as np

L = 1 << 16


def encode_level(sym, idx, cdf, cdf_length, offset, n_streams):
    """-> (states uint32[n_streams], lens uint16[n_streams], payload uint16[...]) exactly as the GPU encoder + ms_pack lay them out."""
    sym, idx = np.asarray(sym, dtype=np.int64), np.asarray(idx, dtype=np.int64)
    states, lens, words_all = [], [], []
    for t in range(n_streams):
        x, words = L, []                      # words collected in EMISSION order (the stream is their reverse)

        def put(start, freq):
            nonlocal x
            if (x >> 16) >= freq:
                words.append(x & 0xFFFF); x >>= 16
            x = ((x // freq) << 16) + (x % freq) + start

        def put_nib(v4):
            nonlocal x
            if (x >> 16) >= 4096:
                words.append(x & 0xFFFF); x >>= 16
            x = ((x >> 12) << 16) + (x & 0xFFF) + (v4 << 12)

        js = list(range(t, len(sym), n_streams))
        for j in reversed(js):
            ci = int(idx[j]); row = cdf[ci]; maxv = int(cdf_length[ci]) - 2
            v = int(sym[j]) - int(offset[ci])
            if not 0 <= v < maxv:
                raw = (-2 * v - 1) if v < 0 else 2 * (v - maxv)
                raw &= 0xFFFFFFFF
                nb = 0
                while nb < 8 and (raw >> (4 * nb)) != 0:
                    nb += 1
                for k in range(nb - 1, -1, -1):
                    put_nib((raw >> (4 * k)) & 15)
                put_nib(nb)
                v = maxv
            start, freq = int(row[v]), int(row[v + 1]) - int(row[v])
            if freq < 65536:
                put(start, freq)
        states.append(x); lens.append(len(words)); words_all.extend(reversed(words))
    return np.asarray(states, dtype=np.uint32), np.asarray(lens, dtype=np.uint16), np.asarray(words_all, dtype=np.uint16)


def decode_level(n, idx, cdf, cdf_length, offset, states, lens, payload):
    idx = np.asarray(idx, dtype=np.int64)
    n_streams = len(states)
    off = np.concatenate([[0], np.cumsum(np.asarray(lens, dtype=np.int64))])
    out = np.zeros(n, dtype=np.int32)
    for t in range(n_streams):
        x, pos, end = int(states[t]), int(off[t]), int(off[t + 1])

        def renorm():
            nonlocal x, pos
            if x < L:
                w = int(payload[pos]) if pos < end else 0
                pos += 1
                x = (x << 16) | w

        for j in range(t, n, n_streams):
            ci = int(idx[j]); row = cdf[ci]; maxv = int(cdf_length[ci]) - 2
            cf = x & 0xFFFF
            s = int(np.searchsorted(row[:maxv + 1], cf, side="right")) - 1
            start, freq = int(row[s]), int(row[s + 1]) - int(row[s])
            if freq < 65536:
                x = freq * (x >> 16) + cf - start
                renorm()
            v = s
            if s == maxv:
                def get_nib():
                    nonlocal x
                    c = x & 0xFFFF
                    x = 4096 * (x >> 16) + (c & 0xFFF)
                    renorm()
                    return c >> 12
                nb = get_nib()
                raw = 0
                for k in range(min(nb, 8)):
                    raw |= get_nib() << (4 * k)
                v = raw >> 1
                v = (-v - 1) if raw & 1 else v + maxv
            out[j] = v + int(offset[ci])
    return out
