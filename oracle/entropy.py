"""Oracle restatement of the compressai 1.2.1 entropy arithmetic used by the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **parity unpinned**: compressai 1.2.1
(reference env.txt:12) is a third-party dependency that is neither vendored under
/root/reference nor installed here; this file restates its published algorithm
(SURVEY.md Appendix A.2-A.5).  Reference call sites anchored:

* EntropyBottleneck      image_model.py:148,155,160,206-207,260
* GaussianConditional    image_model.py:149,181,217-219,237-238,286,290
* BufferedRansEncoder /  image_model.py:221,253-254,273-274,288
  RansDecoder
* get_scale_table        modules.py:12-18 == image_model.py:13-19

All functions are functional: entropy-bottleneck parameters are read from a state_dict
through a prefix (keys ``_matrix{i}``, ``_bias{i}``, ``_factor{i}``, ``quantiles``).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64
SCALE_BOUND = 0.11
LIKELIHOOD_BOUND = 1e-9
TAIL_MASS = 1e-9
PRECISION = 16
BYPASS_BITS = 4
BYPASS_MAX = (1 << BYPASS_BITS) - 1
RANS_L = 1 << 31
EB_FILTERS = (1, 3, 3, 3, 3, 1)


def get_scale_table(lo=SCALES_MIN, hi=SCALES_MAX, levels=SCALES_LEVELS):
    """modules.py:17-18."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


# ----------------------------------------------------------------------------------------------
# pmf -> integer CDF (compressai _CXX.pmf_to_quantized_cdf, ported there from ryg_rans) -- A.4
# ----------------------------------------------------------------------------------------------
def pmf_to_quantized_cdf(pmf, precision=PRECISION):
    pmf = np.asarray(pmf, dtype=np.float32)
    if not np.all(np.isfinite(pmf)) or np.any(pmf < 0):
        raise ValueError("invalid pmf")
    n = len(pmf)
    cdf = np.zeros(n + 1, dtype=np.uint64)
    # std::round on float: half away from zero; values are >= 0
    cdf[1:] = np.floor(pmf * np.float32(1 << precision) + np.float32(0.5)).astype(np.uint64)
    total = int(cdf.sum()) & 0xFFFFFFFF
    if total == 0:
        raise ValueError("pmf sums to zero")
    cdf = ((np.uint64(1 << precision) * cdf) // np.uint64(total)).astype(np.int64)
    cdf = np.cumsum(cdf)
    cdf[-1] = 1 << precision
    cdf = [int(v) for v in cdf]
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = 1 << 40, -1
            for j in range(n):
                f = cdf[j + 1] - cdf[j]
                if 1 < f < best_freq:
                    best_freq, best = f, j
            assert best != -1
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return np.asarray(cdf, dtype=np.int32)


def _pmf_rows_to_cdf(pmf, tail_mass, pmf_length, max_length):
    rows = len(pmf_length)
    cdf = np.zeros((rows, max_length + 2), dtype=np.int32)
    for i in range(rows):
        n = int(pmf_length[i])
        prob = np.concatenate([pmf[i, :n], tail_mass[i]]).astype(np.float32)
        row = pmf_to_quantized_cdf(prob, PRECISION)
        cdf[i, : len(row)] = row
    return cdf


# ----------------------------------------------------------------------------------------------
# EntropyBottleneck (factorised prior) -- A.2
# ----------------------------------------------------------------------------------------------
def eb_logits(sd, prefix, x):
    """x: [C,1,n] -> cumulative logits [C,1,n]."""
    for i in range(5):
        m = sd[f"{prefix}_matrix{i}"]
        x = torch.matmul(F.softplus(m), x) + sd[f"{prefix}_bias{i}"]
        if i < 4:
            x = x + torch.tanh(sd[f"{prefix}_factor{i}"]) * torch.tanh(x)
    return x


def eb_medians(sd, prefix):
    return sd[f"{prefix}quantiles"][:, :, 1:2]  # [C,1,1]


def eb_likelihood_of(sd, prefix, v):
    lower = eb_logits(sd, prefix, v - 0.5)
    upper = eb_logits(sd, prefix, v + 0.5)
    sign = -torch.sign(lower + upper)
    return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))


def eb_forward(sd, prefix, z):
    """Eval-mode forward: (z_hat, likelihood).  z: [B,C,h,w]."""
    B, C, h, w = z.shape
    v = z.permute(1, 0, 2, 3).reshape(C, 1, -1)
    med = eb_medians(sd, prefix)
    out = torch.round(v - med) + med
    lik = eb_likelihood_of(sd, prefix, out).clamp_min(LIKELIHOOD_BOUND)
    back = lambda t: t.reshape(C, B, h, w).permute(1, 0, 2, 3)
    return back(out), back(lik)


def eb_loss(sd, prefix, tail_mass=1e-9):
    """EntropyBottleneck.loss() (A.2; summed by video_model.py:169-177): sum |logits(quantiles) - (-t, 0, t)|,
    t = log(2 / tail_mass - 1)."""
    import math
    t = math.log(2.0 / tail_mass - 1.0)
    target = torch.tensor([-t, 0.0, t], dtype=torch.float32)
    return torch.abs(eb_logits(sd, prefix, sd[f"{prefix}quantiles"]) - target).sum()


def eb_tables(sd, prefix):
    """update(): integer CDF tables {cdf [C,L], cdf_length [C], offset [C]}."""
    q = sd[f"{prefix}quantiles"]
    med = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
    offset = -minima
    pmf_start = med - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
    lower = eb_logits(sd, prefix, samples - 0.5)
    upper = eb_logits(sd, prefix, samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = _pmf_rows_to_cdf(pmf.detach().numpy(), tail.detach().numpy(), pmf_length.numpy(), max_length)
    return {"cdf": cdf, "cdf_length": (pmf_length + 2).numpy().astype(np.int32),
            "offset": offset.numpy().astype(np.int32)}


def eb_compress(sd, prefix, tables, z):
    """One rANS string per batch item; per-element CDF index = its channel."""
    B, C, h, w = z.shape
    med = eb_medians(sd, prefix).reshape(1, C, 1, 1)
    sym = torch.round(z - med).to(torch.int32)
    idx = torch.arange(C, dtype=torch.int32).view(1, C, 1, 1).expand(B, C, h, w)
    return [rans_encode(sym[b].reshape(-1).numpy(), idx[b].reshape(-1).numpy(),
                        tables["cdf"], tables["cdf_length"], tables["offset"]) for b in range(B)]


def eb_decompress(sd, prefix, tables, strings, shape):
    C = sd[f"{prefix}quantiles"].shape[0]
    h, w = int(shape[0]), int(shape[1])
    med = eb_medians(sd, prefix).reshape(1, C, 1, 1)
    idx = torch.arange(C, dtype=torch.int32).view(C, 1, 1).expand(C, h, w).reshape(-1).numpy()
    outs = []
    for s in strings:
        dec = RansDecoder(s)
        sym = dec.decode(idx, tables["cdf"], tables["cdf_length"], tables["offset"])
        outs.append(torch.from_numpy(sym.astype(np.float32)).reshape(1, C, h, w))
    return torch.cat(outs, 0) + med


# ----------------------------------------------------------------------------------------------
# GaussianConditional -- A.3
# ----------------------------------------------------------------------------------------------
def _phi_c(x):
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def gc_likelihood(y_hat, scale, mu):
    v = torch.abs(y_hat - mu)
    s = torch.clamp_min(scale, SCALE_BOUND)
    lik = _phi_c((0.5 - v) / s) - _phi_c((-0.5 - v) / s)
    return lik.clamp_min(LIKELIHOOD_BOUND)


def gc_forward(y, scale, mu):
    y_hat = torch.round(y - mu) + mu
    return y_hat, gc_likelihood(y_hat, scale, mu)


def gc_tables(scale_table=None):
    from scipy.stats import norm
    table = get_scale_table() if scale_table is None else scale_table
    table = torch.tensor([float(s) for s in table])
    mult = -norm.ppf(TAIL_MASS / 2)
    pc = torch.ceil(table * mult).int()
    pmf_length = 2 * pc + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pc[:, None]).float()
    s = table.unsqueeze(1).float()
    upper = _phi_c((0.5 - samples) / s)
    lower = _phi_c((-0.5 - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _pmf_rows_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return {"cdf": cdf, "cdf_length": (pmf_length + 2).numpy().astype(np.int32),
            "offset": (-pc).numpy().astype(np.int32), "scale_table": table}


def gc_build_indexes(scale, table):
    s = torch.clamp_min(scale, SCALE_BOUND)
    idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


# ----------------------------------------------------------------------------------------------
# rANS (ryg_rans rans64.h + compressai rans_interface.cpp) -- A.5
# ----------------------------------------------------------------------------------------------
_clib = None


def _c():
    """The C restatement (oracle/rans_ref.c) when built; None otherwise."""
    global _clib
    if _clib is None:
        from . import build
        _clib = build.load() or False
    return _clib or None


def rans_symbol_stream(symbols, indexes, cdf, cdf_length, offset):
    """encode_with_indexes(): expands (symbol, index) pairs to (start, range, bypass) triples."""
    out = []
    for s, ci in zip(symbols, indexes):
        ci = int(ci)
        row = cdf[ci]
        max_value = int(cdf_length[ci]) - 2
        v = int(s) - int(offset[ci])
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, max_value
        elif v >= max_value:
            raw, v = 2 * (v - max_value), max_value
        out.append((int(row[v]), int(row[v + 1]) - int(row[v]), False))
        if v == max_value:
            n = 0
            while (raw >> (n * BYPASS_BITS)) != 0:
                n += 1
            val = n
            while val >= BYPASS_MAX:
                out.append((BYPASS_MAX, BYPASS_MAX + 1, True))
                val -= BYPASS_MAX
            out.append((val, val + 1, True))
            for j in range(n):
                nib = (raw >> (j * BYPASS_BITS)) & BYPASS_MAX
                out.append((nib, nib + 1, True))
    return out


def rans_flush(stream):
    """BufferedRansEncoder.flush(): code the triples in reverse; 32-bit words, little endian."""
    x = RANS_L
    words = []  # emitted back-to-front
    for start, rng, bypass in reversed(stream):
        if not bypass:
            x_max = ((RANS_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_BITS)
            x_max = ((RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << BYPASS_BITS) | start
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return np.asarray(words, dtype="<u4").tobytes()


def rans_encode_py(symbols, indexes, cdf, cdf_length, offset):
    return rans_flush(rans_symbol_stream(symbols, indexes, cdf, cdf_length, offset))


def rans_encode(symbols, indexes, cdf, cdf_length, offset, force_python=False):
    lib = None if force_python else _c()
    if lib is None:
        return rans_encode_py(symbols, indexes, cdf, cdf_length, offset)
    return lib.encode(symbols, indexes, cdf, cdf_length, offset)


class RansDecoder:
    """RansDecoder.set_stream + decode_stream (stateful: successive calls continue the stream)."""

    def __init__(self, data, force_python=False):
        self._lib = None if force_python else _c()
        if self._lib is not None:
            self._h = self._lib.dec_open(data)
            return
        self.words = np.frombuffer(data, dtype="<u4")
        self.x = int(self.words[0]) | (int(self.words[1]) << 32)
        self.pos = 2

    def _renorm(self):
        if self.x < RANS_L:
            self.x = (self.x << 32) | int(self.words[self.pos])
            self.pos += 1

    def _bits(self):
        val = self.x & BYPASS_MAX
        self.x >>= BYPASS_BITS
        self._renorm()
        return val

    def decode(self, indexes, cdf, cdf_length, offset):
        if self._lib is not None:
            return self._lib.dec_run(self._h, indexes, cdf, cdf_length, offset)
        out = np.empty(len(indexes), dtype=np.int32)
        for i, ci in enumerate(indexes):
            ci = int(ci)
            row = cdf[ci]
            n = int(cdf_length[ci])
            max_value = n - 2
            cf = self.x & 0xFFFF
            s = 0
            while s + 1 < n and int(row[s + 1]) <= cf:  # first entry > cf, minus one
                s += 1
            start, rng = int(row[s]), int(row[s + 1]) - int(row[s])
            self.x = rng * (self.x >> PRECISION) + cf - start
            self._renorm()
            v = s
            if v == max_value:
                val = self._bits()
                nby = val
                while val == BYPASS_MAX:
                    val = self._bits()
                    nby += val
                raw = 0
                for j in range(nby):
                    raw |= self._bits() << (j * BYPASS_BITS)
                v = raw >> 1
                v = -v - 1 if (raw & 1) else v + max_value
            out[i] = v + int(offset[ci])
        return out
