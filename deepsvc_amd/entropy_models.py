"""Entropy models of the hot path: parameters/buffers keyed like compressai 1.2.1 (so reference
checkpoints load), integer CDF construction on the host, symbol arithmetic on the device.

Replaces `compressai.entropy_models.EntropyBottleneck / GaussianConditional` as used at
image_model.py:148-149,155-162,181,206-207,217-219,237-239,260,286,290 (semantics restated from the
published compressai 1.2.1 algorithm; that package is not vendored in the
reference, so this restatement is *unpinned* -- see DESIGN.md).

Division of labour:
  host  (torch-CPU fp32, once per `update()`): pmf tabulation -> `dsvc_pmf_to_quantized_cdf` -> int32 tables;
  host  (C++): rANS encode / decode (`ops.RansEncoder/RansDecoder`);
  device (HIP): quantise, index lookup, dequantise, likelihoods (`csrc/entropy.hip`).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from . import ops
from .ops import Act, to_act


class _Bound(nn.Module):
    """Holds the `bound` buffer of compressai's LowerBound (key compatibility only)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


def _rows_to_cdf(pmf, tail_mass, pmf_length, max_length):
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        n = int(pmf_length[i])
        row = ops.pmf_to_quantized_cdf(np.concatenate([pmf[i, :n], tail_mass[i]]), 16)
        cdf[i, : len(row)] = row
    return cdf


_TABLE_GEN = [0]          # bumped whenever any entropy model's tables change (see _EntropyModel.tables_changed)


class _EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9):
        super().__init__()
        self._table_gen = 0
        self.likelihood_lower_bound = _Bound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._host_tables = None

    offset = property(lambda s: s._offset)
    quantized_cdf = property(lambda s: s._quantized_cdf)
    cdf_length = property(lambda s: s._cdf_length)

    def device_tables(self, device):
        """ops.DeviceCdfTables for the multi-stream coder (rebuilt when the tables change)."""
        key = (str(device), self._table_gen)
        cur = self.__dict__.get("_dev_cdf")
        if cur is None or cur[0] != key:
            cur = self.__dict__["_dev_cdf"] = (key, ops.DeviceCdfTables(self.host_tables(), device))
        return cur[1]

    def tables_changed(self):
        """The host / device tables of this model are about to differ from what any recording made so far has seen: recorded frame
        programs (video_model.DeepSVC._segments) and scan programs freeze POINTERS to the device-side tables and would keep replaying
        with the old ones while the host range coder already uses the new CDFs -- silently undecodable streams.  A process-wide
        generation number, sampled into every invalidation fingerprint."""
        _TABLE_GEN[0] += 1
        self._table_gen = _TABLE_GEN[0]
        self._host_tables = None

    def _store_tables(self, cdf, cdf_length, offset):
        self.tables_changed()
        dev = self._offset.device
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._cdf_length = torch.from_numpy(cdf_length.astype(np.int32)).to(dev)
        self._offset = torch.from_numpy(offset.astype(np.int32)).to(dev)
        self._host_tables = None

    def host_tables(self):
        """ops.CdfTables built from the registered buffers (so tables loaded from a checkpoint are honoured)."""
        if self._host_tables is None:
            if self._quantized_cdf.numel() == 0:
                raise RuntimeError("entropy model tables are empty: call update() first (test_video.py:235)")
            self._host_tables = ops.CdfTables(self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(),
                                              self._offset.cpu().numpy())
        return self._host_tables


# ---- multi-stream strings (csrc/rans_ms.hip; a LABELLED non-reference format: bitstream.FLAG_MULTISTREAM) -------------------------------------
# string := "MSR1" | u32 n_levels | n_levels x {u32 n_symbols, u32 n_streams} | per level: u32 state[n_streams] | u16 words[n_streams] (+ pad to
#           4 bytes) | u16 payload[sum words] (+ pad)        -- a level = one dependency level of the slice scan; stream t of a level codes its
#           symbols t, t + n_streams, ... (plane order); all integers little-endian
MS_MAGIC = b"MSR1"


def ms_pack(levels):
    """levels: [(n_symbols, states uint32[n_streams], lens uint16[n_streams], payload uint16[sum lens])] -> bytes."""
    import struct
    out = [MS_MAGIC, struct.pack("<I", len(levels))]
    for n, st, ln, _ in levels:
        out.append(struct.pack("<II", int(n), len(st)))
    for _, st, ln, pay in levels:
        assert int(ln.sum()) == len(pay)
        out.append(np.ascontiguousarray(st, dtype="<u4").tobytes())
        b = np.ascontiguousarray(ln, dtype="<u2").tobytes()
        out.append(b + b"\0" * (-len(b) % 4))
        b = np.ascontiguousarray(pay, dtype="<u2").tobytes()
        out.append(b + b"\0" * (-len(b) % 4))
    return b"".join(out)


def ms_unpack(data):
    """bytes -> [(n_symbols, states, lens, payload)] (numpy views); raises ValueError on a malformed string."""
    import struct
    data = bytes(data)
    if len(data) < 8 or data[:4] != MS_MAGIC:
        raise ValueError("not a multi-stream string")
    n_levels, = struct.unpack_from("<I", data, 4)
    if n_levels > 64 or len(data) < 8 + 8 * n_levels:
        raise ValueError("multi-stream string: bad level count")
    heads = [struct.unpack_from("<II", data, 8 + 8 * k) for k in range(n_levels)]
    pos, out = 8 + 8 * n_levels, []
    for n, ns in heads:
        if ns == 0 or ns > max(n, 1) or pos + 4 * ns + 2 * ns > len(data):
            raise ValueError("multi-stream string: truncated stream table")
        st = np.frombuffer(data, dtype="<u4", count=ns, offset=pos); pos += 4 * ns
        ln = np.frombuffer(data, dtype="<u2", count=ns, offset=pos); pos += 2 * ns + (-(2 * ns) % 4)
        nw = int(ln.astype(np.int64).sum())
        if pos + 2 * nw > len(data):
            raise ValueError("multi-stream string: truncated payload")
        pay = np.frombuffer(data, dtype="<u2", count=nw, offset=pos); pos += 2 * nw + (-(2 * nw) % 4)
        out.append((n, st, ln, pay))
    return out


class EntropyBottleneck(_EntropyModel):
    """Factorised prior (A.2).  Parameter names `_matrix{i}`, `_bias{i}`, `_factor{i}`, `quantiles`."""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3)):
        super().__init__()
        self.channels, self.filters = int(channels), tuple(int(f) for f in filters)
        self.init_scale, self.tail_mass = float(init_scale), float(tail_mass)
        assert self.filters == (3, 3, 3, 3), "device kernel is specialised for the 1-3-3-3-3-1 density network"
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))
        self._dev_params = None

    # ---- host side ------------------------------------------------------------------------------
    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    @staticmethod
    def _logits(params, x, nlayers=5):
        """Cumulative logits of the density network; params: name -> tensor."""
        for i in range(nlayers):
            x = torch.matmul(F.softplus(params[f"_matrix{i:d}"]), x) + params[f"_bias{i:d}"]
            if i < nlayers - 1:
                x = x + torch.tanh(params[f"_factor{i:d}"]) * torch.tanh(x)
        return x

    def _logits_cumulative(self, x):
        return self._logits(dict(self.named_parameters(recurse=False)), x)

    def loss(self):
        """aux loss (video_model.py:169-177); training-only, evaluated with torch on whatever device holds the params."""
        return torch.abs(self._logits_cumulative(self.quantiles) - self.target).sum()

    @torch.no_grad()
    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        host = {n: p.detach().to("cpu", torch.float32) for n, p in self.named_parameters(recurse=False)}
        q = host["quantiles"]
        med = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits(host, samples - 0.5)
        upper = self._logits(host, samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        cdf = _rows_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
        self._store_tables(cdf, (pmf_length + 2).numpy(), (-minima).numpy())
        self._dev_params = None
        return True

    def device_params(self, device):
        """[C][59] fp32 blob for csrc/entropy.hip: softplus(matrix), bias, tanh(factor), median."""
        key = (str(device), tuple(p._version for p in self.parameters(recurse=False)))
        if self._dev_params is None or self._dev_params[0] != key:
            with torch.no_grad():
                C = self.channels
                g = lambda n: getattr(self, n).detach().to("cpu", torch.float32)
                parts = [F.softplus(g(f"_matrix{i}")).reshape(C, -1) for i in range(5)]
                parts += [g(f"_bias{i}").reshape(C, -1) for i in range(5)]
                parts += [torch.tanh(g(f"_factor{i}")).reshape(C, -1) for i in range(4)]
                parts += [g("quantiles")[:, 0, 1:2]]
                blob = torch.cat(parts, 1).contiguous()
                assert blob.shape == (C, N.EB_PARAMS_PER_CH)
                self._dev_params = (key, blob.to(device))
        return self._dev_params[1]

    # ---- device side ----------------------------------------------------------------------------
    def quantize(self, z, want_symbols, z_hat=None, lik=None, sym_out=None):
        """z: Act.  Returns int32 device symbols in [C][h][w] order (or None)."""
        return ops.eb_process(z, self.device_params(z.device), want_symbols, z_hat, lik, sym_out)

    def encode_symbols(self, symbols_host, B, C, h, w):
        """One rANS string per batch item, CDF index = channel (EntropyBottleneck.compress, image_model.py:206)."""
        tables = self.host_tables()
        idx = np.repeat(np.arange(C, dtype=np.int32), h * w)
        sym = symbols_host.reshape(B, C * h * w)
        out = []
        for b in range(B):
            enc = ops.RansEncoder()
            enc.encode_with_indexes(sym[b], idx, tables)
            out.append(enc.flush())
        return out

    def decode_symbols(self, strings, h, w):
        tables = self.host_tables()
        C = self.channels
        idx = np.repeat(np.arange(C, dtype=np.int32), h * w)
        return np.stack([ops.RansDecoder(s).decode_stream(idx, tables) for s in strings], 0)

    def compress(self, x):
        """Reference-style API: NCHW tensor -> list of byte strings."""
        z = to_act(x)
        sym = self.quantize(z, True)
        return self.encode_symbols(sym.cpu().numpy(), z.B, z.C, z.H, z.W)

    def decompress(self, strings, size):
        h, w = int(size[0]), int(size[1])
        dev = self.quantiles.device
        sym = torch.from_numpy(self.decode_symbols(strings, h, w).reshape(-1)).to(dev)
        z_hat = ops.new_act(len(strings), h, w, self.channels, torch.float32, dev)
        return ops.eb_dequantize(sym, self.device_params(dev), z_hat).nchw()

    def forward(self, x, training=None):
        """Eval-mode forward -> (z_hat, likelihood) as NCHW tensors (image_model.py:155)."""
        z = to_act(x)
        z_hat = ops.new_act(z.B, z.H, z.W, z.C, torch.float32, z.device)
        lik = ops.new_act(z.B, z.H, z.W, z.C, torch.float32, z.device)
        self.quantize(z, False, z_hat, lik)
        return z_hat.nchw(), lik.nchw()


class GaussianConditional(_EntropyModel):
    """Conditional Gaussian model (A.3): scale_bound 0.11, tail_mass 1e-9, 64 log-spaced scale levels."""

    def __init__(self, scale_table, scale_bound=0.11, tail_mass=1e-9):
        super().__init__()
        self.tail_mass = float(tail_mass)
        self.register_buffer("scale_table", torch.Tensor() if scale_table is None else torch.Tensor(tuple(float(s) for s in scale_table)))
        self.lower_bound_scale = _Bound(scale_bound)
        self._dev_table = None

    @torch.no_grad()
    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.scale_table.device
        self.scale_table = torch.Tensor(tuple(float(s) for s in scale_table)).to(dev)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        from scipy.stats import norm
        table = self.scale_table.detach().to("cpu", torch.float32)
        phi_c = lambda v: 0.5 * torch.erfc(-(2 ** -0.5) * v)
        multiplier = -norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(pmf_length.max())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        s = table.unsqueeze(1)
        upper, lower = phi_c((0.5 - samples) / s), phi_c((-0.5 - samples) / s)
        cdf = _rows_to_cdf((upper - lower).numpy(), (2 * lower[:, :1]).numpy(), pmf_length.numpy(), max_length)
        self._store_tables(cdf, (pmf_length + 2).numpy(), (-pmf_center).numpy())
        self._dev_table = None

    def device_table(self, device):
        if self._dev_table is None or self._dev_table.device != torch.device(device):
            self._dev_table = self.scale_table.detach().to(device, torch.float32).contiguous()
        return self._dev_table

    def build_indexes(self, scales):
        """Reference-style API (image_model.py:237): NCHW scales -> int32 NCHW indexes."""
        s = to_act(scales)
        _, idx = ops.gc_quantize(None, s, s, self.device_table(s.device), want_symbols=False, want_indexes=True)
        return idx.view(s.B, s.C, s.H, s.W)

    def forward(self, inputs, scales, means=None, training=None):
        """Eval-mode forward -> (y_hat, likelihood) NCHW (image_model.py:181)."""
        y, sc = to_act(inputs), to_act(scales)
        mu = to_act(means) if means is not None else Act(torch.zeros_like(y.buf))
        y_hat = ops.new_act(y.B, y.H, y.W, y.C, torch.float32, y.device)
        lik = ops.new_act(y.B, y.H, y.W, y.C, torch.float32, y.device)
        ops.gc_likelihood(y, mu, sc, y_hat, lik)
        return y_hat.nchw(), lik.nchw()
