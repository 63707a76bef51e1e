"""MI355X-native `ChannelSplitICIP2020ResB` (reference: image_model.py:41-328): analysis/synthesis transforms,
hyperprior and the 8-slice channel-conditional entropy model of the motion-vector and residual codecs.

API, attribute paths and bitstream format follow the reference; the execution does not:
  * transforms run as fused MFMA conv launches on NHWC slices (`modules.Conv2d.run`);
  * the slice loop never concatenates: supports are (latent, y_hat[:k*d]) segment lists read in place,
    `y_hat = q + mu` and `+ 0.5*tanh(lrp)` are written straight into the y_hat buffer;
  * ENCODE needs one device->host transfer for all 8 slices (symbols/indexes accumulate on the device);
  * DECODE runs the scan at its true dependency depth of 5, not 8: slices 4..7 depend only on slices 0..3
    (`y_hat_slices[:4]`, image_model.py:227), so their parameter nets and their range-decoder call are batched;
  * the encoder derives y_hat/z_hat directly (`q + mu`) instead of entropy-decoding its own output
    (image_model.py:207, video_model.py:141) -- identical values, since every kernel involved is deterministic.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .entropy_models import EntropyBottleneck, GaussianConditional
from .modules import Conv2d, Deconv5x5s2, ResBottleneckBlock, conv, get_scale_table, subpel_conv3x3
from .ops import Act, to_act

F32 = torch.float32
_CODER_POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix="dsvc-rans")   # ctypes calls release the GIL


class PendingStrings:
    """Range coding of one codec's symbols running on a worker thread (it first waits for the async device->host
    copy of the symbol planes); `.result()` joins and returns the reference-style {"strings", "shape"} dict."""

    def __init__(self, future):
        self._future = future

    def result(self):
        return self._future.result()



def conv3x3(cin, cout, stride=1):
    """compressai.layers.conv3x3."""
    return Conv2d(cin, cout, 3, stride, padding=1)


def deconv(cin, cout, kernel_size=5, stride=2):
    """compressai.models.utils.deconv."""
    assert (kernel_size, stride) == (5, 2)
    return Deconv5x5s2(cin, cout)


def _gelu_stack(widths):
    """conv3x3(s1) stack with GELU between layers, keyed 0,2,4,... like the reference's nn.Sequential."""
    layers = []
    for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        layers.append(conv(a, b, stride=1, kernel_size=3))
        if i < len(widths) - 2:
            layers.append(nn.GELU())
    return nn.Sequential(*layers)


def _run_gelu_stack(stack, segments, out=None, **last_epilogue):
    convs = [m for m in stack if isinstance(m, Conv2d)]
    x = segments
    for c in convs[:-1]:
        x = c.run(x, act=N.ACT_GELU, out_dtype=F32)
    return convs[-1].run(x, out, out_dtype=F32, **last_epilogue)


class ChannelSplitICIP2020ResB(nn.Module):
    def __init__(self, in_ch=3, N=192, out_ch=3):
        super().__init__()
        self.N = N
        self.num_slices = 8
        self.max_support_slices = 4
        d = self.N // self.num_slices
        if d * self.num_slices != self.N:
            raise ValueError(f"Slice do not evenly divide latent depth ({self.N}/{self.num_slices})")
        self.slice_depth = d

        def down(first_in):
            mods, cin = [], first_in
            for stage in range(4):
                mods.append(conv(cin, N, kernel_size=5, stride=2))
                cin = N
                if stage < 3:
                    mods += [ResBottleneckBlock(N) for _ in range(3)]
            return nn.Sequential(*mods)

        def up(last_out):
            mods = []
            for stage in range(4):
                mods.append(deconv(N, N if stage < 3 else last_out, kernel_size=5, stride=2))
                if stage < 3:
                    mods += [ResBottleneckBlock(N) for _ in range(3)]
            return nn.Sequential(*mods)

        self.g_a, self.g_s = down(in_ch), up(out_ch)
        self.h_a = nn.Sequential(conv3x3(N, N), nn.GELU(), conv3x3(N, N, stride=2), nn.GELU(),
                                 conv3x3(N, N), nn.GELU(), conv3x3(N, N, stride=2))
        hyper_synthesis = lambda: nn.Sequential(subpel_conv3x3(N, N, 2), nn.GELU(), conv3x3(N, N), nn.GELU(),
                                                subpel_conv3x3(N, N, 2), nn.GELU(), conv3x3(N, N))
        self.h_mean_s, self.h_scale_s = hyper_synthesis(), hyper_synthesis()
        k = self.max_support_slices
        self.cc_mean_transforms = nn.ModuleList(_gelu_stack([N + d * min(i, k), N, 64, 32, d]) for i in range(self.num_slices))
        self.cc_scale_transforms = nn.ModuleList(_gelu_stack([N + d * min(i, k), N, 64, 32, d]) for i in range(self.num_slices))
        self.lrp_transforms = nn.ModuleList(_gelu_stack([N + d * min(i + 1, k + 1), N, N // 2, d]) for i in range(self.num_slices))
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)

    # ------------------------------------------------------------------------------------------
    # transforms on slices
    # ------------------------------------------------------------------------------------------
    def _g_a(self, segments):
        x = segments
        last = len(self.g_a) - 1
        for i, m in enumerate(self.g_a):
            if isinstance(m, Conv2d):
                x = m.run(x, out_dtype=F32 if i == last else None)   # the latent y stays fp32
            else:
                x = m.run(x)
        return x

    def _g_s(self, y_hat, out=None, out_dtype=None):
        x = y_hat
        last = len(self.g_s) - 1
        for i, m in enumerate(self.g_s):
            if isinstance(m, Deconv5x5s2):
                x = m.run(x, out if i == last else None, out_dtype=out_dtype if i == last else None)
            else:
                x = m.run(x)
        return x

    def _h_a(self, y):
        x = self.h_a[0].run(y, act=N.ACT_GELU, out_dtype=F32)
        x = self.h_a[2].run(x, act=N.ACT_GELU, out_dtype=F32)
        x = self.h_a[4].run(x, act=N.ACT_GELU, out_dtype=F32)
        return self.h_a[6].run(x, out_dtype=F32)

    @staticmethod
    def _h_s(net, z_hat):
        x = net[0].run(z_hat, act=N.ACT_GELU, out_dtype=F32)
        x = net[2].run(x, act=N.ACT_GELU, out_dtype=F32)
        x = net[4].run(x, act=N.ACT_GELU, out_dtype=F32)
        return net[6].run(x, out_dtype=F32)

    def _support(self, latent, y_hat, i):
        k = min(i, self.max_support_slices)
        return [latent] if k == 0 else [latent, y_hat.ch(0, k * self.slice_depth)]

    def _slice_params(self, i, means, scales, y_hat):
        mu = _run_gelu_stack(self.cc_mean_transforms[i], self._support(means, y_hat, i))
        sc = _run_gelu_stack(self.cc_scale_transforms[i], self._support(scales, y_hat, i))
        return mu, sc

    def _lrp(self, i, means, y_hat):
        """y_hat_i += 0.5*tanh(lrp_i(cat[mean_support, y_hat_i])) in place (image_model.py:185-188,244-247,292-295)."""
        d = self.slice_depth
        cur = y_hat.ch(i * d, d)
        if i < self.max_support_slices:
            segs = [means, y_hat.ch(0, (i + 1) * d)]
        else:
            segs = [means, y_hat.ch(0, self.max_support_slices * d), cur]
        _run_gelu_stack(self.lrp_transforms[i], segs, out=cur, act=N.ACT_HALF_TANH, res1=cur)

    def _hyper(self, z_hat):
        return self._h_s(self.h_scale_s, z_hat), self._h_s(self.h_mean_s, z_hat)

    # ------------------------------------------------------------------------------------------
    # slice-level entry points (used by video_model.DeepSVC)
    # ------------------------------------------------------------------------------------------
    def run_forward(self, segments, x_hat_out=None, x_hat_dtype=None):
        """Estimate path (image_model.py:151-199) -> (x_hat Act, y_lik Act, z_lik Act)."""
        y = self._g_a(segments)
        z = self._h_a(y)
        dev = y.device
        z_hat = ops.new_act(z.B, z.H, z.W, z.C, F32, dev)
        z_lik = ops.new_act(z.B, z.H, z.W, z.C, F32, dev)
        self.entropy_bottleneck.quantize(z, False, z_hat, z_lik)
        scales, means = self._hyper(z_hat)
        y_hat = ops.new_act(y.B, y.H, y.W, y.C, F32, dev)
        y_lik = ops.new_act(y.B, y.H, y.W, y.C, F32, dev)
        d = self.slice_depth
        for i in range(self.num_slices):
            mu, sc = self._slice_params(i, means, scales, y_hat)
            ops.gc_likelihood(y.ch(i * d, d), mu, sc, y_hat.ch(i * d, d), y_lik.ch(i * d, d))
            self._lrp(i, means, y_hat)
        return self._g_s(y_hat, x_hat_out, x_hat_dtype), y_lik, z_lik

    def run_compress(self, segments, want_x_hat=False, x_hat_dtype=None, defer=False):
        """-> ({"strings", "shape"} or PendingStrings when defer=True, x_hat Act or None)."""
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        y = self._g_a(segments)
        z = self._h_a(y)
        dev = y.device
        z_hat = ops.new_act(z.B, z.H, z.W, z.C, F32, dev)
        nz, ny = z.B * z.C * z.H * z.W, y.B * y.C * y.H * y.W
        d = self.slice_depth
        ns = y.B * d * y.H * y.W
        staging = torch.empty(nz + 2 * ny, dtype=torch.int32, device=dev)   # [z symbols | y symbols | y indexes]
        eb.quantize(z, True, z_hat, sym_out=staging[:nz])
        scales, means = self._hyper(z_hat)
        y_hat = ops.new_act(y.B, y.H, y.W, y.C, F32, dev)
        table = gc.device_table(dev)
        for i in range(self.num_slices):
            mu, sc = self._slice_params(i, means, scales, y_hat)
            ops.gc_quantize(y.ch(i * d, d), mu, sc, table, True, True, y_hat.ch(i * d, d),
                            sym_out=staging[nz + i * ns: nz + (i + 1) * ns],
                            idx_out=staging[nz + ny + i * ns: nz + ny + (i + 1) * ns])
            self._lrp(i, means, y_hat)
        # Symbols leave the device through one async copy into pinned memory; the host range coder runs on a worker
        # thread behind the copy's event, so the caller can keep queuing GPU work (synthesis transform, MC, ...).
        host = torch.empty(staging.numel(), dtype=torch.int32, pin_memory=True)
        host.copy_(staging, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        zB, zC, zH, zW = z.B, z.C, z.H, z.W

        def encode():
            done.synchronize()
            packed = host.numpy()
            z_strings = eb.encode_symbols(packed[:nz], zB, zC, zH, zW)
            enc = ops.RansEncoder()
            enc.encode_with_indexes(packed[nz:nz + ny], packed[nz + ny:], gc.host_tables())
            return {"strings": [[enc.flush()], z_strings], "shape": (zH, zW)}

        pending = PendingStrings(_CODER_POOL.submit(encode))
        x_hat = self._g_s(y_hat, None, x_hat_dtype) if want_x_hat else None
        return (pending if defer else pending.result()), x_hat

    def run_decompress(self, strings, shape, device, x_hat_out=None, x_hat_dtype=None):
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        h4, w4 = int(shape[0]), int(shape[1])
        B = len(strings[1])
        z_sym = torch.from_numpy(eb.decode_symbols(strings[1], h4, w4).reshape(-1)).to(device)
        z_hat = ops.eb_dequantize(z_sym, eb.device_params(device), ops.new_act(B, h4, w4, self.N, F32, device))
        scales, means = self._hyper(z_hat)
        h, w, d = h4 * 4, w4 * 4, self.slice_depth
        y_hat = ops.new_act(B, h, w, self.N, F32, device)
        table, tables = gc.device_table(device), gc.host_tables()
        dec = ops.RansDecoder(strings[0][0])
        k = self.max_support_slices
        groups = [[i] for i in range(min(k, self.num_slices))] + ([list(range(k, self.num_slices))] if self.num_slices > k else [])
        n = B * d * h * w
        gmax = max(len(g) for g in groups)
        idx_dev = torch.empty(gmax * n, dtype=torch.int32, device=device)
        sym_dev = torch.empty(gmax * n, dtype=torch.int32, device=device)
        idx_host = torch.empty(gmax * n, dtype=torch.int32, pin_memory=True)     # pinned: async copies, no staging hop
        sym_host = torch.empty(gmax * n, dtype=torch.int32, pin_memory=True)
        stream = torch.cuda.current_stream(device)
        for group in groups:
            mus = []
            for j, i in enumerate(group):        # parameter nets of a dependency level are queued together
                mu, sc = self._slice_params(i, means, scales, y_hat)
                ops.gc_quantize(None, mu, sc, table, False, True, idx_out=idx_dev[j * n:(j + 1) * n])
                mus.append(mu)
            m = len(group) * n
            idx_host[:m].copy_(idx_dev[:m], non_blocking=True)
            stream.synchronize()                 # the one sync point of this level
            dec.decode_stream(idx_host.numpy()[:m], tables, out=sym_host.numpy()[:m])
            sym_dev[:m].copy_(sym_host[:m], non_blocking=True)
            for j, i in enumerate(group):
                ops.gc_dequantize(sym_dev[j * n:(j + 1) * n], mus[j], y_hat.ch(i * d, d))
                self._lrp(i, means, y_hat)
        return self._g_s(y_hat, x_hat_out, x_hat_dtype)

    # ------------------------------------------------------------------------------------------
    # reference-style API (NCHW tensors)
    # ------------------------------------------------------------------------------------------
    def forward(self, x):
        x_hat, y_lik, z_lik = self.run_forward([to_act(x)], x_hat_dtype=F32)
        return {"x_hat": x_hat.nchw(), "likelihoods": {"y": y_lik.nchw(), "z": z_lik.nchw()}}

    def compress(self, x):
        enc, _ = self.run_compress([to_act(x)])
        return enc

    def decompress(self, strings, shape):
        dev = self.entropy_bottleneck.quantiles.device
        return {"x_hat": self.run_decompress(strings, shape, dev, x_hat_dtype=F32).nchw()}

    def load_state_dict(self, state_dict, strict=True):
        """reference image_model.py:304-317: size the (initially empty) table buffers from the checkpoint first."""
        for mod, prefix, names in ((self.entropy_bottleneck, "entropy_bottleneck", ("_quantized_cdf", "_offset", "_cdf_length")),
                                   (self.gaussian_conditional, "gaussian_conditional",
                                    ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"))):
            for n in names:
                key = f"{prefix}.{n}"
                if key in state_dict:
                    cur = getattr(mod, n)
                    setattr(mod, n, torch.empty(state_dict[key].shape, dtype=cur.dtype, device=cur.device))
            mod._host_tables = None
        return super().load_state_dict(state_dict, strict=strict)

    def update(self, scale_table=None, force=False):
        """reference image_model.py:319-324."""
        updated = self.entropy_bottleneck.update(force=force)
        if scale_table is None:
            scale_table = get_scale_table()
        updated |= self.gaussian_conditional.update_scale_table(scale_table, force=force)
        return updated

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))
