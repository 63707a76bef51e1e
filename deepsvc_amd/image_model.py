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
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .entropy_models import EntropyBottleneck, GaussianConditional, ms_pack, ms_unpack
from .modules import Conv2d, Deconv5x5s2, ResBottleneckBlock, conv, get_scale_table, subpel_conv3x3
from .ops import Act, to_act

F32 = torch.float32
ITEM_LRP = __import__("os").environ.get("DSVC_NO_ITEM_LRP", "") == ""     # A/B switch: LRP stacks of the last level + the hyper-synthesis pair as batch items
_CODER_POOL = ThreadPoolExecutor(max_workers=12, thread_name_prefix="dsvc-rans")   # ctypes calls release the GIL; 2 per GOP lane, up to 6 lanes


class PendingStrings(dict):
    """The reference-style {"strings": [[y], [z...]], "shape": (h, w)} dict of one codec, whose range coding is still
    running on a worker thread (behind the async device->host copy of the symbol planes).  Any access joins first, so
    callers see a plain dict -- but a caller that does not look inside (DeepSVC.decompress hands the residual
    encoding straight to its decode thread) lets the coding overlap with whatever it does next."""

    def __init__(self, future):
        super().__init__()
        self._future = future
        self._lock = threading.Lock()
        self._error = None

    def _join(self):
        """Idempotent and thread-safe: the first caller waits for the coder and fills the dict; concurrent callers wait on the
        lock and then see the filled dict; a coder exception is kept and re-raised on every later access."""
        if self._future is not None or self._error is not None:
            with self._lock:
                if self._error is not None:
                    raise self._error
                if self._future is not None:
                    try:
                        super().update(self._future.result())
                    except BaseException as e:
                        self._error = e
                        raise
                    finally:
                        self._future = None
        return self

    def result(self):
        return self._join()

    def __getitem__(self, k):
        return dict.__getitem__(self._join(), k)

    def __contains__(self, k):
        return dict.__contains__(self._join(), k)

    def __iter__(self):
        return dict.__iter__(self._join())

    def __len__(self):
        return dict.__len__(self._join())

    def keys(self):
        return dict.keys(self._join())

    def values(self):
        return dict.values(self._join())

    def items(self):
        return dict.items(self._join())

    def get(self, k, default=None):
        return dict.get(self._join(), k, default)

    def __repr__(self):
        return dict.__repr__(self._join())

    def __eq__(self, other):
        return dict.__eq__(self._join(), other)

    def copy(self):
        return dict(self._join().items())

    __hash__ = None


class _ScanPlan:
    """Static device/pinned buffers of one codec at one latent shape: graph-captured scan segments address them."""

    def __init__(self, codec, B, h, w, device):
        Nc, Mc, d = codec.N, codec.M, codec.slice_depth          # hyper-latent z: N channels; latent y: M channels
        self.device, self.epoch, self.graphs, self.programs = device, -1, {}, {}
        act = lambda H, W, C: ops.new_act(B, H, W, C, F32, device)
        self.y, self.y_hat = act(h, w, Mc), act(h, w, Mc)
        # (means, scales) and every slice's (mu_i, sigma_i) sit in ONE buffer each, pair after pair: the mean and the scale
        # stack of a scan level run as batch items of single launches (ops.conv2d items=...), which read / write them at a
        # fixed item stride
        # (a codec whose parameter nets are also conditioned on a side feature -- semantic_layer.cFeatureCompress -- keeps that
        # feature in `latent_extra` further channels of the same pixel rows, so cat[latent, side, supports] stays <= 3 segments)
        extra = getattr(codec, "latent_extra", 0)
        self.pair = ops.new_act(2 * B, h, w, Mc + extra, F32, device)
        self.means, self.scales = ops.Act(self.pair.buf[:B], 0, Mc), ops.Act(self.pair.buf[B:], 0, Mc)
        self.z_hat = act(h // 4, w // 4, Nc)
        self.musc = ops.new_act(2 * B * codec.num_slices, h, w, d, F32, device)
        self.mus = [ops.Act(self.musc.buf[2 * i * B:(2 * i + 1) * B]) for i in range(codec.num_slices)]
        self.scs = [ops.Act(self.musc.buf[(2 * i + 1) * B:(2 * i + 2) * B]) for i in range(codec.num_slices)]
        self.B = B
        self.nz, self.ny, self.ns = B * Nc * (h // 4) * (w // 4), B * Mc * h * w, B * d * h * w
        k = codec.max_support_slices
        self.groups = [[i] for i in range(min(k, codec.num_slices))] + \
                      ([list(range(k, codec.num_slices))] if codec.num_slices > k else [])
        gmax = max(len(g) for g in self.groups)
        i32 = lambda n, **kw: torch.empty(n, dtype=torch.int32, **kw)
        self.staging = i32(self.nz + 2 * self.ny, device=device)      # encoder: [z symbols | y symbols | y indexes]
        self.z_sym, self.z_sym_host = i32(self.nz, device=device), i32(self.nz, pin_memory=True)
        self.idx_dev, self.sym_dev = i32(gmax * self.ns, device=device), i32(gmax * self.ns, device=device)
        self.idx_host, self.sym_host = i32(gmax * self.ns, pin_memory=True), i32(gmax * self.ns, pin_memory=True)


def rename_legacy_keys(state_dict):
    """Key spellings of older checkpoints (reference utils.py:105-125 `rename_key`, applied by `load_pretrained`): a
    DataParallel `module.` prefix and EntropyBottleneck parameters kept in nn.ParameterLists
    (`entropy_bottleneck._matrices.N / _biases.N / _factors.N` -> `_matrixN / _biasN / _factorN`), at any nesting depth."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[7:]
        for old, new in (("entropy_bottleneck._matrices.", "entropy_bottleneck._matrix"),
                         ("entropy_bottleneck._biases.", "entropy_bottleneck._bias"),
                         ("entropy_bottleneck._factors.", "entropy_bottleneck._factor")):
            if old in k:
                k = k.replace(old, new)
        out[k] = v
    return out


def conv3x3(cin, cout, stride=1):
    """compressai.layers.conv3x3."""
    return Conv2d(cin, cout, 3, stride, padding=1)


def deconv(cin, cout, kernel_size=5, stride=2):
    """compressai.models.utils.deconv."""
    assert (kernel_size, stride) == (5, 2)
    return Deconv5x5s2(cin, cout)


def _gelu_stack(widths, act=nn.GELU):
    """conv3x3(s1) stack with an activation between layers, keyed 0,2,4,... like the reference's nn.Sequential."""
    layers = []
    for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        layers.append(conv(a, b, stride=1, kernel_size=3))
        if i < len(widths) - 2:
            layers.append(act())
    return nn.Sequential(*layers)


def _run_stack(stack, segments, out=None, hidden_act=N.ACT_GELU, hidden_slope=0.01, **last_epilogue):
    """conv3x3 stack of the slice scan: `hidden_act` after every conv but the last, whose fused epilogue is the caller's."""
    convs = [m for m in stack if isinstance(m, Conv2d)]
    x = segments
    for c in convs[:-1]:
        # hidden activations in the storage type: in f16 mode they would be rounded to f16 when staged for the MFMA
        # anyway (same single rounding), and f16 storage puts the next layer on the conv's fast staging path
        x = c.run(x, act=hidden_act, slope=hidden_slope, out_dtype=ops.RT.storage)
    return convs[-1].run(x, out, out_dtype=F32, **last_epilogue)


class ChannelSplitICIP2020ResB(nn.Module):
    _stack_act = (N.ACT_GELU, 0.01)      # activation between the convs of the cc_mean / cc_scale / lrp stacks
    # Range coder of the y string: "rans" = compressai 1.2.1's format on the host (the reference's bitstream, image_model.py:253-254,273-274,288);
    # "rans_ms" = the GPU-decodable multi-stream coder (csrc/rans_ms.hip): same symbols, same tables, another -- LABELLED --
    # string layout (entropy_models.ms_pack); a level decodes in one launch between the scan kernels, the decode chain has no host round trip.
    # The z string stays in the reference's format in both modes (32 k symbols, decoded before any GPU work of the chain).
    coder = "rans"
    ms_symbols_per_stream = 256

    def __init__(self, in_ch=3, N=192, out_ch=3):
        super().__init__()
        self.N = self.M = N              # hyper-latent and latent channel counts (they differ in ICIP2020ResB)
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
    def _g_a(self, segments, out=None):
        x = segments
        last = len(self.g_a) - 1
        for i, m in enumerate(self.g_a):
            if isinstance(m, Conv2d):
                x = m.run(x, out if i == last else None, out_dtype=F32 if i == last else None)   # the latent y stays fp32
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

    def _h_s(self, net, z_hat, out=None):
        x = net[0].run(z_hat, act=N.ACT_GELU, out_dtype=F32)
        x = net[2].run(x, act=N.ACT_GELU, out_dtype=F32)
        x = net[4].run(x, act=N.ACT_GELU, out_dtype=F32)
        return net[6].run(x, out, out_dtype=F32)

    def _stack(self, stack, segments, out=None, **last_epilogue):
        return _run_stack(stack, segments, out, self._stack_act[0], self._stack_act[1], **last_epilogue)

    latent_extra = 0         # side-feature channels stored behind the latent means / scales (see _ScanPlan)

    def _latent_seg(self, latent):
        """The first segment of every parameter-net input: the hyper-prior's means (or scales), plus the side feature kept
        behind them when the codec has one."""
        return latent if self.latent_extra == 0 else Act(latent.buf, latent.c0, latent.C + self.latent_extra)

    def _support(self, latent, y_hat, i):
        k = min(i, self.max_support_slices)
        latent = self._latent_seg(latent)
        return [latent] if k == 0 else [latent, y_hat.ch(0, k * self.slice_depth)]

    def _slice_params(self, i, means, scales, y_hat, mu_out=None, sc_out=None):
        """mu_i, sigma_i: two independent 4-conv stacks -> forked onto two streams."""
        return ops.run_parallel([
            lambda: self._stack(self.cc_mean_transforms[i], self._support(means, y_hat, i), out=mu_out),
            lambda: self._stack(self.cc_scale_transforms[i], self._support(scales, y_hat, i), out=sc_out)])

    def _item_stacks(self, group, segC, device):
        """Per layer, the packed weights of the 2*len(group) mean / scale stacks of one dependency level, item order
        (mean_i, scale_i) for i in group.  Rebuilt (and this codec's graphs invalidated) when a layer has been repacked."""
        stacks = [[m for m in st[i] if isinstance(m, Conv2d)] for i in group for st in (self.cc_mean_transforms, self.cc_scale_transforms)]
        cache = self.__dict__.setdefault("_item_cache", {})
        key = (tuple(group), tuple(segC), ops.RT.compute, str(device))
        out = []
        for l in range(len(stacks[0])):
            packs = [st[l]._packed(list(segC) if l == 0 else [st[l].in_channels], device) for st in stacks]
            ipc = cache.get(key + (l,))
            if ipc is None or any(a is not b for a, b in zip(ipc.sources, packs)):
                ipc = cache[key + (l,)] = ops.ItemPackedConv(packs)
                self._epoch_ref()[0] += 1
            out.append(ipc)
        return out

    def _group_params(self, ws, group):
        """mu_i, sigma_i for every slice of one dependency level: 2*len(group) independent 4-conv stacks over the same
        support.  With one image in flight they run as the batch items of ONE launch per layer (include/dsvc.h: batch_inner)
        -- a level costs 4 small launches instead of 8..32; the launches are latency-bound, so the items are free."""
        k = min(group[0], self.max_support_slices)
        segs = self._support(ws.means, ws.y_hat, group[0])
        if ops.ITEM_SCAN[0] and ws.B == 1 and ops.items_ok(segs + [ws.scales]) and \
                all(min(i, self.max_support_slices) == k for i in group) and group == list(range(group[0], group[0] + len(group))):
            G, img = 2 * len(group), ws.means.npix * ws.means.ld * 4
            ipcs = self._item_stacks(group, [a.C for a in segs], ws.device)
            hidden, slope = self._stack_act
            x = ops.conv2d(ipcs[0], segs, act=hidden, slope=slope, out_dtype=ops.RT.storage,
                           items=(2, [(img, 0)] + [(0, 0)] * (len(segs) - 1)))           # item b: means|scales = b % 2, shared support
            for ipc in ipcs[1:-1]:
                x = ops.conv2d(ipc, [x], act=hidden, slope=slope, out_dtype=ops.RT.storage,
                               items=(G, [(x.H * x.W * x.ld * x.buf.element_size(), 0)]))
            out = ops.Act(ws.musc.buf[2 * group[0]:2 * group[0] + G])
            ops.conv2d(ipcs[-1], [x], out, out_dtype=F32, items=(G, [(x.H * x.W * x.ld * x.buf.element_size(), 0)]))
            return
        fns = []
        for i in group:
            fns.append(lambda i=i: self._stack(self.cc_mean_transforms[i], self._support(ws.means, ws.y_hat, i), out=ws.mus[i]))
            fns.append(lambda i=i: self._stack(self.cc_scale_transforms[i], self._support(ws.scales, ws.y_hat, i), out=ws.scs[i]))
        ops.run_parallel(fns)

    def _packed_items(self, tag, stacks, segC, device):
        """Per layer, the ItemPackedConv of several shape-identical conv stacks (`stacks`: lists of Conv2d / Sequential(Conv2d, PixelShuffle)
        wrappers are unwrapped by the caller).  Cached; rebuilt (and this codec's graphs invalidated) when a layer has been repacked."""
        cache = self.__dict__.setdefault("_item_cache", {})
        key = (tag, tuple(segC), ops.RT.compute, str(device))
        out = []
        for l in range(len(stacks[0])):
            packs = [st[l]._packed(list(segC) if l == 0 else [st[l].in_channels], device) for st in stacks]
            ipc = cache.get(key + (l,))
            if ipc is None or any(a is not b for a, b in zip(ipc.sources, packs)):
                ipc = cache[key + (l,)] = ops.ItemPackedConv(packs)
                self._epoch_ref()[0] += 1
            out.append(ipc)
        return out

    def _lrp_group(self, ws, group):
        """The latent-residual-prediction stacks of one dependency level.  The slices of the LAST level (4 of the 8, 5 of the I-frame
        codec's 10) share their support and differ only in their own y_hat slice, which is also where each writes: they run as the
        batch items of one launch per layer (include/dsvc.h: batch_inner with out_bstride) -- 3 launches instead of 12 / 15."""
        d, k = self.slice_depth, self.max_support_slices
        y_hat, means = ws.y_hat, self._latent_seg(ws.means)
        esz = y_hat.buf.element_size()
        cur0 = y_hat.ch(group[0] * d, d)
        segs = [means, y_hat.ch(0, k * d), cur0]
        ok = (ops.ITEM_SCAN[0] and ITEM_LRP and ws.B == 1 and len(group) > 1 and group[0] >= k and group == list(range(group[0], group[0] + len(group)))
              and ops.items_ok(segs) and (d * esz) % 16 == 0)
        if not ok:
            ops.run_parallel([(lambda i=i: self._lrp(i, ws.means, y_hat)) for i in group])
            return
        G = len(group)
        stacks = [[m for m in self.lrp_transforms[i] if isinstance(m, Conv2d)] for i in group]
        ipcs = self._packed_items(("lrp", tuple(group)), stacks, [a.C for a in segs], ws.device)
        hidden, slope = self._stack_act
        x = ops.conv2d(ipcs[0], segs, act=hidden, slope=slope, out_dtype=ops.RT.storage, items=(G, [(0, 0), (0, 0), (d * esz, 0)]))
        for ipc in ipcs[1:-1]:
            x = ops.conv2d(ipc, [x], act=hidden, slope=slope, out_dtype=ops.RT.storage, items=(G, [(x.H * x.W * x.ld * x.buf.element_size(), 0)]))
        ops.conv2d(ipcs[-1], [x], cur0, act=N.ACT_HALF_TANH, res1=cur0, out_dtype=F32,
                   items=(G, [(x.H * x.W * x.ld * x.buf.element_size(), 0)]), item_out=(d * esz, d * esz))

    def _hyper_into(self, ws):
        """h_scale_s(z_hat) -> ws.scales and h_mean_s(z_hat) -> ws.means.  The two nets have identical shapes and the two outputs are
        neighbours in one buffer (`ws.pair`): with one image in flight they run as two batch items of one launch per layer."""
        nets = [self.h_mean_s, self.h_scale_s]                        # item order = ws.pair's: means | scales
        convs = [[m[0] if isinstance(m, nn.Sequential) else m for m in net if isinstance(m, (Conv2d, nn.Sequential))] for net in nets]
        plain = type(self)._h_s is ChannelSplitICIP2020ResB._h_s and all(isinstance(c, Conv2d) for st in convs for c in st) and len(convs[0]) == 4
        if not (plain and ops.ITEM_SCAN[0] and ITEM_LRP and ws.B == 1 and ops.items_ok([ws.z_hat])):
            ops.run_parallel([lambda: self._h_s(self.h_scale_s, ws.z_hat, ws.scales), lambda: self._h_s(self.h_mean_s, ws.z_hat, ws.means)])
            return
        ipcs = self._packed_items(("h_s",), convs, [ws.z_hat.C], ws.device)
        shuf = [2 if isinstance(m, nn.Sequential) else 1 for m in self.h_mean_s if isinstance(m, (Conv2d, nn.Sequential))]
        x, first = ws.z_hat, True
        for l, ipc in enumerate(ipcs):
            last = l == len(ipcs) - 1
            stride = 0 if first else x.H * x.W * x.ld * x.buf.element_size()
            x = ops.conv2d(ipc, [x], ops.Act(ws.pair.buf, 0, ws.means.C) if last else None, act=N.ACT_NONE if last else N.ACT_GELU, shuffle=shuf[l],
                           out_dtype=F32, items=(2, [(stride, 0)]))
            first = False

    def _lrp(self, i, means, y_hat):
        """y_hat_i += 0.5*tanh(lrp_i(cat[mean_support, y_hat_i])) in place (image_model.py:185-188,244-247,292-295)."""
        d = self.slice_depth
        cur = y_hat.ch(i * d, d)
        means = self._latent_seg(means)
        if i < self.max_support_slices:
            segs = [means, y_hat.ch(0, (i + 1) * d)]
        else:
            segs = [means, y_hat.ch(0, self.max_support_slices * d), cur]
        self._stack(self.lrp_transforms[i], segs, out=cur, act=N.ACT_HALF_TANH, res1=cur)

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

    # ------------------------------------------------------------------------------------------
    # scan plans: static workspace + HIP graphs of the launch-bound scan segments
    # ------------------------------------------------------------------------------------------
    def _epoch_ref(self):
        """Packing epoch of THIS codec: bumped by its own conv layers when their weights are (re)packed (modules.
        _bump_owner_epoch).  The scan graphs hold pointers to exactly those buffers, so a pack elsewhere -- another model
        instance running on a concurrent GOP lane, a lazily packed layer of the video model -- leaves them valid."""
        ref = self.__dict__.get("_pack_epoch")
        if ref is None:
            ref = self.__dict__["_pack_epoch"] = [0]
            for m in self.modules():
                if isinstance(m, (Conv2d, Deconv5x5s2)):
                    m.__dict__["_epoch_ref"] = ref
        return ref

    def _weights_fingerprint(self):
        """Sum of the autograd version counters of every parameter of this codec: moves when a weight is edited in place
        (`p.copy_`, `p.zero_`, optimiser steps) or replaced by load_state_dict.  Recorded scan programs and captured graphs hold
        pointers to PACKED copies of the weights and never pass through Conv2d._packed again, so this is what invalidates them
        (tests/test_gpu_model.py::test_weight_edit_invalidates_recorded_scan).  ~10 us per call."""
        ps = self.__dict__.get("_fp_params")
        if ps is None:
            ps = self.__dict__["_fp_params"] = [p for p in self.parameters()]
        # (+ the table generations of the two entropy models: a direct gaussian_conditional.update_scale_table() / entropy_bottleneck.update()
        #  replaces the device-side tables the recordings point to without touching a parameter)
        # a tuple, not an arithmetic mix: _TABLE_GEN is process-wide and unbounded, shifted sums would alias once a generation passes 2^20
        return (sum(p._version for p in ps), id(ps[0]) if ps else 0, self.entropy_bottleneck._table_gen, self.gaussian_conditional._table_gen)

    def _plan(self, B, h, w, device):
        key = (B, h, w, str(device), ops.RT.mode)
        plans = self.__dict__.setdefault("_plans", {})
        fp = self._weights_fingerprint()
        if self.__dict__.get("_fp_seen") != fp:          # weights changed since the last call: everything recorded is stale
            if "_fp_seen" in self.__dict__:
                self._epoch_ref()[0] += 1
            self.__dict__["_fp_seen"] = fp
        plan = plans.get(key)
        if plan is None or plan.epoch != self._epoch_ref()[0]:
            if plan is None:
                plan = plans[key] = _ScanPlan(self, B, h, w, device)
            plan.graphs.clear()                  # weights were repacked since capture: the graphs hold stale pointers
            plan.programs.clear()                # ... and so do the recorded scan steps
            plan.epoch = self._epoch_ref()[0]
        return plan

    def _step(self, plan, key, fn):
        """Run one scan segment (the encoder scan, or one dependency level of the decoder scan): a fixed sequence of ~20-90
        tiny launches on static buffers with no host interaction.
          * first execution: issued launch by launch from here while `ops.ScanRecorder` records the sequence;
          * afterwards: ONE call into the C library replays it (include/dsvc.h: dsvc_entropy_scan_step) -- no Python between
            the launches;
          * once captured (capture_scan_graphs), the same sequence is replayed as a HIP graph: the CPU cost per segment drops
            from ~30 us per launch to one graph launch."""
        if ops._rec() is not None:            # inside a recorded frame segment: the launches become part of THAT program
            fn()
            return
        valid = plan.epoch == self._epoch_ref()[0]
        g = plan.graphs.get(key) if ops.USE_GRAPHS[0] and valid else None
        if g is not None:
            g.replay()
            return
        if not (ops.SCAN_STEP[0] and valid) or ops.USE_FORKS[0]:
            fn()
            return
        prog = plan.programs.get(key)
        if prog is None:
            with ops.ScanRecorder() as rec:
                fn()                              # executed AND recorded
            if rec.poisoned is not None:
                plan.programs[key] = False        # holds an op without a scan-op form: this segment stays launch by launch
            elif self._epoch_ref()[0] == plan.epoch:      # nothing was (re)packed while recording
                plan.programs[key] = ops.ScanProgram(rec)
        elif prog is False:
            fn()
        else:
            prog.run()

    def scan_graphs_ready(self, B, h, w, device):
        key = (B, h, w, str(device), ops.RT.mode)
        plan = self.__dict__.get("_plans", {}).get(key)
        return plan is not None and plan.epoch == self._epoch_ref()[0] and all(k in plan.graphs for k in self._segments(plan))

    def capture_scan_graphs(self, B, h, w, device):
        """Capture the scan segments of this codec at latent shape (h, w).  Must be called from a quiescent point (no
        other thread issuing HIP work): DeepSVC does so at the entry of compress()/decompress() once the weight-packing
        epoch has been stable for a whole call.  Returns True when the graphs are in place."""
        if not ops.USE_GRAPHS[0]:
            return False
        ws = self._plan(B, h, w, device)
        steps = self._segments(ws)
        if all(k in ws.graphs for k in steps):
            return True
        for f in self.__dict__.get("_pending_futures", []):
            f.result()
        self.__dict__["_pending_futures"] = []
        for t in (ws.z_sym, ws.sym_dev, ws.idx_dev, ws.staging):
            t.zero_()
        ws.y.buf.zero_()
        epoch = self._epoch_ref()[0]
        use_graphs, ops.USE_GRAPHS[0] = ops.USE_GRAPHS[0], False
        try:
            for key, fn in steps.items():         # eager warm-up on dummy data: packs weights, sets kernel attributes, records
                self._step(ws, key, fn)           # the segment as a scan step (ops.ScanProgram)
        finally:
            ops.USE_GRAPHS[0] = use_graphs
        torch.cuda.synchronize(device)
        if self._epoch_ref()[0] != epoch:            # something was packed just now: pointers may move again, retry later
            self._plan(B, h, w, device)
            return False
        for key, fn in steps.items():
            g = torch.cuda.CUDAGraph()
            prog = ws.programs.get(key)
            with torch.cuda.graph(g):             # recorded, not executed: the scan step's launches become the graph's nodes
                if prog:
                    prog.run()
                else:
                    fn()
            ws.graphs[key] = g
        torch.cuda.synchronize(device)
        return True

    def _segments(self, ws):
        """key -> closure of every capturable scan segment on workspace `ws` (encoder scan, decoder levels)."""
        eb, device, groups = self.entropy_bottleneck, ws.device, ws.groups

        def dec_first():
            ops.eb_dequantize(ws.z_sym, eb.device_params(device), ws.z_hat)
            self._hyper_into(ws)
            self._dec_nets(ws, groups[0])

        def dec_level(group, nxt):
            def level():
                self._dec_finish(ws, group)
                if nxt is not None:
                    self._dec_nets(ws, nxt)
            return level

        steps = {"enc": lambda: self._enc_scan(ws), "dec0": dec_first}
        for k, group in enumerate(groups):
            steps[f"dec{k + 1}"] = dec_level(group, groups[k + 1] if k + 1 < len(groups) else None)
        return steps

    def _enc_scan(self, ws):
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        d, nz, ny, ns = self.slice_depth, ws.nz, ws.ny, ws.ns
        z = self._h_a(ws.y)
        eb.quantize(z, True, ws.z_hat, sym_out=ws.staging[:nz])
        self._hyper_into(ws)
        table = gc.device_table(ws.device)

        for group in ws.groups:       # slices of one dependency level are independent: their stacks run as batch items
            self._group_params(ws, group)
            for i in group:
                ops.gc_quantize(ws.y.ch(i * d, d), ws.mus[i], ws.scs[i], table, True, True, ws.y_hat.ch(i * d, d),
                                sym_out=ws.staging[nz + i * ns: nz + (i + 1) * ns],
                                idx_out=ws.staging[nz + ny + i * ns: nz + ny + (i + 1) * ns])
            self._lrp_group(ws, group)

    def enc_gpu(self, segments, ws=None):
        """GPU half of compress(): analysis transform into the scan workspace + the whole encoder scan (no host interaction)."""
        x0 = segments[0]
        ws = ws or self._plan(x0.B, x0.H // 16, x0.W // 16, x0.device)
        self._g_a(segments, ws.y)
        self._step(ws, "enc", lambda: self._enc_scan(ws))
        return ws

    def enc_finish(self, ws, defer=False):
        """Host half: the symbol / index planes leave the device through one async copy into pinned memory; the host range coder
        runs on a worker thread behind the copy's event, so the caller can keep queuing GPU work (synthesis transform, MC, ...).
        -> {"strings", "shape"} dict (a lazy PendingStrings when defer=True)."""
        if self.coder == "rans_ms":
            return self._enc_finish_ms(ws, defer)
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        nz, ny = ws.nz, ws.ny
        host = torch.empty(ws.staging.numel(), dtype=torch.int32, pin_memory=True)
        host.copy_(ws.staging, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        zB, zC, zH, zW = ws.z_hat.B, ws.z_hat.C, ws.z_hat.H, ws.z_hat.W

        def encode():
            done.synchronize()
            packed = host.numpy()
            z_strings = eb.encode_symbols(packed[:nz], zB, zC, zH, zW)
            enc = ops.RansEncoder()
            enc.encode_with_indexes(packed[nz:nz + ny], packed[nz + ny:], gc.host_tables())
            return {"strings": [[enc.flush()], z_strings], "shape": (zH, zW)}

        future = _CODER_POOL.submit(encode)
        self.__dict__.setdefault("_pending_futures", []).append(future)
        del self.__dict__["_pending_futures"][:-4]      # keep only the most recent few (older ones have long finished)
        pending = PendingStrings(future)
        return pending if defer else pending.result()

    # ---- multi-stream coder (coder = "rans_ms") ------------------------------------------------------------------------
    def _ms_layout(self, ws):
        """Per dependency level: (first symbol of the level in the y plane, n symbols, n streams, slot words, stream base, slot-word base)."""
        lay = ws.__dict__.get("_ms_layout")
        if lay is None or lay[0] != self.ms_symbols_per_stream:
            S, rows, sb, wb = int(self.ms_symbols_per_stream), [], 0, 0
            for g in ws.groups:
                n = len(g) * ws.ns
                nst = -(-n // S)
                sw = ops.rans_ms_slot_words(-(-n // nst))
                rows.append((g[0] * ws.ns, n, nst, sw, sb, wb))
                sb, wb = sb + nst, wb + nst * sw
            i16 = lambda k, **kw: torch.empty(k, dtype=torch.int16, **kw)
            bufs = dict(slots=i16(wb, device=ws.device), states=torch.empty(sb, dtype=torch.int32, device=ws.device), lens=i16(sb, device=ws.device))
            lay = ws.__dict__["_ms_layout"] = (S, rows, bufs, sb, wb)
        return lay

    def _enc_finish_ms(self, ws, defer=False):
        """Host half in multi-stream mode: the y symbols are range-coded ON THE DEVICE, level by level (one thread per stream); what leaves the
        device is the z symbol plane (coded on the host in the reference's format) and the streams' states / lengths / word slots, which a
        worker thread concatenates into the string."""
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        nz, ny = ws.nz, ws.ny
        _, rows, bufs, n_st, n_words = self._ms_layout(ws)
        tab = gc.device_tables(ws.device)
        for first, n, nst, sw, sb, wb in rows:
            ops.rans_ms_encode(ws.staging[nz + first: nz + first + n], ws.staging[nz + ny + first: nz + ny + first + n], n, nst, tab,
                               bufs["slots"][wb: wb + nst * sw], sw, bufs["states"][sb: sb + nst], bufs["lens"][sb: sb + nst])
        h_z = torch.empty(nz, dtype=torch.int32, pin_memory=True)
        h_states = torch.empty(n_st, dtype=torch.int32, pin_memory=True)
        h_lens = torch.empty(n_st, dtype=torch.int16, pin_memory=True)
        h_slots = torch.empty(n_words, dtype=torch.int16, pin_memory=True)
        h_z.copy_(ws.staging[:nz], non_blocking=True)
        h_states.copy_(bufs["states"], non_blocking=True)
        h_lens.copy_(bufs["lens"], non_blocking=True)
        h_slots.copy_(bufs["slots"], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        zB, zC, zH, zW = ws.z_hat.B, ws.z_hat.C, ws.z_hat.H, ws.z_hat.W

        def encode():
            done.synchronize()
            z_strings = eb.encode_symbols(h_z.numpy(), zB, zC, zH, zW)
            st, ln, sl = h_states.numpy().view(np.uint32), h_lens.numpy().view(np.uint16), h_slots.numpy().view(np.uint16)
            levels = []
            for first, n, nst, sw, sb, wb in rows:
                l = ln[sb: sb + nst]
                words = sl[wb: wb + nst * sw].reshape(nst, sw)
                keep = np.arange(sw, dtype=np.int32)[None, :] >= (sw - l.astype(np.int32))[:, None]       # a stream = the LAST `len` words of its slot
                levels.append((n, st[sb: sb + nst], l, words[keep]))
            return {"strings": [[ms_pack(levels)], z_strings], "shape": (zH, zW)}

        future = _CODER_POOL.submit(encode)
        self.__dict__.setdefault("_pending_futures", []).append(future)
        del self.__dict__["_pending_futures"][:-4]
        pending = PendingStrings(future)
        return pending if defer else pending.result()

    def _dec_chain_ms(self, strings, shape, device):
        """Decode chain in multi-stream mode: the string's stream tables and payload go to the device in ONE copy; every dependency level is
        then [decode kernel (one thread per stream) -> dequantise + LRP + next level's nets], stream-ordered, with no host round trip."""
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        h4, w4 = int(shape[0]), int(shape[1])
        ws = self._plan(len(strings[1]), h4 * 4, w4 * 4, device)
        S, rows, _, _, _ = self._ms_layout(ws)
        levels = ms_unpack(strings[0][0])
        if len(levels) != len(rows) or any(n != r[1] or len(st) != r[2] for (n, st, _, _), r in zip(levels, rows)):
            raise ValueError("multi-stream string does not match this codec's level layout (symbols per stream: %d)" % S)
        # one upload: per level [states u32 | word offsets u32 (n_streams + 1) | payload u16, padded to 4 bytes]
        parts, meta, pos = [], [], 0
        for n, st, ln, pay in levels:
            off = np.zeros(len(st) + 1, dtype=np.uint32)
            np.cumsum(ln.astype(np.uint32), out=off[1:])
            b = st.astype("<u4").tobytes() + off.tobytes() + pay.tobytes()
            b += b"\0" * (-len(b) % 16)
            meta.append((pos, pos + 4 * len(st), pos + 4 * len(st) + 4 * len(off), len(pay)))
            parts.append(b)
            pos += len(b)
        blob = b"".join(parts)
        up = ws.__dict__.get("_ms_up")
        if up is None or up[0].numel() < len(blob):
            cap = max(len(blob) * 2, 1 << 16)
            up = ws.__dict__["_ms_up"] = [torch.empty(cap, dtype=torch.uint8, pin_memory=True), torch.empty(cap, dtype=torch.uint8, device=device), None]
        if up[2] is not None:
            up[2].synchronize()                   # the previous frame's upload out of this pinned buffer has completed
        up[0].numpy()[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        up[1][:len(blob)].copy_(up[0][:len(blob)], non_blocking=True)
        up[2] = torch.cuda.Event()
        up[2].record()
        ws.z_sym_host.numpy()[:] = eb.decode_symbols(strings[1], h4, w4).reshape(-1)
        ws.z_sym.copy_(ws.z_sym_host, non_blocking=True)
        steps = self._segments(ws)
        self._step(ws, "dec0", steps["dec0"])
        tab, base = gc.device_tables(device), up[1].data_ptr()
        for k, (group, (first, n, nst, sw, sb, wb)) in enumerate(zip(ws.groups, rows)):
            o_st, o_off, o_pay, n_words = meta[k]
            ops.rans_ms_decode(ws.idx_dev[:n], n, nst, tab, base + o_st, base + o_off, base + o_pay, n_words, ws.sym_dev[:n])
            self._step(ws, f"dec{k + 1}", steps[f"dec{k + 1}"])
        return ws

    def run_compress(self, segments, want_x_hat=False, x_hat_dtype=None, defer=False):
        """-> ({"strings", "shape"} dict (lazy PendingStrings when defer=True), x_hat Act or None)."""
        ws = self.enc_gpu(segments)
        pending = self.enc_finish(ws, defer)
        x_hat = self._g_s(ws.y_hat, None, x_hat_dtype) if want_x_hat else None
        return pending, x_hat

    def _dec_nets(self, ws, group):
        """parameter nets + index planes of one dependency level (slices of `group` share their support)."""
        table, n = self.gaussian_conditional.device_table(ws.device), ws.ns

        self._group_params(ws, group)
        for j, i in enumerate(group):
            ops.gc_quantize(None, ws.mus[i], ws.scs[i], table, False, True, idx_out=ws.idx_dev[j * n:(j + 1) * n])

    def _dec_finish(self, ws, group):
        """symbols of one level -> y_hat slices (+ latent residual prediction)."""
        d, n = self.slice_depth, ws.ns

        for j, i in enumerate(group):
            ops.gc_dequantize(ws.sym_dev[j * n:(j + 1) * n], ws.mus[i], ws.y_hat.ch(i * d, d))
        self._lrp_group(ws, group)

    def run_decompress(self, strings, shape, device, x_hat_out=None, x_hat_dtype=None):
        ws = self.dec_chain(strings, shape, device)
        return self._g_s(ws.y_hat, x_hat_out, x_hat_dtype)

    def dec_chain(self, strings, shape, device):
        """strings -> y_hat in the scan workspace: host range decoder and scan kernels, level by level (one sync point per level)."""
        if self.coder == "rans_ms":
            return self._dec_chain_ms(strings, shape, device)
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        h4, w4 = int(shape[0]), int(shape[1])
        B = len(strings[1])
        ws = self._plan(B, h4 * 4, w4 * 4, device)
        tables = gc.host_tables()
        stream = torch.cuda.current_stream(device)
        ws.z_sym_host.numpy()[:] = eb.decode_symbols(strings[1], h4, w4).reshape(-1)
        ws.z_sym.copy_(ws.z_sym_host, non_blocking=True)
        dec = ops.RansDecoder(strings[0][0])
        groups, n = ws.groups, ws.ns
        steps = self._segments(ws)
        self._step(ws, "dec0", steps["dec0"])
        idx_taps = self.__dict__.get("debug_idx_taps")      # tests only: the index planes the decoder looked its CDFs up with, level by level
        for k, group in enumerate(groups):
            m = len(group) * n
            ws.idx_host[:m].copy_(ws.idx_dev[:m], non_blocking=True)
            stream.synchronize()                 # the one sync point of this level
            if idx_taps is not None:
                idx_taps.append(ws.idx_host[:m].clone())
            dec.decode_stream(ws.idx_host.numpy()[:m], tables, out=ws.sym_host.numpy()[:m])
            ws.sym_dev[:m].copy_(ws.sym_host[:m], non_blocking=True)
            self._step(ws, f"dec{k + 1}", steps[f"dec{k + 1}"])
        return ws

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
        """reference image_model.py:304-317: size the (initially empty) table buffers from the checkpoint first.
        Older checkpoints' key spellings (utils.py:105-125 `rename_key`) are accepted as well."""
        state_dict = rename_legacy_keys(state_dict)
        for mod, prefix, names in ((self.entropy_bottleneck, "entropy_bottleneck", ("_quantized_cdf", "_offset", "_cdf_length")),
                                   (self.gaussian_conditional, "gaussian_conditional",
                                    ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"))):
            for n in names:
                key = f"{prefix}.{n}"
                if key in state_dict:
                    cur = getattr(mod, n)
                    setattr(mod, n, torch.empty(state_dict[key].shape, dtype=cur.dtype, device=cur.device))
            mod.tables_changed()
            if hasattr(mod, "_dev_table"):
                mod._dev_table = None
        self.__dict__.pop("_fp_params", None)      # parameters may be replaced objects afterwards
        self._epoch_ref()[0] += 1                  # recorded scan programs / graphs hold packed copies of the old weights
        return super().load_state_dict(state_dict, strict=strict)

    def update(self, scale_table=None, force=False):
        """reference image_model.py:319-324."""
        updated = self.entropy_bottleneck.update(force=force)
        if scale_table is None:
            scale_table = get_scale_table()
        updated |= self.gaussian_conditional.update_scale_table(scale_table, force=force)
        if updated:                          # the device-side tables are rebuilt lazily: captured scan graphs hold the old ones
            self._epoch_ref()[0] += 1
        return updated

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))


class ICIP2020ResB(ChannelSplitICIP2020ResB):
    """MI355X-native I-frame codec (reference image_model.py:331-619): image [B,3,H,W] -> latent y
    (M = 320 channels @1/16) -> hyper-latent z (N = 192 @1/64); 10 slices of 32 channels, each conditioned on the
    hyperprior and on the first 5 decoded slices, so the scan has depth 6 (slices 5..9 are one dependency level).

    Same attribute paths / state_dict keys, `forward / compress / decompress / update / load_state_dict / aux_loss`
    contracts and bitstream as the reference.  Everything below the layer definitions -- slice scan on static workspaces,
    HIP graphs of the scan segments, batched decode levels, one device->host transfer per encode -- is the shared
    machinery of `ChannelSplitICIP2020ResB`; only the transforms differ (LeakyReLU(0.01) stacks of three convs,
    5x5/s2 convs in h_a, transposed 5x5/s2 convs in h_mean_s / h_scale_s)."""
    _stack_act = (N.ACT_LRELU, 0.01)

    def __init__(self, N=192, M=320):
        nn.Module.__init__(self)
        self.N, self.M = N, M
        self.num_slices = 10
        self.max_support_slices = 5
        d = M // self.num_slices
        if d * self.num_slices != M:
            raise ValueError(f"Slice do not evenly divide latent depth ({M}/{self.num_slices})")
        self.slice_depth = d
        lrelu = lambda: nn.LeakyReLU(inplace=True)

        def stage_list(first, make, last_out):
            mods = [first]
            for stage in range(3):
                mods += [ResBottleneckBlock(N) for _ in range(3)]
                mods.append(make(N, last_out if stage == 2 else N))
            return nn.Sequential(*mods)

        self.g_a = stage_list(conv(3, N), conv, M)                    # image_model.py:344-358
        self.g_s = stage_list(deconv(M, N), deconv, 3)                # :360-374
        self.h_a = nn.Sequential(conv(M, N, stride=1, kernel_size=3), lrelu(), conv(N, N), lrelu(), conv(N, N))   # :376-382
        hyper_synthesis = lambda: nn.Sequential(deconv(N, N), lrelu(), deconv(N, 256), lrelu(),
                                                conv(256, M, stride=1, kernel_size=3))                              # :384-398
        self.h_mean_s, self.h_scale_s = hyper_synthesis(), hyper_synthesis()
        k = self.max_support_slices
        stack = lambda cin: _gelu_stack([cin, 224, 128, d], act=lrelu)                                               # :408-436
        self.cc_mean_transforms = nn.ModuleList(stack(M + d * min(i, k)) for i in range(self.num_slices))
        self.cc_scale_transforms = nn.ModuleList(stack(M + d * min(i, k)) for i in range(self.num_slices))
        self.lrp_transforms = nn.ModuleList(stack(M + d * min(i + 1, k + 1)) for i in range(self.num_slices))
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)

    def _h_a(self, y):
        x = self.h_a[0].run(y, act=N.ACT_LRELU, slope=0.01, out_dtype=F32)
        x = self.h_a[2].run(x, act=N.ACT_LRELU, slope=0.01, out_dtype=F32)
        return self.h_a[4].run(x, out_dtype=F32)

    def _h_s(self, net, z_hat, out=None):
        x = net[0].run(z_hat, act=N.ACT_LRELU, slope=0.01, out_dtype=F32)
        x = net[2].run(x, act=N.ACT_LRELU, slope=0.01, out_dtype=F32)
        return net[4].run(x, out, out_dtype=F32)

    def decompress(self, strings, shape):
        """image_model.py:554-595: the reconstruction is clamped to [0, 1]."""
        dev = self.entropy_bottleneck.quantiles.device
        x_hat = self.run_decompress(strings, shape, dev, x_hat_dtype=F32)
        return {"x_hat": ops.eltwise(N.EW_CLAMP01, x_hat).nchw()}
