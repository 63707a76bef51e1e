"""Host-side op layer: NHWC activation slices (`Act`) + thin wrappers over the C ABI (libdsvc.so).

PyTorch is used here for device memory (torch.empty), streams and dtype tags only; every compute
step is a hand-written HIP kernel reached through `deepsvc_amd._native`.  CPU tensors are refused:
the product has no CPU path (the CPU restatement lives in `oracle/` and is test infrastructure).
"""
import ctypes
from ctypes import byref

import numpy as np
import torch

from . import _native as N

_TORCH2CODE = {torch.float32: N.F32, torch.float16: N.F16}


class Precision:
    """compute: MFMA input type of the conv kernels; storage: default activation storage type."""

    def __init__(self):
        self.compute = N.F32
        self.storage = torch.float32

    def set(self, mode):
        if mode == "fp32":
            self.compute, self.storage = N.F32, torch.float32
        elif mode == "f16":
            self.compute, self.storage = N.F16, torch.float16
        else:
            raise ValueError(f"unknown precision mode {mode!r} (use 'fp32' or 'f16')")
        self.mode = mode

    mode = "fp32"


RT = Precision()
PACK_EPOCH = [0]          # bumped whenever a conv's weights are (re)packed: captured HIP graphs hold those pointers
USE_GRAPHS = [True]       # entropy-scan segments are captured into HIP graphs and replayed (DSVC_NO_GRAPHS=1 disables)
import os as _os
if _os.environ.get("DSVC_NO_GRAPHS"):
    USE_GRAPHS[0] = False


def set_precision(mode):
    RT.set(mode)


def get_precision():
    return RT.mode


_SLOW_STREAM = __import__("os").environ.get("DSVC_SLOW_STREAM", "") != ""      # A/B switch (tools/host_overhead.py)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The calling thread's current HIP stream as a void*.  `torch.cuda.current_stream()` costs ~10 us of Python per call (device
    index plumbing), i.e. ~4 ms per frame over ~380 launches -- a third of a lane thread's host time (tools/host_overhead.py);
    the raw accessor torch itself uses is a plain C call."""
    if _raw_stream is not None and _cur_device is not None and not _SLOW_STREAM:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


import threading as _threading

_fork_pools = _threading.local()
ITEM_SCAN = [_os.environ.get("DSVC_NO_ITEM_SCAN", "0") != "1"]   # mean+scale stacks of a scan level as one launch per layer
FUSE_1X1 = [_os.environ.get("DSVC_NO_FUSE_1X1", "") == ""]   # ResBottleneck conv2 -> act -> conv3 -> + x in one launch (f16, 64 ch)
S2D = [_os.environ.get("DSVC_NO_S2D", "") == ""]       # conv5x5/s2 through the space-to-depth view (f16 mode)
USE_FORKS = [_os.environ.get("DSVC_FORKS", "") != ""]   # opt-in: measured 17.15 vs 17.6 fps (event edges cost more than the overlap)


# --------------------------------------------------------------------------------------------
# scan steps: a launch sequence recorded once, replayed by ONE C call (include/dsvc.h: dsvc_entropy_scan_step)
# --------------------------------------------------------------------------------------------
SCAN_STEP = [_os.environ.get("DSVC_NO_SCAN_STEP", "") == ""]
_recording = _threading.local()


class ScanRecorder:
    """While active on this thread, every conv / entropy / element-wise launch issued through this module is ALSO appended to
    `ops` as a dsvc_scan_op, and every tensor it touches (including the temporaries new_act hands out) is kept alive in
    `keep`; an op that has no scan-op form poisons the recording (the caller then keeps issuing the launches itself)."""

    def __init__(self):
        self.ops, self.keep, self.poisoned = [], [], None
        self.temps = []           # buffers handed out by new_act / scratch while recording: a FrameProgram packs them into one arena

    def __enter__(self):
        self.prev = getattr(_recording, "cur", None)
        _recording.cur = self
        return self

    def __exit__(self, *exc):
        _recording.cur = self.prev
        return False

    def add(self, kind, field, struct, keep):
        op = N.ScanOp()
        op.kind = kind
        ctypes.memmove(ctypes.addressof(getattr(op.u, field)), ctypes.addressof(struct), ctypes.sizeof(struct))
        self.ops.append(op)
        self.keep.extend(k for k in keep if k is not None)


def _rec():
    return getattr(_recording, "cur", None)


def poison_recording(why):
    r = _rec()
    if r is not None and r.poisoned is None:
        r.poisoned = why


class ScanProgram:
    """A frozen recording: replay = one call into the C library."""

    def __init__(self, recorder):
        assert recorder.poisoned is None
        self.n = len(recorder.ops)
        self.arr = (N.ScanOp * max(self.n, 1))(*recorder.ops)
        self.keep = recorder.keep

    def run(self):
        N.check(N.lib().dsvc_entropy_scan_step(_stream(), ctypes.addressof(self.arr), self.n), "entropy_scan_step")


class Arena:
    """One workspace shared by the frame segments that run one after the other on the same stream: its size is the LARGEST of their
    needs, not the sum.  Growing it re-patches the temporaries' pointers of every program registered with it."""

    def __init__(self):
        self.buf, self.programs, self.streams = None, [], set()

    def attach(self, prog, nbytes, device):
        self.programs.append(prog)
        if self.buf is None or self.buf.numel() < nbytes:
            if self.buf is not None and self.buf.is_cuda:
                # the old workspace goes back to the caching allocator of the stream it was allocated on; replays may have used it from
                # other streams (a caller that warms up on one stream and runs on another): nothing may still be reading it
                torch.cuda.synchronize(self.buf.device)
            self.buf = torch.empty(max(nbytes, FrameProgram.ALIGN), dtype=torch.uint8, device=device)
            self.streams = {_stream().value} if self.buf.is_cuda else set()       # (CPU tensors only in the host-logic tests)
            for p in self.programs:
                p._patch_arena(self.buf.data_ptr())
        else:
            prog._patch_arena(self.buf.data_ptr())

    def used_on_current_stream(self):
        """Called by every replay: a workspace allocated on one stream and used on another must tell the caching allocator so
        (record_stream), or its memory could be handed out again while a replay on the other stream is still running."""
        if self.buf is None or not self.buf.is_cuda:
            return
        sid = _stream().value
        if sid not in self.streams:
            self.buf.record_stream(torch.cuda.current_stream(self.buf.device))
            self.streams.add(sid)


class FrameProgram:
    """A frozen recording of one host-synchronisation-free SEGMENT of a P-frame (include/dsvc.h: dsvc_pframe_segment_run).

    Pointers inside the recorded ops fall into three classes, found by address range:
      * externals -- the caller's frame tensors (ref / cur / feature / sm_fea ...) and the segment's FRESH outputs (tensors handed
        back to the caller, newly allocated on every run like the reference's return values): patched into the op array per run;
      * temporaries -- everything new_act() handed out while recording: packed by lifetime (first .. last op that touches them) into
        ONE arena owned by the program, patched once;
      * everything else (packed weights, bias, scan workspaces, persistent outputs that a later segment reads): left as recorded and
        kept alive by the program.
    run(externals) = patch ~10-20 pointers + one C call that issues the segment's launches back to back."""

    ALIGN = 512

    def __init__(self, rec, externals, fresh, persistent, arena=None):
        """externals: [Act | tensor] inputs in binding order; fresh: [Act] outputs reallocated per run; persistent: [Act] outputs kept;
        arena: an `Arena` shared with the other segments of the same stream (None: a private one)."""
        assert rec.poisoned is None
        self.n = len(rec.ops)
        self.arr = (N.ScanOp * max(self.n, 1))(*rec.ops)
        buf_of = lambda e: e.buf if isinstance(e, Act) else e
        rng = lambda t: (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
        ext = [buf_of(e) for e in externals] + [a.buf for a in fresh]
        self.n_inputs, self.fresh_meta = len(externals), [(tuple(a.buf.shape), a.buf.dtype, a.buf.device, a.c0, a.C) for a in fresh]
        keep_ids = {id(a.buf) for a in persistent} | {id(t) for t in ext}
        temps = [t for t in {id(t): t for t in rec.temps}.values() if id(t) not in keep_ids]
        table = sorted([(rng(t) + ("ext", i)) for i, t in enumerate(ext)] + [(rng(t) + ("tmp", i)) for i, t in enumerate(temps)])
        starts = [r[0] for r in table]
        import bisect

        def owner(v):
            k = bisect.bisect_right(starts, v) - 1
            if k >= 0 and table[k][0] <= v < table[k][1]:
                return table[k]
            return None

        base = ctypes.addressof(self.arr)
        step, uoff = ctypes.sizeof(N.ScanOp), N.ScanOp.u.offset
        offs = {m: N.pointer_offsets(dict(N._ScanU._fields_)[m]) for m in set(N.SCAN_UNION_MEMBER.values())}
        first, last = [None] * len(temps), [None] * len(temps)
        ext_rel, tmp_rel = [], []                    # (field address, slot, delta)
        for i in range(self.n):
            for o in offs[N.SCAN_UNION_MEMBER[self.arr[i].kind]]:
                addr = base + i * step + uoff + o
                v = ctypes.c_uint64.from_address(addr).value
                if not v:
                    continue
                own = owner(v)
                if own is None:
                    continue
                lo, _, kind, slot = own
                if kind == "ext":
                    ext_rel.append((ctypes.c_uint64.from_address(addr), slot, v - lo))
                else:
                    tmp_rel.append((addr, slot, v - lo))
                    first[slot] = i if first[slot] is None else first[slot]
                    last[slot] = i
        # ---- pack the temporaries into one arena by lifetime (greedy first fit over freed blocks) ----
        order = sorted((k for k in range(len(temps)) if first[k] is not None), key=lambda k: first[k])
        size = lambda k: -(-(temps[k].numel() * temps[k].element_size()) // self.ALIGN) * self.ALIGN
        free, live, offset, top = [], [], {}, 0       # free: [(off, size)], live: [(last, off, size)]
        for k in order:
            for item in [x for x in live if x[0] < first[k]]:
                live.remove(item)
                free.append((item[1], item[2]))
            free.sort()
            merged = []
            for off, sz in free:                       # coalesce neighbours
                if merged and merged[-1][0] + merged[-1][1] == off:
                    merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                else:
                    merged.append((off, sz))
            free = merged
            need, pick = size(k), None
            for j, (off, sz) in enumerate(free):
                if sz >= need and (pick is None or sz < free[pick][1]):
                    pick = j
            if pick is not None:
                off, sz = free.pop(pick)
                if sz > need:
                    free.append((off + need, sz - need))
                offset[k] = off
            else:
                if free and free[-1][0] + free[-1][1] == top:      # grow the last free block instead of leaving a hole
                    off, sz = free.pop()
                    offset[k], top = off, off + need
                else:
                    offset[k], top = top, top + need
            live.append((last[k], offset[k], need))
        dev = temps[0].device if temps else (ext[0].device if ext else None)
        self.arena_bytes, self.temp_bytes = top, sum(size(k) for k in order)
        self._tmp_rel = [(ctypes.c_uint64.from_address(addr), offset[slot] + delta) for addr, slot, delta in tmp_rel]
        self._arena = arena if arena is not None else Arena()
        if dev is not None:
            self._arena.attach(self, top, dev)
        self.ext_rel = ext_rel
        drop = {id(t) for t in temps} | {id(t) for t in ext}         # the arena replaces the former, the caller owns the latter
        self.keep = [k for k in rec.keep if not (isinstance(k, torch.Tensor) and id(k) in drop)]
        self.persistent = list(persistent)

    arena = property(lambda self: self._arena.buf)

    def _patch_arena(self, base):
        for cell, off in self._tmp_rel:
            cell.value = base + off

    def run(self, externals):
        """-> (fresh output Acts, persistent output Acts)"""
        assert len(externals) == self.n_inputs
        bases = [(e.buf if isinstance(e, Act) else e).data_ptr() for e in externals]
        fresh = []
        for shape, dtype, dev, c0, C in self.fresh_meta:
            a = Act(torch.empty(shape, dtype=dtype, device=dev), c0, C)
            fresh.append(a)
            bases.append(a.buf.data_ptr())
        for cell, slot, delta in self.ext_rel:
            cell.value = bases[slot] + delta
        self._arena.used_on_current_stream()
        N.check(N.lib().dsvc_pframe_segment_run(_stream(), ctypes.addressof(self.arr), self.n), "pframe_segment_run")
        return fresh, self.persistent


FRAME_PROGRAMS = [_os.environ.get("DSVC_NO_FRAME_PROGRAMS", "") == ""]


def run_segment(cache, key, externals, fn, arena=None):
    """Run one frame segment: fn(*externals) -> (fresh outputs [Act...], persistent outputs [Act...]).  First call: issued launch by
    launch while recorded; afterwards one FrameProgram.run.  A segment holding an op without a scan-op form stays eager.
    `arena`: the Arena shared by the segments that run serially on this stream."""
    prog = cache.get(key) if (FRAME_PROGRAMS[0] and cache is not None) else False
    if prog is False or _rec() is not None:
        return fn(*externals)
    if prog is None:
        with ScanRecorder() as rec:
            fresh, persistent = fn(*externals)
        cache[key] = False if rec.poisoned is not None else FrameProgram(rec, externals, fresh, persistent, arena)
        return fresh, persistent
    return prog.run(externals)


def set_option(key, value):
    """Kernel-selection switch of the conv dispatcher (include/dsvc.h: dsvc_set_option); returns the previous value."""
    return N.check(N.lib().dsvc_set_option(key.encode(), int(value)), "set_option")


def get_counter(key):
    """Launch counter of one conv kernel family (include/dsvc.h: dsvc_get_counter)."""
    return N.check(N.lib().dsvc_get_counter(key.encode()), "get_counter")


def run_parallel(fns):
    """Run independent launch sequences concurrently: branch 0 on the current stream, the others on side streams that
    fork from it and join back.  Eagerly this overlaps tiny latency-bound kernels; under HIP-graph capture the forks
    become parallel branches of the graph.  Each branch must allocate/free its temporaries itself (they live on its
    stream) and write results into buffers that outlive the join."""
    tl = _fork_pools.__dict__
    if len(fns) == 1 or tl.get("depth", 0) > 0 or not USE_FORKS[0]:
        # forks must stay FLAT: a fork inside a forked branch crashes hipStreamEndCapture on ROCm 7.0 (measured with
        # tools/graph_fork_probe.py), so a nested call degrades to sequential issue on the branch's stream
        return [fn() for fn in fns]
    cur = torch.cuda.current_stream()
    streams = tl.setdefault("streams", {}).setdefault((cur.device.index, cur.cuda_stream), [])
    while len(streams) < len(fns) - 1:
        streams.append(torch.cuda.Stream(cur.device))
    out = [None] * len(fns)
    tl["depth"] = 1
    try:
        for i, fn in enumerate(fns[1:]):
            s = streams[i]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                out[i + 1] = fn()
        out[0] = fns[0]()
    finally:
        tl["depth"] = 0
        for s in streams[: len(fns) - 1]:
            cur.wait_stream(s)
    return out


def _require_gpu(t):
    if not t.is_cuda:
        raise N.NativeError("deepsvc_amd ops run on the GPU only (got a CPU tensor); "
                            "the CPU restatement is oracle/ and is not a product path")


class Act:
    """An NHWC activation: channels [c0, c0+C) of a contiguous torch buffer [B, H, W, ld].  Geometry, dtype code and the device
    pointer are computed once here: a frame issues ~400 launches from Python, each touching several Acts (tools/host_overhead.py)."""
    __slots__ = ("buf", "c0", "C", "B", "H", "W", "ld", "dtype", "code", "ptr", "npix")

    def __init__(self, buf, c0=0, C=None):
        shp = buf.shape
        assert len(shp) == 4 and buf.is_contiguous()
        self.buf, self.c0 = buf, c0
        self.B, self.H, self.W, self.ld = shp
        self.C = self.ld - c0 if C is None else C
        self.dtype = buf.dtype
        self.code = _TORCH2CODE[self.dtype]
        self.ptr = buf.data_ptr() + c0 * buf.element_size()
        self.npix = self.B * self.H * self.W

    device = property(lambda s: s.buf.device)

    def ch(self, c0, C):
        assert 0 <= c0 and c0 + C <= self.C
        return Act(self.buf, self.c0 + c0, C)

    def operand(self):
        return N.Operand(self.ptr, self.ld, self.code)

    def nchw_view(self):
        """Zero-copy logical-NCHW view in the storage dtype (channels_last strides)."""
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)

    def nchw(self):
        """Logical NCHW fp32 view/copy for the reference-style API."""
        if self.dtype == torch.float32:
            return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)
        out = scratch((self.B, self.C, self.H, self.W), torch.float32, self.device)
        _rec_call(N.SCAN_NHWC_TO_NCHW, [self.ptr, out.data_ptr()], [self.ld, self.code, self.B, self.C, self.H, self.W], keep=[self.buf, out])
        N.check(N.lib().dsvc_nhwc_to_nchw(_stream(), self.ptr, self.ld, self.code, out.data_ptr(),
                                          self.B, self.C, self.H, self.W), "nhwc_to_nchw")
        return out


def new_act(B, H, W, C, dtype=None, device=None, ld=None):
    ld = C if ld is None else ld
    a = Act(torch.empty((B, H, W, ld), dtype=dtype or RT.storage, device=device), 0, C)
    r = getattr(_recording, "cur", None)
    if r is not None:            # a temporary of a recorded scan step: replays write to this address, so it must stay allocated
        r.keep.append(a.buf)
        r.temps.append(a.buf)
    return a


def scratch(shape, dtype, device):
    """A plain temporary tensor (workspace / small result) that a recording knows about."""
    t = torch.empty(shape, dtype=dtype, device=device)
    r = getattr(_recording, "cur", None)
    if r is not None:
        r.keep.append(t)
        r.temps.append(t)
    return t


def _rec_call(kind, ptrs, ints, floats=(), keep=()):
    r = _rec()
    if r is not None:
        c = N.ScanCall()
        for k, v in enumerate(ptrs):
            c.p[k] = v
        for k, v in enumerate(ints):
            c.i[k] = int(v)
        for k, v in enumerate(floats):
            c.f[k] = float(v)
        r.add(kind, "call", c, list(keep))


def to_act(x, dtype=None):
    """Reference-style tensor [B,C,H,W] -> Act.  Zero-copy when x is fp32 channels_last memory."""
    if isinstance(x, Act):
        return x
    _require_gpu(x)
    assert x.dim() == 4, "expected an NCHW tensor"
    want = dtype or torch.float32
    xp = x.permute(0, 2, 3, 1)
    if x.dtype == want and xp.is_contiguous():
        return Act(xp)
    B, C, H, W = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        poison_recording("nchw_to_nhwc of a tensor that needs a torch copy first")
    src = x.to(torch.float32).contiguous() if (x.dtype != torch.float32 or not x.is_contiguous()) else x
    out = new_act(B, H, W, C, want, x.device)
    _rec_call(N.SCAN_NCHW_TO_NHWC, [src.data_ptr(), out.ptr], [out.ld, out.code, B, C, H, W], keep=[src, out.buf])
    N.check(N.lib().dsvc_nchw_to_nhwc(_stream(), src.data_ptr(), out.ptr, out.ld, out.code, B, C, H, W), "nchw_to_nhwc")
    return out


# --------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------
class PackedConv:
    """Device-resident packed weights (+ fp32 bias) of one conv for one (segments, compute type)."""

    def __init__(self, w_oihw, bias, seg_channels, stride, pad, compute, device, phys_channels=None, s2d=False):
        """seg_channels: how the weight's input channels split over the input segments (torch.cat order);
        phys_channels: channels each segment physically provides (>= logical; the extra ones are zero padding of a
        packed buffer) -- the weight gets zero columns there."""
        w = w_oihw.detach().to("cpu", torch.float32).contiguous()
        self.Cout, cin, self.kh, self.kw = w.shape
        assert sum(seg_channels) == cin, f"segments {seg_channels} do not add up to Cin={cin}"
        if phys_channels is not None and tuple(phys_channels) != tuple(seg_channels):
            parts, c = [], 0
            for lc, pc in zip(seg_channels, phys_channels):
                assert pc >= lc
                parts.append(w[:, c:c + lc])
                if pc > lc:
                    parts.append(torch.zeros((self.Cout, pc - lc, self.kh, self.kw), dtype=torch.float32))
                c += lc
            w = torch.cat(parts, 1).contiguous()
            seg_channels = phys_channels
        self.s2d = bool(s2d)
        if self.s2d:                         # conv5x5/s2 -> 3x3/s1 over the space-to-depth view (include/dsvc.h: s2d)
            assert (self.kh, self.kw, stride, pad) == (5, 5, 2, 2)
            w, seg_channels = s2d_weights(w, seg_channels, 8 if compute == N.F16 else 4)
            self.kh = self.kw = 3
            stride, pad = 1, 1
        self.segC = tuple(int(c) for c in seg_channels)
        self.stride, self.pad, self.compute = stride, pad, compute
        segs = (ctypes.c_int32 * len(self.segC))(*self.segC)
        L = N.lib()
        nbytes = L.dsvc_conv_packed_bytes(self.Cout, len(self.segC), segs, self.kh, self.kw, compute)
        N.check(nbytes, "conv_packed_bytes")
        host = np.empty(nbytes, dtype=np.uint8)
        N.check(L.dsvc_conv_pack_weights(w.data_ptr(), self.Cout, len(self.segC), segs, self.kh, self.kw, compute,
                                         host.ctypes.data), "conv_pack_weights")
        self.weight = torch.from_numpy(host).to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        PACK_EPOCH[0] += 1


class ItemPackedConv:
    """Packed weights of G convs of identical shape, item after item, for one launch that runs them side by side as the
    batch items of include/dsvc.h's `batch_inner` mode (the slice scan's mean / scale stacks)."""

    def __init__(self, packs):
        p0 = packs[0]
        sig = lambda p: (p.Cout, p.kh, p.kw, p.segC, p.stride, p.pad, p.compute, p.s2d, p.weight.numel(), p.bias is None)
        assert all(sig(p) == sig(p0) for p in packs) and not p0.s2d and p0.weight.numel() % 16 == 0
        self.items, self.sources = len(packs), tuple(packs)
        self.Cout, self.kh, self.kw, self.segC = p0.Cout, p0.kh, p0.kw, p0.segC
        self.stride, self.pad, self.compute, self.s2d = p0.stride, p0.pad, p0.compute, False
        self.weight = torch.cat([p.weight for p in packs])
        self.weight_stride = p0.weight.numel()
        self.bias = None if p0.bias is None else torch.stack([p.bias for p in packs]).contiguous()


def s2d_weights(w5, seg_channels, V):
    """[Cout, Cin, 5, 5] weights of a stride-2 pad-2 conv -> [Cout, 4*Cin, 3, 3] weights of the equivalent stride-1 pad-1
    conv over the space-to-depth view of include/dsvc.h (view channel g*4V + (2*py+px)*V + c = channel g*V + c of image
    pixel (2Y+py, 2X+px), per input segment):  out(y,x) = sum W5[ky][kx] in[2y+ky-2][2x+kx-2], and 2y + ky - 2 =
    2(y+dy) + py with ky = 2*dy + py + 2, so W3[.., dy+1, dx+1] at phase (py,px) = W5[.., 2dy+py+2, 2dx+px+2] (absent taps 0)."""
    cout = w5.shape[0]
    parts, c0 = [], 0
    for C in seg_channels:
        assert C % V == 0, f"s2d needs segment channels % {V} == 0, got {C}"
        ws = w5[:, c0:c0 + C]                                        # [Cout, C, 5, 5]
        w3 = torch.zeros((cout, C // V, 4, V, 3, 3), dtype=torch.float32)
        for py in range(2):
            for px in range(2):
                for dy in range(-1, 2):
                    for dx in range(-1, 2):
                        ky, kx = 2 * dy + py + 2, 2 * dx + px + 2
                        if 0 <= ky < 5 and 0 <= kx < 5:
                            w3[:, :, 2 * py + px, :, dy + 1, dx + 1] = ws[:, :, ky, kx].reshape(cout, C // V, V)
        parts.append(w3.reshape(cout, 4 * C, 3, 3))
        c0 += C
    return torch.cat(parts, 1).contiguous(), [4 * C for C in seg_channels]


def deconv_as_conv_weights(w_iohw, bias):
    """ConvTranspose2d(5,s2,p2,op1) parameters -> (3x3 conv weight [4*Cout,Cin,3,3], bias [4*Cout]) on the host."""
    w = w_iohw.detach().to("cpu", torch.float32).contiguous()
    cin, cout = w.shape[0], w.shape[1]
    assert w.shape[2:] == (5, 5)
    wo = torch.empty((4 * cout, cin, 3, 3), dtype=torch.float32)
    bo = torch.empty(4 * cout, dtype=torch.float32)
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    N.check(N.lib().dsvc_deconv5x5s2_to_conv3x3(w.data_ptr(), None if b is None else b.data_ptr(), cin, cout,
                                                wo.data_ptr(), bo.data_ptr()), "deconv_to_conv")
    return wo, bo


def profile_begin(kh, stride, C, Cout, H, W):
    """Arm per-launch HIP-event timing of one conv shape inside the C library (include/dsvc.h: dsvc_profile_begin)."""
    N.check(N.lib().dsvc_profile_begin(kh, stride, C, Cout, H, W), "profile_begin")


def profile_end():
    """-> ([ms per matching launch], [residual operands per launch])"""
    L = N.lib()
    n = N.check(L.dsvc_profile_end(None, None, 0), "profile_end(count)")
    ms, nres = (ctypes.c_float * max(n, 1))(), (ctypes.c_int32 * max(n, 1))()
    N.check(L.dsvc_profile_end(ms, nres, n), "profile_end")
    return list(ms)[:n], list(nres)[:n]


def conv2d(pc, inputs, out=None, *, act=N.ACT_NONE, pre_act=N.ACT_NONE, slope=0.01, res1=None, gate=None, res2=None,
           shuffle=1, out_dtype=None, post=None, items=None, item_out=None):
    """Run one fused conv.  inputs: list of Act (virtual torch.cat along channels).
    items=(inner, [(stride_lo, stride_hi) per input, bytes]) with an ItemPackedConv `pc`: pc.items independent convs in one
    launch (include/dsvc.h: batch_inner); every input Act is item 0's image, the output is a dense [items, H, W, C] Act."""
    x0 = inputs[0]
    if items is not None:
        return _conv2d_items(pc, inputs, out, items, act=act, pre_act=pre_act, slope=slope, res1=res1, shuffle=shuffle, out_dtype=out_dtype,
                             item_out=item_out)
    _require_gpu(x0.buf)
    s2d = getattr(pc, "s2d", False)
    mult = 4 if s2d else 1
    assert tuple(a.C * mult for a in inputs) == pc.segC, f"input channels {[a.C for a in inputs]} != packed segments {pc.segC}"
    B, H, W = x0.B, x0.H, x0.W
    if s2d:                                 # the conv sees the (H/2 x W/2) space-to-depth view of the inputs
        assert H % 2 == 0 and W % 2 == 0
        for a in inputs:
            assert (a.B, a.H, a.W) == (B, H, W)
        H, W = H // 2, W // 2
    Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.kw) // pc.stride + 1
    if out is None:
        out = new_act(B, Ho * shuffle, Wo * shuffle, pc.Cout // (shuffle * shuffle), out_dtype, x0.device)
    else:
        assert (out.B, out.H, out.W, out.C) == (B, Ho * shuffle, Wo * shuffle, pc.Cout // (shuffle * shuffle)), \
            f"out shape {(out.B, out.H, out.W, out.C)} mismatch"
    tmpl = pc.__dict__.get("_desc_tmpl")
    if tmpl is None:                        # the fields that depend on the packed weights only: filled once, copied per launch
        t = N.ConvDesc()
        t.kh, t.kw, t.stride, t.pad = pc.kh, pc.kw, pc.stride, pc.pad
        t.Cout, t.compute = pc.Cout, pc.compute
        t.weight = pc.weight.data_ptr()
        t.bias = None if pc.bias is None else pc.bias.data_ptr()
        t.s2d = 1 if s2d else 0
        tmpl = pc.__dict__["_desc_tmpl"] = bytes(t)
    d = N.ConvDesc.from_buffer_copy(tmpl)
    d.B, d.H, d.W, d.nseg = B, H, W, len(inputs)
    for i, a in enumerate(inputs):
        assert s2d or (a.B, a.H, a.W) == (B, H, W)
        sg = d.seg[i]
        sg.ptr, sg.C, sg.ld, sg.dtype = a.ptr, a.C * mult, a.ld, a.code
    if post is not None:                    # fused trailing 1x1 conv (include/dsvc.h: post_weight): out = post(act(conv)) + res1
        assert (post.kh, post.kw, post.Cout, post.segC) == (1, 1, pc.Cout, (pc.Cout,)) and post.compute == pc.compute
        d.post_weight = post.weight.data_ptr()
        d.post_bias = None if post.bias is None else post.bias.data_ptr()
    d.pre_act, d.act, d.slope = pre_act, act, slope
    if res1 is not None:
        assert (res1.B, res1.H, res1.W, res1.C) == (out.B, out.H, out.W, out.C)
        d.res1, d.res1_ld, d.res1_dtype = res1.ptr, res1.ld, res1.code
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.numel() == B * pc.Cout
        d.res1_scale = gate.data_ptr()
    if res2 is not None:
        assert (res2.B, res2.H, res2.W, res2.C) == (out.B, out.H, out.W, out.C)
        d.res2, d.res2_ld, d.res2_dtype = res2.ptr, res2.ld, res2.code
    d.shuffle = shuffle
    d.out, d.out_ld, d.out_dtype = out.ptr, out.ld, out.code
    r = _rec()
    if r is not None:
        r.add(N.SCAN_CONV, "conv", d, [pc, out.buf, gate, post] + [a.buf for a in inputs] + [x.buf for x in (res1, res2) if x is not None])
    N.check(N.lib().dsvc_conv2d(_stream(), byref(d)), "conv2d")
    return out


def _conv2d_items(pc, inputs, out, items, *, act, pre_act, slope, res1, shuffle, out_dtype, item_out=None):
    """item_out = (out_bstride, res1_bstride) in bytes: `out` (and `res1`) are item 0's single-image Acts and item b's sit b strides further
    (include/dsvc.h: out_bstride) -- e.g. consecutive channel slices of one buffer."""
    inner, strides = items
    x0 = inputs[0]
    _require_gpu(x0.buf)
    assert isinstance(pc, ItemPackedConv) and tuple(a.C for a in inputs) == pc.segC and len(strides) == len(inputs) and inner >= 1
    G, H, W = pc.items, x0.H, x0.W
    Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.kw) // pc.stride + 1
    oshape = (G, Ho * shuffle, Wo * shuffle, pc.Cout // (shuffle * shuffle))
    if item_out is not None:
        oshape = (1,) + oshape[1:]
        assert out is not None
    if out is None:
        out = new_act(*oshape, out_dtype, x0.device)
    assert (out.B, out.H, out.W, out.C) == oshape, f"out shape {(out.B, out.H, out.W, out.C)} != {oshape}"
    d = N.ConvDesc()
    d.B, d.H, d.W, d.nseg = G, H, W, len(inputs)
    for i, (a, (lo, hi)) in enumerate(zip(inputs, strides)):
        assert (a.H, a.W) == (H, W)
        d.seg[i].ptr, d.seg[i].C, d.seg[i].ld, d.seg[i].dtype = a.ptr, a.C, a.ld, a.code
        d.in_bstride[i][0], d.in_bstride[i][1] = lo, hi
    d.batch_inner, d.weight_bstride, d.bias_bstride = inner, pc.weight_stride, pc.Cout
    d.kh, d.kw, d.stride, d.pad = pc.kh, pc.kw, pc.stride, pc.pad
    d.Cout, d.compute = pc.Cout, pc.compute
    d.weight = pc.weight.data_ptr()
    d.bias = None if pc.bias is None else pc.bias.data_ptr()
    d.pre_act, d.act, d.slope = pre_act, act, slope
    if res1 is not None:
        assert (res1.B, res1.H, res1.W, res1.C) == oshape
        d.res1, d.res1_ld, d.res1_dtype = res1.ptr, res1.ld, res1.code
    d.shuffle = shuffle
    d.out, d.out_ld, d.out_dtype = out.ptr, out.ld, out.code
    if item_out is not None:
        d.out_bstride, d.res1_bstride = item_out
    r = _rec()
    if r is not None:
        r.add(N.SCAN_CONV, "conv", d, [pc, out.buf] + [a.buf for a in inputs] + ([res1.buf] if res1 is not None else []))
    N.check(N.lib().dsvc_conv2d(_stream(), byref(d)), "conv2d (items)")
    return out


def items_ok(acts):
    """Can these inputs feed the vector staging path that include/dsvc.h's batch_inner mode needs?"""
    for a in acts:
        es = a.buf.element_size()
        same = a.code == RT.compute
        q = (8 if RT.compute == N.F16 else 4) if same else 4       # whole 16-byte quarters of a 64-byte chunk (csrc/conv_mfma.hip: fast)
        if a.ptr % 16 or (a.ld * es) % 16 or a.C % q or not (same or (RT.compute == N.F16 and a.dtype == torch.float32)):
            return False
    return True


# --------------------------------------------------------------------------------------------
# resampling / layout / pointwise
# --------------------------------------------------------------------------------------------
def warp(x, flow, out=None):
    assert flow.dtype == torch.float32 and flow.C == 2
    out = out or new_act(x.B, x.H, x.W, x.C, x.dtype, x.device)
    _rec_call(N.SCAN_WARP, [x.ptr, flow.ptr, out.ptr], [x.ld, x.code, flow.ld, out.ld, out.code, x.B, x.H, x.W, x.C], keep=[x.buf, flow.buf, out.buf])
    N.check(N.lib().dsvc_warp_bilinear_border(_stream(), x.ptr, x.ld, x.code, flow.ptr, flow.ld, out.ptr, out.ld, out.code,
                                              x.B, x.H, x.W, x.C), "warp")
    return out


def upsample2x(x, scale=1.0, out=None):
    assert x.dtype == torch.float32
    out = out or new_act(x.B, 2 * x.H, 2 * x.W, x.C, torch.float32, x.device)
    _rec_call(N.SCAN_UPSAMPLE2X, [x.ptr, out.ptr], [x.ld, out.ld, x.B, x.H, x.W, x.C], [scale], keep=[x.buf, out.buf])
    N.check(N.lib().dsvc_upsample2x_bilinear(_stream(), x.ptr, x.ld, out.ptr, out.ld, x.B, x.H, x.W, x.C, float(scale)), "upsample2x")
    return out


def pool2x2(x, mode, out=None):
    out = out or new_act(x.B, x.H // 2, x.W // 2, x.C, x.dtype, x.device)
    assert out.dtype == x.dtype
    _rec_call(N.SCAN_POOL2X2, [x.ptr, out.ptr], [0 if mode == "avg" else 1, x.ld, out.ld, x.code, x.B, x.H, x.W, x.C], keep=[x.buf, out.buf])
    N.check(N.lib().dsvc_pool2x2(_stream(), 0 if mode == "avg" else 1, x.ptr, x.ld, out.ptr, out.ld, x.code, x.B, x.H, x.W, x.C), "pool2x2")
    return out


def pixel_shuffle_nchw(x_nchw, r, out=None, dtype=None):
    _require_gpu(x_nchw)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous():
        poison_recording("pixel_shuffle_nchw of a tensor that needs a torch copy first")
    x = x_nchw.to(torch.float32).contiguous()
    B, C, H, W = x.shape
    co = C // (r * r)
    out = out or new_act(B, H * r, W * r, co, dtype, x.device)
    _rec_call(N.SCAN_PIXEL_SHUFFLE, [x.data_ptr(), out.ptr], [out.ld, out.code, B, co, H, W, r], keep=[x, out.buf])
    N.check(N.lib().dsvc_pixel_shuffle_nchw(_stream(), x.data_ptr(), out.ptr, out.ld, out.code, B, co, H, W, r), "pixel_shuffle")
    return out


def _op(a):
    return None if a is None else byref(a.operand())


def eltwise(op, a, b=None, c=None, d=None, e=None, out=None, out_dtype=None):
    out = out or new_act(a.B, a.H, a.W, a.C, out_dtype or a.dtype, a.device)
    r = _rec()
    if r is not None:
        w = N.ScanEw()
        w.op, w.C, w.n_pix = op, a.C, a.npix
        for name, x in (("a", a), ("b", b), ("c", c), ("d", d), ("e", e)):
            if x is not None:
                o = getattr(w, name)
                o.ptr, o.ld, o.dtype = x.ptr, x.ld, x.code
        w.out, w.out_ld, w.out_dtype = out.ptr, out.ld, out.code
        r.add(N.SCAN_ELTWISE, "ew", w, [out.buf] + [x.buf for x in (a, b, c, d, e) if x is not None])
    N.check(N.lib().dsvc_eltwise(_stream(), op, a.npix, a.C, _op(a), _op(b), _op(c), _op(d), _op(e), out.ptr, out.ld, out.code), "eltwise")
    return out


def se_gate(x, w1, w2):
    """SELayer gate [B, C] fp32 for activation x (w1 [C/r, C], w2 [C, C/r] device fp32)."""
    L = N.lib()
    ws = scratch((L.dsvc_se_workspace_floats(x.B, x.C),), torch.float32, x.device)
    gate = scratch((x.B, x.C), torch.float32, x.device)
    _rec_call(N.SCAN_SE_GATE, [x.ptr, w1.data_ptr(), w2.data_ptr(), ws.data_ptr(), gate.data_ptr()], [x.ld, x.code, x.B, x.H, x.W, x.C, w1.shape[0]],
              keep=[x.buf, w1, w2, ws, gate])
    N.check(L.dsvc_se_gate(_stream(), x.ptr, x.ld, x.code, x.B, x.H, x.W, x.C, w1.data_ptr(), w2.data_ptr(), w1.shape[0],
                           ws.data_ptr(), gate.data_ptr()), "se_gate")
    return gate


def refine_aggregate(x, rx, rs, mean=True, relu=False):
    """RefineNet1's aggregation (include/dsvc.h: dsvc_refine_aggregate): x, rx [1,H,W,C] fp32 Acts, rs [L,H,W,C] fp32 Act."""
    poison_recording("refine_aggregate")          # not a scan-step op
    for a in (x, rx, rs):
        _require_gpu(a.buf)
        assert a.dtype == torch.float32, "refine_aggregate works on fp32 features"
    out = new_act(1, x.H, x.W, x.C, torch.float32, x.device)
    N.check(N.lib().dsvc_refine_aggregate(_stream(), x.ptr, x.ld, rx.ptr, rx.ld, rs.ptr, rs.ld, rs.B, x.npix, x.C, int(bool(mean)), int(bool(relu)),
                                          out.ptr, out.ld), "refine_aggregate")
    return out


def reduce(mode, a, b=None):
    """mode: 'sum' | 'sumlog' | 'sqdiff' -> 0-dim fp64 device tensor."""
    poison_recording("reduce")          # not a scan-step op
    L = N.lib()
    ws = torch.empty(L.dsvc_reduce_workspace_floats(), dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float64, device=a.device)
    N.check(L.dsvc_reduce(_stream(), {"sum": 0, "sumlog": 1, "sqdiff": 2}[mode], a.npix, a.C, _op(a), _op(b),
                          ws.data_ptr(), out.data_ptr()), "reduce")
    return out


def ssim_level(x, y, data_range=1.0):
    """One MS-SSIM level -> (ssim[B*C], cs[B*C]) fp32 device vectors (include/dsvc.h: dsvc_ssim_level)."""
    assert (x.B, x.H, x.W, x.C) == (y.B, y.H, y.W, y.C)
    poison_recording("ssim_level")          # not a scan-step op
    L = N.lib()
    ws = torch.empty(max(L.dsvc_ssim_workspace_floats(x.B, x.H, x.W, x.C), 1), dtype=torch.float32, device=x.device)
    out = torch.empty((2, x.B * x.C), dtype=torch.float32, device=x.device)
    N.check(L.dsvc_ssim_level(_stream(), _op(x), _op(y), x.B, x.H, x.W, x.C, float(data_range), ws.data_ptr(),
                              out[0].data_ptr(), out[1].data_ptr()), "ssim_level")
    return out[0], out[1]


def avgpool2_pad(x):
    """F.avg_pool2d(x, 2, padding=(H % 2, W % 2)) -> dense fp32 Act."""
    poison_recording("avgpool2_pad")          # not a scan-step op
    Ho, Wo = (x.H + 2 * (x.H & 1) - 2) // 2 + 1, (x.W + 2 * (x.W & 1) - 2) // 2 + 1
    out = new_act(x.B, Ho, Wo, x.C, torch.float32, x.device)
    N.check(N.lib().dsvc_avgpool2_pad(_stream(), _op(x), out.ptr, x.B, x.H, x.W, x.C), "avgpool2_pad")
    return out


# --------------------------------------------------------------------------------------------
# entropy-model device arithmetic
# --------------------------------------------------------------------------------------------
def gc_quantize(y, mu, scale, table, want_symbols=True, want_indexes=True, y_hat=None, sym_out=None, idx_out=None):
    """sym_out / idx_out: optional preallocated int32 device vectors (views into a staging buffer)."""
    B, H, W, C = mu.B, mu.H, mu.W, mu.C
    dev, n = mu.device, mu.B * mu.C * mu.H * mu.W
    sym = (sym_out if sym_out is not None else torch.empty(n, dtype=torch.int32, device=dev)) if want_symbols else None
    idx = (idx_out if idx_out is not None else torch.empty(n, dtype=torch.int32, device=dev)) if want_indexes else None
    assert (sym is None or (sym.numel() == n and sym.dtype == torch.int32)) and (idx is None or (idx.numel() == n and idx.dtype == torch.int32))
    r = _rec()
    if r is not None:
        g = N.ScanGc()
        g.y, g.y_ld = (None, 0) if y is None else (y.ptr, y.ld)
        g.mu, g.mu_ld = mu.ptr, mu.ld
        g.scale, g.scale_ld = (None, 0) if scale is None else (scale.ptr, scale.ld)
        g.dtype, g.table, g.levels = mu.code, table.data_ptr(), table.numel()
        g.B, g.H, g.W, g.C = B, H, W, C
        g.symbols = None if sym is None else sym.data_ptr()
        g.indexes = None if idx is None else idx.data_ptr()
        g.y_hat, g.y_hat_ld = (None, 0) if y_hat is None else (y_hat.ptr, y_hat.ld)
        r.add(N.SCAN_GC_QUANTIZE, "gc", g, [table, sym, idx, mu.buf] + [x.buf for x in (y, scale, y_hat) if x is not None])
    N.check(N.lib().dsvc_gc_quantize(_stream(), None if y is None else y.ptr, 0 if y is None else y.ld, mu.ptr, mu.ld,
                                     None if scale is None else scale.ptr, 0 if scale is None else scale.ld, mu.code,
                                     table.data_ptr(), table.numel(), B, H, W, C,
                                     None if sym is None else sym.data_ptr(), None if idx is None else idx.data_ptr(),
                                     None if y_hat is None else y_hat.ptr, 0 if y_hat is None else y_hat.ld), "gc_quantize")
    return sym, idx


def gc_dequantize(symbols, mu, y_hat):
    r = _rec()
    if r is not None:
        g = N.ScanGc()
        g.symbols, g.mu, g.mu_ld, g.dtype = symbols.data_ptr(), mu.ptr, mu.ld, mu.code
        g.B, g.H, g.W, g.C = mu.B, mu.H, mu.W, mu.C
        g.y_hat, g.y_hat_ld = y_hat.ptr, y_hat.ld
        r.add(N.SCAN_GC_DEQUANTIZE, "gc", g, [symbols, mu.buf, y_hat.buf])
    N.check(N.lib().dsvc_gc_dequantize(_stream(), symbols.data_ptr(), mu.ptr, mu.ld, mu.code, mu.B, mu.H, mu.W, mu.C,
                                       y_hat.ptr, y_hat.ld), "gc_dequantize")
    return y_hat


def gc_likelihood(y, mu, scale, y_hat=None, lik=None):
    poison_recording("gc_likelihood")          # not a scan-step op
    N.check(N.lib().dsvc_gc_likelihood(_stream(), y.ptr, y.ld, mu.ptr, mu.ld, scale.ptr, scale.ld, mu.code, mu.npix, mu.C,
                                       None if y_hat is None else y_hat.ptr, 0 if y_hat is None else y_hat.ld,
                                       None if lik is None else lik.ptr, 0 if lik is None else lik.ld), "gc_likelihood")


def eb_process(z, params, symbols=False, z_hat=None, lik=None, sym_out=None):
    B, H, W, C = z.B, z.H, z.W, z.C
    sym = (sym_out if sym_out is not None else torch.empty(B * C * H * W, dtype=torch.int32, device=z.device)) if symbols else None
    mode = (1 if symbols else 0) | (2 if z_hat is not None else 0) | (4 if lik is not None else 0)
    r = _rec()
    if r is not None:
        e = N.ScanEb()
        e.z, e.z_ld, e.dtype, e.params = z.ptr, z.ld, z.code, params.data_ptr()
        e.B, e.H, e.W, e.C, e.mode = B, H, W, C, mode
        e.symbols = None if sym is None else sym.data_ptr()
        e.z_hat, e.z_hat_ld = (None, 0) if z_hat is None else (z_hat.ptr, z_hat.ld)
        e.lik, e.lik_ld = (None, 0) if lik is None else (lik.ptr, lik.ld)
        r.add(N.SCAN_EB_PROCESS, "eb", e, [params, sym, z.buf] + [x.buf for x in (z_hat, lik) if x is not None])
    N.check(N.lib().dsvc_eb_process(_stream(), z.ptr, z.ld, z.code, params.data_ptr(), B, H, W, C, mode,
                                    None if sym is None else sym.data_ptr(),
                                    None if z_hat is None else z_hat.ptr, 0 if z_hat is None else z_hat.ld,
                                    None if lik is None else lik.ptr, 0 if lik is None else lik.ld), "eb_process")
    return sym


def eb_dequantize(symbols, params, z_hat):
    r = _rec()
    if r is not None:
        e = N.ScanEb()
        e.symbols, e.params, e.dtype = symbols.data_ptr(), params.data_ptr(), z_hat.code
        e.B, e.H, e.W, e.C = z_hat.B, z_hat.H, z_hat.W, z_hat.C
        e.z_hat, e.z_hat_ld = z_hat.ptr, z_hat.ld
        r.add(N.SCAN_EB_DEQUANTIZE, "eb", e, [symbols, params, z_hat.buf])
    N.check(N.lib().dsvc_eb_dequantize(_stream(), symbols.data_ptr(), params.data_ptr(), z_hat.code, z_hat.B, z_hat.H, z_hat.W,
                                       z_hat.C, z_hat.ptr, z_hat.ld), "eb_dequantize")
    return z_hat


# --------------------------------------------------------------------------------------------
# host coder
# --------------------------------------------------------------------------------------------
class CdfTables:
    """Integer CDF tables (host, int32) in the layout the C coder takes."""

    def __init__(self, cdf, cdf_length, offset):
        self.cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        self.cdf_length = np.ascontiguousarray(cdf_length, dtype=np.int32).reshape(-1)
        self.offset = np.ascontiguousarray(offset, dtype=np.int32).reshape(-1)
        self.c = N.CdfTable(self.cdf.ctypes.data, self.cdf.shape[0], self.cdf.shape[1],
                            self.cdf_length.ctypes.data, self.offset.ctypes.data, None)
        self.lut = np.empty((self.cdf.shape[0], 256), dtype=np.uint16)          # decode accelerator, owned here
        N.check(N.lib().dsvc_cdf_build_lut(byref(self.c), self.lut.ctypes.data), "cdf_build_lut")
        self.c.lut = self.lut.ctypes.data

    def __del__(self):              # the registry entry must not outlive the LUT's memory (a recycled address would look registered)
        try:
            N.lib().dsvc_cdf_forget_lut(self.lut.ctypes.data)
        except Exception:
            pass


class DeviceCdfTables:
    """Device copies of a validated host table (cdf, cdf_length, offset, lut) for the multi-stream coder (include/dsvc.h: dsvc_rans_ms_*)."""

    def __init__(self, host, device):
        self.host = host
        self.cdf = torch.from_numpy(host.cdf).to(device)
        self.cdf_length = torch.from_numpy(host.cdf_length).to(device)
        self.offset = torch.from_numpy(host.offset).to(device)
        self.lut = torch.from_numpy(host.lut.view(np.int16)).to(device)           # (torch has no uint16 arithmetic: the bytes are what matters)
        self.c = N.CdfTable(self.cdf.data_ptr(), host.cdf.shape[0], host.cdf.shape[1], self.cdf_length.data_ptr(), self.offset.data_ptr(), self.lut.data_ptr())


def rans_ms_slot_words(symbols_per_stream):
    return N.check(N.lib().dsvc_rans_ms_slot_words(int(symbols_per_stream)), "rans_ms_slot_words")


def rans_ms_encode(symbols, indexes, n, n_streams, dev_tables, slots, slot_words, states, lens):
    """symbols / indexes: int32 device vectors (views); slots: int16 device [n_streams * slot_words]; states: int32 / lens: int16 device [n_streams]."""
    N.check(N.lib().dsvc_rans_ms_encode(_stream(), symbols.data_ptr(), indexes.data_ptr(), int(n), int(n_streams), byref(dev_tables.c),
                                        slots.data_ptr(), int(slot_words), states.data_ptr(), lens.data_ptr()), "rans_ms_encode")


def rans_ms_decode(indexes, n, n_streams, dev_tables, states_ptr, offsets_ptr, payload_ptr, payload_words, symbols_out):
    N.check(N.lib().dsvc_rans_ms_decode(_stream(), indexes.data_ptr(), int(n), int(n_streams), byref(dev_tables.c), states_ptr, offsets_ptr, payload_ptr,
                                        int(payload_words), symbols_out.data_ptr()), "rans_ms_decode")


def pack_small_inputs(acts, buf):
    """Copy (and cast) a run of small activations into consecutive channels of the zero-padded packed buffer `buf`."""
    c = 0
    for a in acts:
        eltwise(N.EW_COPY, a, out=Act(buf.buf, buf.c0 + c, a.C))
        c += a.C
    return buf


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.empty(len(p) + 1, dtype=np.int32)
    N.check(N.lib().dsvc_pmf_to_quantized_cdf(p.ctypes.data, len(p), precision, out.ctypes.data), "pmf_to_quantized_cdf")
    return out


class RansEncoder:
    """BufferedRansEncoder (image_model.py:221,253-254)."""

    def __init__(self):
        self.h = ctypes.c_void_p()
        N.check(N.lib().dsvc_rans_encoder_create(byref(self.h)), "rans_encoder_create")

    def encode_with_indexes(self, symbols, indexes, tables):
        s = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
        i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        assert s.shape == i.shape
        N.check(N.lib().dsvc_rans_encode_with_indexes(self.h, s.ctypes.data, i.ctypes.data, len(s), byref(tables.c)), "rans_encode")

    def flush(self):
        L = N.lib()
        n = N.check(L.dsvc_rans_flush(self.h, None, 0), "rans_flush(size)")
        buf = ctypes.create_string_buffer(n)
        N.check(L.dsvc_rans_flush(self.h, buf, n), "rans_flush")
        return buf.raw

    def __del__(self):
        try:
            N.lib().dsvc_rans_encoder_destroy(self.h)
        except Exception:
            pass


class RansDecoder:
    """RansDecoder.set_stream + decode_stream (image_model.py:273-274,288)."""

    def __init__(self, data):
        self.h = ctypes.c_void_p()
        N.check(N.lib().dsvc_rans_decoder_create(byref(self.h), data, len(data)), "rans_decoder_create")

    def decode_stream(self, indexes, tables, out=None):
        """out: optional int32 numpy vector to decode into (e.g. the view of a pinned host tensor)."""
        i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        if out is None:
            out = np.empty(len(i), dtype=np.int32)
        assert out.dtype == np.int32 and out.size == i.size and out.flags["C_CONTIGUOUS"]
        N.check(N.lib().dsvc_rans_decode_stream(self.h, i.ctypes.data, len(i), byref(tables.c), out.ctypes.data), "rans_decode")
        return out

    def __del__(self):
        try:
            N.lib().dsvc_rans_decoder_destroy(self.h)
        except Exception:
            pass
