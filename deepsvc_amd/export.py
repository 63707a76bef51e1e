"""Export a warmed `DeepSVC` as a self-contained P-frame PROGRAM BLOB for the C-level frame API of include/dsvc.h
(`dsvc_pframe_program_load / dsvc_pframe_compress / dsvc_pframe_decompress`; SURVEY.md section 8b "frame level": the reference's unit of
work is one call, `DeepSVC.compress` / `DeepSVC.decompress`, video_model.py:137-151,153-167, called at test_video.py:342,349).

The Python layer records every host-synchronisation-free segment of a P-frame once (ops.FrameProgram: E1, E2a, E2, D1, DR, D2) and every
dependency level of the two decoder scans (ops.ScanProgram: dec0..dec5); their op arrays are what `dsvc_pframe_segment_run` /
`dsvc_entropy_scan_step` replay.  This module serialises those arrays together with everything they point to, so that a caller holding only
`libdsvc.so`, `dsvc.h` and the blob can code frames -- no Python, no torch:

  header | tensor table | programs (op arrays + relocation lists) | codec tables (scan geometry, staging buffers, integer CDF tables) | payload

  * every pointer inside an op is replaced by a relocation (kind, index, delta):
      TENSOR  a persistent device buffer of the model -- packed weights, biases, entropy-model tables (content stored in the blob), scan
              workspaces and segment outputs that a later segment reads (zero-initialised at load);
      ARENA   the per-stream workspace the segment's temporaries were packed into by lifetime (ops.FrameProgram);
      SLOT    a frame tensor: ref / cur (NHWC fp32 copies the library makes from the caller's NCHW tensors), sm_fea, the carried
              feature (in and out), the three decoded frames;
  * both frame variants are stored: FIRST P-frame of a GOP (feature = None, modules.py:425-428) and STEADY state;
  * the integer CDF tables of the four entropy models ride along for the host range coder.

    blob = export_pframe_program(model, H, W)         # model: a DeepSVC on the GPU, update()d, in the precision mode to export
"""
import ctypes
import struct

import numpy as np
import torch

from . import _native as N
from . import ops

MAGIC = b"DSVCPRG1"
VERSION = 1
KIND_TENSOR, KIND_ARENA0, KIND_ARENA1, KIND_SLOT = 0, 1, 2, 3
# frame tensors (include/dsvc.h: dsvc_pframe_compress / _decompress)
SLOT_REF, SLOT_CUR, SLOT_SM_FEA, SLOT_FEATURE_IN, SLOT_FEATURE_OUT, SLOT_RECON, SLOT_WARPED, SLOT_PREDICT = range(8)
SEGMENTS = ("E1", "E2a", "E2", "D1", "DR", "D2")


def _storage_key(t):
    s = t.untyped_storage()
    return s.data_ptr(), s.nbytes()


def _tensors_in(obj, out):
    """Every torch tensor reachable from an entry of a recording's keep list (tensors, PackedConv / ItemPackedConv objects, Acts)."""
    if obj is None:
        return
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, ops.Act):
        out.append(obj.buf)
    elif isinstance(obj, (ops.PackedConv, ops.ItemPackedConv)):
        out.append(obj.weight)
        if obj.bias is not None:
            out.append(obj.bias)
        for src in getattr(obj, "sources", ()):
            _tensors_in(src, out)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors_in(o, out)


class _Registry:
    """Persistent device storages referenced by the exported programs: storage -> table index."""

    def __init__(self):
        self.by_ptr, self.entries = {}, []          # (ptr, nbytes) -> index; [(storage-owning tensor, zero_init)]

    def add(self, t, zero=False):
        key = _storage_key(t)
        if key[1] == 0:
            return None
        i = self.by_ptr.get(key)
        if i is None:
            i = self.by_ptr[key] = len(self.entries)
            self.entries.append([t, zero])
        elif zero:
            self.entries[i][1] = True
        return i

    def ranges(self):
        return sorted((ptr, ptr + nb, i) for (ptr, nb), i in self.by_ptr.items())


def _field_offsets(arr, n):
    """(byte offset of every pointer field of every op in the array, its current value)."""
    base, step, uoff = ctypes.addressof(arr), ctypes.sizeof(N.ScanOp), N.ScanOp.u.offset
    offs = {m: N.pointer_offsets(dict(N._ScanU._fields_)[m]) for m in set(N.SCAN_UNION_MEMBER.values())}
    for i in range(n):
        for o in offs[N.SCAN_UNION_MEMBER[arr[i].kind]]:
            off = i * step + uoff + o
            v = ctypes.c_uint64.from_address(base + off).value
            if v:
                yield off, v


def _resolve(reg_ranges, v):
    import bisect
    k = bisect.bisect_right([r[0] for r in reg_ranges], v) - 1
    if k >= 0 and reg_ranges[k][0] <= v < reg_ranges[k][1]:
        return reg_ranges[k][2], v - reg_ranges[k][0]
    return None


def _segment_slots(name, has_feature, persist):
    """Binding of a frame segment's externals (ops.FrameProgram slot order: inputs, then fresh outputs) to frame tensors (SLOT, id) or to
    persistent outputs of an earlier segment (TENSOR, Act) -- exactly the lists video_model.DeepSVC.compress / decompress pass."""
    S = lambda i: ("slot", i)
    T = lambda a: ("tensor", a)
    feat = [S(SLOT_FEATURE_IN)] if has_feature else []
    if name == "E1":
        return [S(SLOT_REF), S(SLOT_CUR)]
    if name in ("E2a", "DR"):
        return []
    if name == "E2":
        return [S(SLOT_REF), S(SLOT_CUR), S(SLOT_SM_FEA), T(persist["E2a"][0])] + feat
    if name == "D1":
        return [S(SLOT_REF), S(SLOT_SM_FEA)] + feat
    if name == "D2":
        return [S(SLOT_REF)] + [T(a) for a in persist["D1"]] + [T(persist["DR"][0])] + \
               [S(SLOT_FEATURE_OUT), S(SLOT_RECON), S(SLOT_WARPED), S(SLOT_PREDICT)]
    raise KeyError(name)


def export_pframe_program(model, H, W, frames=None, sm_feas=None):
    """-> bytes.  `model`: DeepSVC on a GPU, tables built (update), frame programs enabled; (H, W): padded frame size (multiples of 64).
    Runs two chained P-frames (first + steady state) to make sure every segment of both variants is recorded, then serialises them."""
    from . import synth
    dev = next(model.parameters()).device
    assert H % 64 == 0 and W % 64 == 0 and ops.FRAME_PROGRAMS[0] and ops.SCAN_STEP[0], "export needs recorded frame / scan programs"
    frames = frames or [f.to(dev) for f in synth.make_frames(H, W, 4)]
    sm_feas = sm_feas or [None] + [synth.make_sm_fea(H, W, t).to(dev) for t in (1, 2, 3)]
    ref, feat = frames[0], None
    for t in (1, 2, 3):                       # frame 1: first variant recorded; frames 2-3: steady variant recorded, scan graphs captured
        mv, res = model.compress(ref, frames[t], sm_feas[t], feat)
        feat, ref, _, _ = model.decompress(ref, dict(mv.items()), dict(res.items()), sm_feas[t], feat)
    torch.cuda.synchronize(dev)
    cache = model.__dict__["_frame_programs"]
    arenas = model.__dict__["_arenas"]
    variants = {}
    for (name, key), prog in cache.items():
        if not isinstance(prog, ops.FrameProgram):
            raise RuntimeError(f"segment {name} stayed launch-by-launch (an op without a scan-op form): cannot be exported")
        variants.setdefault(name, []).append((key, prog))
    for name in SEGMENTS:
        assert name in variants and len(variants[name]) == 2, f"expected a first-frame and a steady-state recording of {name}, got {len(variants.get(name, []))}"
    # variant 0 = first P-frame (feature None): the key's last element (sig of the feature) is None there
    def split(name):
        a, b = variants[name]
        first = a if a[0][-1] is None else b
        steady = b if first is a else a
        assert first[0][-1] is None and steady[0][-1] is not None
        return first[1], steady[1]
    progs = {name: split(name) for name in SEGMENTS}

    reg = _Registry()
    arena_ranges = []
    for k, a in enumerate(arenas):
        if a.buf is not None:
            arena_ranges.append((a.buf.data_ptr(), a.buf.data_ptr() + a.buf.numel(), k))
    # ---- persistent storages: everything the recordings keep alive, minus arenas and frame tensors ----
    out_acts = {}
    for v in (0, 1):
        persist = {name: progs[name][v].persistent for name in SEGMENTS}
        out_acts[v] = persist
        for name in SEGMENTS:
            p = progs[name][v]
            for a in p.persistent:
                reg.add(a.buf, zero=True)                     # segment outputs: rewritten by every run
            ts = []
            _tensors_in(p.keep, ts)
            for t in ts:
                reg.add(t)
    codecs = (model.mv_codec, model.res_codec)
    plans = [c._plan(1, H // 16, W // 16, dev) for c in codecs]
    dec_progs = []
    for c, ws in zip(codecs, plans):
        steps = [k for k in c._segments(ws) if k.startswith("dec")]
        row = []
        for k in steps:
            sp = ws.programs.get(k)
            if not isinstance(sp, ops.ScanProgram):
                raise RuntimeError(f"decoder scan segment {k} has no recorded scan program")
            ts = []
            _tensors_in(sp.keep, ts)
            for t in ts:
                reg.add(t)
            row.append((k, sp))
        dec_progs.append(row)
        for t in (ws.staging, ws.z_sym, ws.idx_dev, ws.sym_dev, ws.y.buf, ws.y_hat.buf, ws.pair.buf, ws.z_hat.buf, ws.musc.buf):
            reg.add(t, zero=True)                              # scan workspaces: fully rewritten before they are read
    ranges = reg.ranges()

    def classify(v):
        for lo, hi, k in arena_ranges:
            if lo <= v < hi:
                return (KIND_ARENA0 + k, 0, v - lo)
        r = _resolve(ranges, v)
        if r is None:
            raise RuntimeError(f"pointer {v:#x} in a recorded op belongs to no known buffer: cannot be exported")
        return (KIND_TENSOR, r[0], r[1])

    # ---- programs ----
    records = []          # (name, variant, n_ops, ops bytes, [(field_off, kind, index, delta)])
    for v in (0, 1):
        for name in SEGMENTS:
            p = progs[name][v]
            base = ctypes.addressof(p.arr)
            slots = _segment_slots(name, v == 1, out_acts[v])
            assert len(slots) == p.n_inputs + len(p.fresh_meta), (name, len(slots), p.n_inputs, len(p.fresh_meta))
            ext_fields = {}
            for cell, slot, delta in p.ext_rel:
                ext_fields[ctypes.addressof(cell) - base] = (slot, delta)
            relocs = []
            for off, val in _field_offsets(p.arr, p.n):
                if off in ext_fields:
                    slot, delta = ext_fields[off]
                    kind, what = slots[slot]
                    if kind == "slot":
                        relocs.append((off, KIND_SLOT, what, delta))
                    else:                                  # a persistent output of an earlier segment (an Act)
                        k2, idx, d2 = classify(what.buf.data_ptr())
                        relocs.append((off, k2, idx, d2 + delta))
                else:
                    relocs.append((off,) + classify(val))
            records.append((name, v, p.n, bytes(p.arr)[: p.n * ctypes.sizeof(N.ScanOp)], relocs))
    for ci, row in enumerate(dec_progs):
        for k, sp in row:
            relocs = [(off,) + classify(val) for off, val in _field_offsets(sp.arr, sp.n)]
            records.append((f"c{ci}.{k}", 2, sp.n, bytes(sp.arr)[: sp.n * ctypes.sizeof(N.ScanOp)], relocs))

    # ---- payload assembly ----
    payload = bytearray()

    def put(b):
        pad = (-len(payload)) % 16
        payload.extend(b"\0" * pad)
        off = len(payload)
        payload.extend(b)
        return off

    tensor_rows = []
    for t, zero in reg.entries:
        ptr, nb = _storage_key(t)
        if zero:
            tensor_rows.append((nb, -1))
        else:
            flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(t.untyped_storage(), 0, (nb,))
            tensor_rows.append((nb, put(flat.cpu().numpy().tobytes())))
    prog_rows = []
    for name, v, n, ob, relocs in records:
        o_ops = put(ob)
        o_rel = put(b"".join(struct.pack("<IIIIQ", off, kind, idx, 0, delta) for off, kind, idx, delta in relocs))
        prog_rows.append((name.encode()[:15].ljust(16, b"\0"), v, n, len(relocs), o_ops, o_rel))
    codec_rows = []
    for c, ws in zip(codecs, plans):
        groups = [len(g) for g in ws.groups]
        refs = [classify(t.data_ptr()) for t in (ws.staging, ws.z_sym, ws.idx_dev, ws.sym_dev)]
        assert all(r[0] == KIND_TENSOR for r in refs)
        tabs = []
        for em in (c.entropy_bottleneck, c.gaussian_conditional):
            ht = em.host_tables()
            tabs.append((ht.cdf.shape[0], ht.cdf.shape[1], put(ht.cdf.astype("<i4").tobytes()), put(ht.cdf_length.astype("<i4").tobytes()),
                         put(ht.offset.astype("<i4").tobytes())))
        codec_rows.append((c.M, c.N, ws.z_hat.H, ws.z_hat.W, c.slice_depth, ws.y.H, ws.y.W, groups, refs, tabs))

    # ---- header ----
    head = bytearray()
    head += MAGIC + struct.pack("<IIIIII", VERSION, 0 if ops.RT.mode == "fp32" else 1, H, W, len(tensor_rows), len(prog_rows))
    head += struct.pack("<II", len(codec_rows), ctypes.sizeof(N.ScanOp))
    head += struct.pack("<QQ", *(int(a.buf.numel()) if a.buf is not None else 0 for a in arenas))
    fm = progs["D2"][1].fresh_meta[0]                        # the carried feature: shape [1,H,W,64], storage dtype
    head += struct.pack("<II", fm[0][3], 4 if fm[1] == torch.float32 else 2)
    tt = b"".join(struct.pack("<Qq", nb, off) for nb, off in tensor_rows)
    pt = b"".join(struct.pack("<16sIIIIQQ", nm, v, n, nr, 0, oo, orl) for nm, v, n, nr, oo, orl in prog_rows)
    ct = b""
    for M, Nz, zh, zw, d, yh, yw, groups, refs, tabs in codec_rows:
        ct += struct.pack("<IIIIIIII", M, Nz, zh, zw, d, yh, yw, len(groups)) + struct.pack("<8I", *(groups + [0] * (8 - len(groups))))
        for kind, idx, delta in refs:
            ct += struct.pack("<IIQ", kind, idx, delta)
        for rows, stride, o_cdf, o_len, o_off in tabs:
            ct += struct.pack("<IIQQQ", rows, stride, o_cdf, o_len, o_off)
    fixed = len(head) + 8 + len(tt) + len(pt) + len(ct)
    pay_off = -(-fixed // 16) * 16
    blob = bytes(head) + struct.pack("<Q", pay_off) + tt + pt + ct
    blob += b"\0" * (pay_off - len(blob)) + bytes(payload)
    return blob
