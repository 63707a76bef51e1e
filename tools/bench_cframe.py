"""The frame-level C boundary at the headline size: export an f16 program blob at 1088x1920, then code BASELINE config 2's GOP (6 chained P-frames)
through dsvc_pframe_compress / dsvc_pframe_decompress alone (ctypes) and through the Python layer on the
same frames: strings and decoded frames must agree bit for bit; prints ms per frame of both.   python tools/bench_cframe.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, export, ops, synth

dev = torch.device("cuda:0")
H, W = 1088, 1920
ops.set_precision("f16")
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
t0 = time.perf_counter()
blob = export.export_pframe_program(m, H, W)
print(f"export: {len(blob) / 1e6:.0f} MB blob in {time.perf_counter() - t0:.1f} s", flush=True)
L = N.lib()


class Info(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int32), ("W", ctypes.c_int32), ("precision", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("feature_bytes", ctypes.c_int64), ("string_cap", ctypes.c_int64 * 4)]


prog = ctypes.c_void_p()
t0 = time.perf_counter()
N.check(L.dsvc_pframe_program_load(ctypes.c_char_p(blob), len(blob), 0, ctypes.byref(prog)), "load")
info = Info()
N.check(L.dsvc_pframe_program_info(prog, ctypes.byref(info)), "info")
print(f"load: {time.perf_counter() - t0:.2f} s; feature state {info.feature_bytes / 1e6:.0f} MB, string caps {[int(c) for c in info.string_cap]}", flush=True)
fr = [f.to(dev) for f in synth.make_frames(H, W, 7)]
sms = [None] + [synth.make_sm_fea(H, W, t).to(dev) for t in range(1, 7)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bufs = [ctypes.create_string_buffer(int(info.string_cap[i])) for i in range(4)]
outp = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
caps = (ctypes.c_int64 * 4)(*[int(info.string_cap[i]) for i in range(4)])


def c_gop():
    ref, feat, strs, recons = fr[0], None, [], []
    feats = [torch.empty(int(info.feature_bytes), dtype=torch.uint8, device=dev) for _ in range(2)]
    for t in range(1, 7):
        nb = (ctypes.c_int64 * 4)()
        N.check(L.dsvc_pframe_compress(prog, stream, P(ref), P(fr[t]), P(sms[t]), None if feat is None else P(feat), outp, caps, nb), "compress")
        s = [bufs[i].raw[:nb[i]] for i in range(4)]
        sb = [ctypes.create_string_buffer(x, len(x)) for x in s]
        sp = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in sb])
        sn = (ctypes.c_int64 * 4)(*[len(x) for x in s])
        fo = feats[t % 2]
        recon = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        N.check(L.dsvc_pframe_decompress(prog, stream, P(ref), sp, sn, P(sms[t]), None if feat is None else P(feat), P(fo), P(recon), None, None), "decompress")
        strs.append(s); recons.append(recon)
        ref, feat = recon, fo
    torch.cuda.synchronize()
    return strs, recons


def py_gop():
    ref, feat, strs, recons = fr[0], None, [], []
    for t in range(1, 7):
        mv, res = m.compress(ref, fr[t], sms[t], feat)
        s = [mv["strings"][0][0], mv["strings"][1][0], res["strings"][0][0], res["strings"][1][0]]
        feat, ref, _, _ = m.decompress(ref, mv, res, sms[t], feat)
        strs.append(s); recons.append(ref)
    torch.cuda.synchronize()
    return strs, recons


for name, fn in (("C frame API (dsvc_pframe_compress + dsvc_pframe_decompress)", c_gop), ("Python layer (one GOP in flight)", py_gop)):
    fn()
    t0 = time.perf_counter(); out = fn(); dt = (time.perf_counter() - t0) / 6
    print(f"{name}: {dt * 1e3:.2f} ms per frame = {1 / dt:.2f} fps", flush=True)
    if name.startswith("C"):
        c_out = out
same = all(a == b for a, b in zip(c_out[0], out[0])) and all(torch.equal(a, b) for a, b in zip(c_out[1], out[1]))
print("strings and decoded frames of the two paths identical over the GOP:", same)
L.dsvc_pframe_program_destroy(prog)
assert same
