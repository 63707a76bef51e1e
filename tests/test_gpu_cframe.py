"""The frame-level C boundary (include/dsvc.h: dsvc_pframe_program_load / dsvc_pframe_compress / dsvc_pframe_decompress; SURVEY section 8b
"frame level"): `DeepSVC.compress / decompress` (video_model.py:137-151,153-167) reachable from `libdsvc.so` +
`dsvc.h` + a program blob alone.

The parent process exports the blob from a warmed model (deepsvc_amd/export.py) and writes the inputs; a CHILD process that imports
only ctypes, numpy and torch (device memory) -- never deepsvc_amd.ops / modules / video_model -- loads the library and the blob and codes
BASELINE config 1's septuplet (six chained P-frames at 256x256: first-frame variant, then steady state) through the two C calls.  Its byte
strings must be the REFERENCE's own (tests/golden/deepsvc_ref_256.npz, fp32 mode) and its decoded frames the reference's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

CHILD = r'''
import ctypes, json, sys
import numpy as np
import torch
root, blob_path, io_path, out_path = sys.argv[1:5]
L = ctypes.CDLL(root + "/deepsvc_amd/lib/libdsvc.so")
L.dsvc_last_error.restype = ctypes.c_char_p
class Info(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int32), ("W", ctypes.c_int32), ("precision", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("feature_bytes", ctypes.c_int64), ("string_cap", ctypes.c_int64 * 4)]
def ok(rc, what):
    if rc != 0:
        raise RuntimeError(what + ": " + L.dsvc_last_error().decode())
blob = open(blob_path, "rb").read()
prog = ctypes.c_void_p()
ok(L.dsvc_pframe_program_load(ctypes.c_char_p(blob), ctypes.c_int64(len(blob)), 0, ctypes.byref(prog)), "load")
info = Info()
ok(L.dsvc_pframe_program_info(prog, ctypes.byref(info)), "info")
io = np.load(io_path)
frames, sms = io["frames"], io["sm"]
n = frames.shape[0] - 1
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
ref = torch.from_numpy(frames[0]).to(dev)
feature = None
bufs = [ctypes.create_string_buffer(int(info.string_cap[i])) for i in range(4)]
out_ptrs = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
caps = (ctypes.c_int64 * 4)(*[int(info.string_cap[i]) for i in range(4)])
result = {"strings": [], "recon": [], "psnr": []}
for t in range(1, n + 1):
    cur, sm = torch.from_numpy(frames[t]).to(dev), torch.from_numpy(sms[t - 1]).to(dev)
    nb = (ctypes.c_int64 * 4)()
    ok(L.dsvc_pframe_compress(prog, stream, P(ref), P(cur), P(sm), None if feature is None else P(feature), out_ptrs, caps, nb), "compress")
    strs = [bufs[i].raw[:nb[i]] for i in range(4)]
    # a separate decoder would receive exactly these bytes
    sbufs = [ctypes.create_string_buffer(s, len(s)) for s in strs]
    sp = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in sbufs])
    sn = (ctypes.c_int64 * 4)(*[len(s) for s in strs])
    feat_out = torch.empty(int(info.feature_bytes), dtype=torch.uint8, device=dev)
    recon, warped, predict = (torch.empty((1, 3, info.H, info.W), dtype=torch.float32, device=dev) for _ in range(3))
    ok(L.dsvc_pframe_decompress(prog, stream, P(ref), sp, sn, P(sm), None if feature is None else P(feature), P(feat_out), P(recon), P(warped), P(predict)), "decompress")
    torch.cuda.synchronize()
    result["strings"].append([s.hex() for s in strs])
    result["recon"].append(recon.cpu().numpy())
    result["psnr"].append(float(-10 * torch.log10(((recon - cur) ** 2).mean())))
    ref, feature = recon, feat_out
L.dsvc_pframe_program_destroy(prog)
assert not any(m.startswith("deepsvc_amd") for m in sys.modules), "the child must not import the Python layer"
np.savez(out_path, recon=np.stack(result["recon"]), psnr=np.asarray(result["psnr"]), strings=json.dumps(result["strings"]))
print("child ok", len(result["strings"]), "frames")
'''


def _export(precision, H, W, tmp_path):
    from deepsvc_amd import export, ops, synth
    ops.set_precision(precision)
    try:
        m = synth.make_model("W0"); m.update(force=True); m = m.to(DEV)
        blob = export.export_pframe_program(m, H, W)
    finally:
        ops.set_precision("fp32")
    path = tmp_path / f"pframe_{precision}_{H}x{W}.dsvcprog"
    path.write_bytes(blob)
    return str(path), len(blob)


def _run_child(blob_path, io_path, out_path, tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT, blob_path, io_path, out_path], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"}, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + "\n" + r.stderr[-3000:]
    return np.load(out_path)


def test_c_frame_api_codes_config1_septuplet_byte_identical_to_the_reference(golden256, tmp_path):
    from deepsvc_amd import synth
    import parity_util as PU
    H = W = 256
    blob_path, nbytes = _export("fp32", H, W, tmp_path)
    fr = synth.make_frames(H, W, 7)
    io_path, out_path = str(tmp_path / "io.npz"), str(tmp_path / "out.npz")
    np.savez(io_path, frames=np.stack([f.numpy() for f in fr]), sm=np.stack([synth.make_sm_fea(H, W, t).numpy() for t in range(1, 7)]))
    got = _run_child(blob_path, io_path, out_path, tmp_path)
    strings = json.loads(str(got["strings"]))
    g = golden256
    for t in range(1, 7):
        want = [g[f"s256_gop/f{t}/enc/{c}_{k}"].tobytes() for c in ("mv", "res") for k in ("y", "z")]
        mine = [bytes.fromhex(s) for s in strings[t - 1]]
        assert [len(x) for x in mine] == [len(x) for x in want], f"frame {t}: string lengths {[len(x) for x in mine]} vs the reference's {[len(x) for x in want]}"
        assert mine == want, f"frame {t}: the C frame API's strings differ from the reference's"
        PU.check_golden(g, f"s256_gop/f{t}/dec/recon", torch.from_numpy(got["recon"][t - 1]), 1e-5)
    np.testing.assert_allclose(got["psnr"], g["s256_gop/psnr"], atol=1e-3)
    print(f"[cframe] blob {nbytes / 1e6:.1f} MB; six chained P-frames through dsvc_pframe_compress / _decompress: bytes == the reference's, PSNR {got['psnr'][0]:.3f} .. {got['psnr'][-1]:.3f} dB")


def test_c_frame_api_equals_the_python_layer_in_f16_mode(tmp_path):
    """The fast mode through the C calls: same strings and the same decoded frames, bit for bit, as DeepSVC.compress / decompress of the
    Python layer on the same frames (two chained P-frames at 256x1024: the large-image kernels)."""
    from deepsvc_amd import ops, synth
    H, W = 256, 1024
    blob_path, _ = _export("f16", H, W, tmp_path)
    fr = synth.make_frames(H, W, 3)
    sms = [synth.make_sm_fea(H, W, t) for t in (1, 2)]
    io_path, out_path = str(tmp_path / "io.npz"), str(tmp_path / "out.npz")
    np.savez(io_path, frames=np.stack([f.numpy() for f in fr]), sm=np.stack([s.numpy() for s in sms]))
    got = _run_child(blob_path, io_path, out_path, tmp_path)
    strings = json.loads(str(got["strings"]))
    ops.set_precision("f16")
    try:
        m = synth.make_model("W0"); m.update(force=True); m = m.to(DEV)
        ref, feat = fr[0].to(DEV), None
        for t in (1, 2):
            mv, res = m.compress(ref, fr[t].to(DEV), sms[t - 1].to(DEV), feat)
            want = [mv["strings"][0][0], mv["strings"][1][0], res["strings"][0][0], res["strings"][1][0]]
            assert [bytes.fromhex(s) for s in strings[t - 1]] == want, f"frame {t}: C frame API and Python layer disagree"
            feat, ref, _, _ = m.decompress(ref, mv, res, sms[t - 1].to(DEV), feat)
            assert np.array_equal(got["recon"][t - 1], ref.cpu().numpy()), f"frame {t}: decoded frames differ"
    finally:
        ops.set_precision("fp32")


def test_c_frame_api_refuses_malformed_blobs(tmp_path):
    """dsvc_pframe_program_load on damaged blobs: every truncation point sampled, a wrong magic / version / ABI size, a non-padded frame size,
    table counts and payload offsets beyond the file, tensors and CDF tables pointing outside the blob -- each one must come back as a
    negative status with a message (never a crash, never a handle), and the intact blob must still load afterwards."""
    import ctypes
    import struct
    from deepsvc_amd import _native as N
    path, n = _export("fp32", 64, 64, tmp_path)
    blob = open(path, "rb").read()
    L = ctypes.CDLL(N.LIB_PATH)
    L.dsvc_last_error.restype = ctypes.c_char_p
    L.dsvc_pframe_program_load.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    L.dsvc_pframe_program_destroy.argtypes = [ctypes.c_void_p]

    def load(b):
        h = ctypes.c_void_p()
        rc = L.dsvc_pframe_program_load(bytes(b), len(b), 0, ctypes.byref(h))
        return rc, h, (L.dsvc_last_error() or b"").decode()

    def refused(b, what):
        rc, h, msg = load(b)
        assert rc < 0 and not h.value and msg, f"{what}: rc {rc}, handle {h.value}, message {msg!r}"

    HEAD = struct.Struct("<8s8I2Q2IQ")                       # csrc/pframe_program.cpp: BlobHead (72 bytes)
    assert HEAD.size == 72
    f = list(HEAD.unpack_from(blob))
    magic, version, precision, H, W, n_tensors, n_programs, n_codecs, op_size, a0, a1, fch, fes, payload_off = f
    assert magic == b"DSVCPRG1" and (H, W) == (64, 64)

    def with_head(**kw):
        g = dict(magic=magic, version=version, precision=precision, H=H, W=W, n_tensors=n_tensors, n_programs=n_programs, n_codecs=n_codecs,
                 op_size=op_size, a0=a0, a1=a1, fch=fch, fes=fes, payload_off=payload_off)
        g.update(kw)
        return HEAD.pack(*[g[k] for k in ("magic", "version", "precision", "H", "W", "n_tensors", "n_programs", "n_codecs", "op_size", "a0", "a1", "fch", "fes", "payload_off")]) + blob[72:]

    for cut in [0, 7, 40, 71, 72, 100, payload_off - 1, payload_off, payload_off + 17, len(blob) // 3, len(blob) // 2, len(blob) - 4096, len(blob) - 1]:
        refused(blob[:max(cut, 0)], f"truncated to {cut} bytes")
    refused(with_head(magic=b"DSVCPRG2"), "magic")
    refused(with_head(version=2), "version")
    refused(with_head(op_size=op_size + 8), "ABI op size")
    refused(with_head(H=72), "frame height not a multiple of 64")
    refused(with_head(precision=7), "precision")
    refused(with_head(n_codecs=3), "codec count")
    refused(with_head(n_tensors=n_tensors + (1 << 20)), "tensor table beyond the payload offset")
    refused(with_head(n_programs=n_programs + (1 << 20)), "program table beyond the payload offset")
    refused(with_head(payload_off=len(blob) + 1), "payload offset beyond the file")
    # first tensor: data beyond the blob / absurd size
    bad = bytearray(blob)
    struct.pack_into("<Qq", bad, 72, 1 << 20, len(blob))
    refused(bad, "tensor data outside the blob")
    bad = bytearray(blob)
    struct.pack_into("<Qq", bad, 72, 1 << 40, 0)
    refused(bad, "tensor of 2^40 bytes")
    # first codec: a CDF table pointing outside the payload (BlobCodec = 192 bytes: 16 uint32, 4 x 16-byte refs, 2 x 32-byte tables)
    codec0 = 72 + 16 * n_tensors + 48 * n_programs
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, codec0 + 64 + 64 + 8, len(blob))          # tables[0].cdf_off
    refused(bad, "CDF table outside the blob")
    bad = bytearray(blob)
    struct.pack_into("<II", bad, codec0 + 64, 9, 1 << 30)                  # refs[0] = {kind 9, tensor index 2^30}
    refused(bad, "codec buffer reference")
    # 64-bit offsets that WRAP when a length is added to them: `off + len <= size` passes for off near 2^64;
    # the loader must compare `len <= size - off` instead, and bound rows / stride before multiplying them
    U64 = (1 << 64) - 1
    prog0 = 72 + 16 * n_tensors                                             # BlobProgram: char[16], 4 x uint32, ops_off, relocs_off
    for name, at, fmt, val in [("ops_off near 2^64", prog0 + 32, "<Q", U64 - 15), ("relocs_off near 2^64", prog0 + 40, "<Q", U64 - 7),
                               ("tables[0].cdf_off near 2^64", codec0 + 128 + 8, "<Q", U64 - 3), ("tables[0].len_off near 2^64", codec0 + 128 + 16, "<Q", U64 - 3),
                               ("tables[1].off_off near 2^64", codec0 + 160 + 24, "<Q", U64), ("refs[0].delta near 2^64", codec0 + 64 + 8, "<Q", U64 - 63),
                               ("refs[2].delta beyond its tensor", codec0 + 64 + 2 * 16 + 8, "<Q", 1 << 33)]:
        bad = bytearray(blob)
        struct.pack_into(fmt, bad, at, val)
        refused(bad, name)
    bad = bytearray(blob)
    struct.pack_into("<II", bad, codec0 + 128, 1 << 31, 1 << 31)           # rows * stride * 4 = 2^64: wraps to 0
    refused(bad, "CDF table whose rows x stride x 4 wraps")
    bad = bytearray(blob)
    struct.pack_into("<II", bad, codec0 + 128, (1 << 32) - 1, 2)
    refused(bad, "CDF table with 2^32 - 1 rows")
    bad = bytearray(blob)
    struct.pack_into("<4I", bad, codec0 + 8, 1 << 30, 1 << 30, 8, 1 << 30)  # zh, zw, d, yh: products overflow int64 sizes
    refused(bad, "codec geometry beyond any picture")
    bad = bytearray(blob)
    nrel = struct.unpack_from("<I", blob, prog0 + 24)[0]
    rel0 = payload_off + struct.unpack_from("<Q", blob, prog0 + 40)[0]      # first relocation of the first program: {field_off, kind, index, pad, delta}
    assert nrel > 0
    struct.pack_into("<Q", bad, rel0 + 16, U64 - 8)
    refused(bad, "relocation delta near 2^64")
    rc, h, msg = load(blob)
    assert rc == 0 and h.value, f"the intact blob no longer loads: {msg}"
    L.dsvc_pframe_program_destroy(h)
