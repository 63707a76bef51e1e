"""On-disk frame container for the codec's byte strings.

The reference keeps each frame's encoding as an in-memory dict `{"strings": [[y], [z, ...]], "shape": (h, w)}` per codec
and only ever sums the string lengths (test_video.py:309,364-365); nothing is written to disk.  This module defines the
missing container so that encoder and decoder can live in different processes.  The rANS payloads are stored untouched
(they are byte-compatible with compressai 1.2.1's coder), so the container adds a fixed 8 + 12 * codecs bytes per frame
(+ 4 per extra batch item) to the rate the harness reports:

    frame   := magic "DSVC" | u8 version=1 | u8 type (0 = I, 1 = P) | u8 n_codecs | u8 flags
    flags   := bits 0-1 arithmetic of the writer (0 = not stated, 1 = fp32 mode, 2 = f16 mode): the entropy parameters are floating-point
               convolutions, so a stream decodes bit-exactly only with the arithmetic that wrote it ("Stream interchange");
               bit 2 = payloads coded by the interleaved multi-stream range coder (FLAG_MULTISTREAM; 0 = compressai-compatible rANS)
               codec * n_codecs
    codec   := u16 shape_h | u16 shape_w | u32 len(y) | u16 n_z | u16 reserved | u32 len(z_i) * n_z | y | z_0 | z_1 ...

All integers little-endian.  A P-frame has two codecs in the order (motion, residual); an I-frame has one.
"""
import struct

MAGIC = b"DSVC"
VERSION = 1
TYPE_I, TYPE_P = 0, 1
FLAG_MODE_MASK, FLAG_FP32, FLAG_F16, FLAG_MULTISTREAM = 0x3, 0x1, 0x2, 0x4
MODE_FLAG = {"fp32": FLAG_FP32, "f16": FLAG_F16}


class ContainerError(ValueError):
    pass


def _pack_codec(enc):
    strings, shape = enc["strings"], enc["shape"]
    y_list, z_list = strings[0], strings[1]
    if len(y_list) != 1:
        raise ContainerError("one y string per frame expected (image_model.py:253-255 codes the whole batch into one)")
    y = bytes(y_list[0])
    zs = [bytes(z) for z in z_list]
    head = struct.pack("<HHIHH", int(shape[0]), int(shape[1]), len(y), len(zs), 0)
    return head + b"".join(struct.pack("<I", len(z)) for z in zs) + y + b"".join(zs)


def _unpack_codec(buf, pos):
    if len(buf) < pos + 12:
        raise ContainerError("truncated codec header")
    h, w, ny, nz, _ = struct.unpack_from("<HHIHH", buf, pos)
    pos += 12
    if len(buf) < pos + 4 * nz:
        raise ContainerError("truncated z-length table")
    zlens = struct.unpack_from("<%dI" % nz, buf, pos) if nz else ()
    pos += 4 * nz
    end = pos + ny + sum(zlens)
    if len(buf) < end:
        raise ContainerError("truncated payload")
    y = bytes(buf[pos:pos + ny])
    pos += ny
    zs = []
    for n in zlens:
        zs.append(bytes(buf[pos:pos + n]))
        pos += n
    return {"strings": [[y], zs], "shape": (h, w)}, pos


def pack_frame(*encodings, frame_type=None, flags=0):
    """pack_frame(i_enc) for an I-frame, pack_frame(mv_enc, res_enc) for a P-frame -> bytes.
    flags: e.g. MODE_FLAG[ops.get_precision()] so that a decoder in another process can refuse a stream of the other arithmetic."""
    if frame_type is None:
        frame_type = TYPE_I if len(encodings) == 1 else TYPE_P
    if not 1 <= len(encodings) <= 255:
        raise ContainerError("1..255 codecs per frame")
    if not 0 <= int(flags) <= 0x7:
        raise ContainerError("unknown flag bits")
    return MAGIC + struct.pack("<BBBB", VERSION, frame_type, len(encodings), int(flags)) + b"".join(_pack_codec(e) for e in encodings)


def frame_flags(buf):
    """The flags byte of a packed frame (see the module docstring)."""
    buf = bytes(buf[:8])
    if len(buf) < 8 or buf[:4] != MAGIC:
        raise ContainerError("not a DSVC frame")
    return buf[7]


def unpack_frame(buf, expect_mode=None):
    """-> (frame_type, [encoding dict, ...]); raises ContainerError on a malformed or truncated frame, and -- with expect_mode =
    "fp32" | "f16" -- on a frame whose writer stated the OTHER arithmetic (a frame that states none is accepted)."""
    buf = bytes(buf)
    if len(buf) < 8 or buf[:4] != MAGIC:
        raise ContainerError("not a DSVC frame")
    version, frame_type, n, flags = struct.unpack_from("<BBBB", buf, 4)
    if version != VERSION:
        raise ContainerError(f"unsupported container version {version}")
    if flags & ~0x7:
        raise ContainerError(f"unknown flag bits {flags:#x}")
    if expect_mode is not None and (flags & FLAG_MODE_MASK) not in (0, MODE_FLAG[expect_mode]):
        raise ContainerError(f"frame written in {'fp32' if flags & FLAG_MODE_MASK == FLAG_FP32 else 'f16'} mode, decoder runs {expect_mode}")
    pos, out = 8, []
    for _ in range(n):
        enc, pos = _unpack_codec(buf, pos)
        out.append(enc)
    if pos != len(buf):
        raise ContainerError(f"{len(buf) - pos} trailing bytes")
    return frame_type, out


def frame_overhead_bytes(*encodings):
    """Bytes the container adds on top of sum(len(s[0]) for s in strings) of test_video.py:364-365 (first batch item)."""
    return 8 + sum(12 + 4 * len(e["strings"][1]) + sum(len(z) for z in e["strings"][1][1:]) for e in encodings)
