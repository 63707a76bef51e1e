"""Frame container: round trip, overhead accounting, malformed input.  CPU only."""
import pytest

from deepsvc_amd import bitstream as bs


def _enc(ny, nz, shape, batch=1):
    return {"strings": [[bytes(range(256)) * (ny // 256) + bytes(ny % 256)], [bytes([7 + i]) * nz for i in range(batch)]], "shape": shape}


def test_round_trip_p_and_i_frames():
    mv, res = _enc(1000, 13, (17, 30)), _enc(70000, 40, (17, 30))
    blob = bs.pack_frame(mv, res)
    ftype, (mv2, res2) = bs.unpack_frame(blob)
    assert ftype == bs.TYPE_P and mv2 == mv and res2 == res
    payload = sum(len(s[0]) for e in (mv, res) for s in e["strings"])
    assert len(blob) == payload + bs.frame_overhead_bytes(mv, res) == payload + 8 + 2 * 16
    i_enc = _enc(0, 0, (1, 1))                       # empty strings are legal
    ftype, (i2,) = bs.unpack_frame(bs.pack_frame(i_enc))
    assert ftype == bs.TYPE_I and i2 == i_enc
    b2 = _enc(300, 5, (2, 3), batch=2)               # two z strings (batch of two, image_model.py:206)
    assert bs.unpack_frame(bs.pack_frame(b2))[1][0] == b2


def test_pending_dicts_are_accepted():
    """The codec returns lazy dict subclasses; the container only needs the mapping interface."""
    class Lazy(dict):
        pass
    e = Lazy(_enc(10, 2, (1, 1)))
    assert bs.unpack_frame(bs.pack_frame(e))[1][0] == dict(e)


@pytest.mark.parametrize("mangle", ["magic", "version", "truncate", "trailing"])
def test_malformed_frames_raise(mangle):
    blob = bytearray(bs.pack_frame(_enc(100, 4, (1, 1)), _enc(50, 4, (1, 1))))
    if mangle == "magic":
        blob[0] ^= 0xFF
    elif mangle == "version":
        blob[4] = 9
    elif mangle == "truncate":
        blob = blob[:-3]
    else:
        blob += b"\x00"
    with pytest.raises(bs.ContainerError):
        bs.unpack_frame(bytes(blob))


def test_flags_byte_states_the_writer_arithmetic():
    """INTEGRATION.md "Stream interchange": the entropy parameters are floating-point convolutions, so a frame decodes bit-exactly only
    with the arithmetic that wrote it; the container can say which one that was."""
    mv, res = _enc(100, 4, (1, 1)), _enc(50, 4, (1, 1))
    blob = bs.pack_frame(mv, res, flags=bs.MODE_FLAG["f16"])
    assert bs.frame_flags(blob) == bs.FLAG_F16 and len(blob) == len(bs.pack_frame(mv, res))
    assert bs.unpack_frame(blob, expect_mode="f16")[1] == [mv, res]
    assert bs.unpack_frame(blob)[1] == [mv, res]                       # a decoder that does not ask is not refused
    with pytest.raises(bs.ContainerError):
        bs.unpack_frame(blob, expect_mode="fp32")
    assert bs.unpack_frame(bs.pack_frame(mv, res), expect_mode="fp32")[1] == [mv, res]      # unstated mode: accepted
    with pytest.raises(bs.ContainerError):
        bs.pack_frame(mv, res, flags=0x80)
    bad = bytearray(blob); bad[7] = 0x40
    with pytest.raises(bs.ContainerError):
        bs.unpack_frame(bytes(bad))
