"""CPU-side checks of the product library: it loads without a GPU, exports every symbol include/dsvc.h declares,
and its HOST functions (range coder, CDF construction, weight packing) agree with the oracle.
No device kernel is launched here."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deepsvc_amd import _native as N
from deepsvc_amd import ops
from oracle import entropy as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dsvc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsvc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 30
    L = N.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dsvc.h but not exported by libdsvc.so"
        assert n in N.SIGNATURES, f"{n} has no ctypes signature in deepsvc_amd/_native.py"
    assert set(N.SIGNATURES) == set(names)
    assert L.dsvc_version() == N.ABI_VERSION == 3


def test_conv_desc_ctypes_mirror_matches_the_header(tmp_path):
    """The ctypes mirror of dsvc_conv_desc must lay out exactly like the C struct: offsets and size from gcc against ctypes."""
    import ctypes
    import subprocess
    fields = [n for n, _ in N.ConvDesc._fields_]
    cname = {"seg": "in"}
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dsvc.h"\nint main(void){\n' +
                   "".join(f'printf("%zu\\n", offsetof(dsvc_conv_desc, {cname.get(f, f)}));\n' for f in fields) +
                   'printf("%zu\\n", sizeof(dsvc_conv_desc)); printf("%zu\\n", sizeof(dsvc_seg)); return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    for f, off in zip(fields, nums):
        assert getattr(N.ConvDesc, f).offset == off, f"dsvc_conv_desc.{f}: ctypes offset {getattr(N.ConvDesc, f).offset} != C {off}"
    assert ctypes.sizeof(N.ConvDesc) == nums[len(fields)] and ctypes.sizeof(N.Seg) == nums[len(fields) + 1]


def test_cpu_tensors_are_refused():
    from deepsvc_amd.modules import Conv2d, torch_warp
    with pytest.raises(N.NativeError):
        Conv2d(3, 8, 3, 1, 1)(torch.zeros(1, 3, 8, 8))
    with pytest.raises(N.NativeError):
        torch_warp(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))


def test_error_convention():
    L = N.lib()
    assert L.dsvc_conv2d(None, None) == -1
    assert b"null descriptor" in L.dsvc_last_error()
    with pytest.raises(N.NativeError):
        ops.pmf_to_quantized_cdf(np.array([0.5, -1.0], dtype=np.float32))
    with pytest.raises(N.NativeError):
        ops.RansDecoder(b"\x00\x01\x02")


@pytest.fixture(scope="module")
def gc():
    return E.gc_tables()


@pytest.mark.parametrize("n", [0, 1, 3, 1000, 60000])
def test_rans_bytes_equal_oracle(gc, n):
    rng = np.random.default_rng(n)
    idx = rng.integers(0, 64, n).astype(np.int32)
    sym = np.round(rng.standard_normal(n) * gc["scale_table"].numpy()[idx] * 1.3).astype(np.int32)
    if n > 10:
        sym[::9] += 123456
        sym[4::9] -= 654321
    t = ops.CdfTables(gc["cdf"], gc["cdf_length"], gc["offset"])
    enc = ops.RansEncoder()
    k = n // 2          # two encode calls accumulate into one stream (BufferedRansEncoder)
    enc.encode_with_indexes(sym[:k], idx[:k], t)
    enc.encode_with_indexes(sym[k:], idx[k:], t)
    data = enc.flush()
    assert data == E.rans_encode(sym, idx, gc["cdf"], gc["cdf_length"], gc["offset"])
    dec = ops.RansDecoder(data)
    out = np.concatenate([dec.decode_stream(idx[:k], t), dec.decode_stream(idx[k:], t)])
    assert np.array_equal(out, sym)
    # the oracle decodes the product's stream too
    assert np.array_equal(E.RansDecoder(data).decode(idx, gc["cdf"], gc["cdf_length"], gc["offset"]), sym)


def test_pmf_to_quantized_cdf_equals_oracle():
    rng = np.random.default_rng(7)
    for n in (2, 5, 64, 777):
        p = (rng.random(n) ** 10).astype(np.float32)
        p /= p.sum()
        assert np.array_equal(ops.pmf_to_quantized_cdf(p), E.pmf_to_quantized_cdf(p))


def test_entropy_model_tables_equal_oracle():
    from deepsvc_amd.entropy_models import EntropyBottleneck, GaussianConditional
    from deepsvc_amd.modules import get_scale_table
    torch.manual_seed(11)
    eb = EntropyBottleneck(8)
    with torch.no_grad():
        eb.quantiles[:, 0, 0] -= torch.rand(8) * 5
        eb.quantiles[:, 0, 2] += torch.rand(8) * 7
        eb._matrix1 += torch.randn_like(eb._matrix1) * 0.1
    assert eb.update() and not eb.update()
    t = E.eb_tables({k: v.detach() for k, v in eb.state_dict().items()}, "")
    assert np.array_equal(eb.quantized_cdf.numpy(), t["cdf"])
    assert np.array_equal(eb.cdf_length.numpy(), t["cdf_length"]) and np.array_equal(eb.offset.numpy(), t["offset"])
    g = GaussianConditional(None)
    assert g.update_scale_table(get_scale_table())
    o = E.gc_tables()
    assert np.array_equal(g.quantized_cdf.numpy(), o["cdf"]) and np.array_equal(g.offset.numpy(), o["offset"])
    assert np.array_equal(g.cdf_length.numpy(), o["cdf_length"])
    assert torch.equal(g.scale_table, o["scale_table"])


def test_deconv_rewrite_is_exact():
    """ConvTranspose2d(5,s2,p2,op1) == PixelShuffle(2)(conv3x3 with the rewritten weights) -- checked with torch-CPU."""
    torch.manual_seed(0)
    w, b = torch.randn(5, 3, 5, 5), torch.randn(3)
    w3, b4 = ops.deconv_as_conv_weights(w, b)
    x = torch.randn(2, 5, 6, 7)
    want = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
    got = F.pixel_shuffle(F.conv2d(x, w3, b4, padding=1), 2)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("compute", [N.F32, N.F16])
def test_weight_packing_layout(compute):
    """[chunk][tap][Np][64 B]; chunks never straddle segments; zero padded."""
    torch.manual_seed(1)
    segs, cout, k = (5, 20), 6, 3
    w = torch.randn(cout, sum(segs), k, k)
    pc = ops.PackedConv.__new__(ops.PackedConv)
    import ctypes
    sa = (ctypes.c_int32 * 2)(*segs)
    nb = N.lib().dsvc_conv_packed_bytes(cout, 2, sa, k, k, compute)
    kce = 16 if compute == N.F32 else 32
    nch = sum(-(-c // kce) for c in segs)
    assert nb == nch * 9 * 32 * 64
    host = np.empty(nb, np.uint8)
    assert N.lib().dsvc_conv_pack_weights(w.contiguous().data_ptr(), cout, 2, sa, k, k, compute, host.ctypes.data) == 0
    if compute == N.F32:
        arr = host.view(np.float32).reshape(nch, 9, 32, 16)
        ref = w.numpy()
    else:
        arr = host.view(np.float16).astype(np.float32).reshape(nch, 9, 32, 32)
        ref = w.to(torch.float16).float().numpy()
    chunk, cbase = 0, 0
    for c in segs:
        for j in range(-(-c // kce)):
            for e in range(kce):
                cl = j * kce + e
                for n in range(32):
                    want = ref[n, cbase + cl].reshape(9) if (n < cout and cl < c) else np.zeros(9, np.float32)
                    assert np.array_equal(arr[chunk, :, n, e], want)
            chunk += 1
        cbase += c


def test_malformed_cdf_tables_are_rejected():
    """CDF tables reach the coder from checkpoint buffers (entropy_models.host_tables): a row that does not start at 0, end at
    65536 or has a zero-frequency symbol is refused when the table is built (dsvc_cdf_build_lut) instead of producing an
    undecodable stream or an out-of-bounds read in the decoder."""
    good = np.array([[0, 10, 40000, 65536, 0], [0, 1, 2, 3, 65536]], dtype=np.int32)
    lens, offs = np.array([4, 5], dtype=np.int32), np.array([-1, -2], dtype=np.int32)
    t = ops.CdfTables(good, lens, offs)
    enc = ops.RansEncoder()
    enc.encode_with_indexes(np.array([0, -1, 1, 5, -7], dtype=np.int32), np.array([0, 0, 1, 1, 0], dtype=np.int32), t)
    data = enc.flush()
    out = ops.RansDecoder(data).decode_stream(np.array([0, 0, 1, 1, 0], dtype=np.int32), t)
    assert list(out) == [0, -1, 1, 5, -7]
    for bad_row, why in (([1, 10, 40000, 65536, 0], "does not start at 0"), ([0, 10, 10, 65536, 0], "zero-frequency symbol"),
                         ([0, 10, 40000, 65535, 0], "does not end at 65536"), ([0, 50000, 40000, 65536, 0], "not monotone")):
        bad = good.copy()
        bad[0] = bad_row
        with pytest.raises(RuntimeError):
            ops.CdfTables(bad, lens, offs)
    with pytest.raises(RuntimeError):
        ops.CdfTables(good, np.array([4, 6], dtype=np.int32), offs)        # cdf_length beyond the row stride
    # a LUT pointer alone buys no trust: only a LUT that dsvc_cdf_build_lut made for THIS table skips the per-call validation
    bad = good.copy()
    bad[0] = [0, 10, 10, 65536, 0]
    forged_lut = np.zeros((2, 256), dtype=np.uint16)
    forged = N.CdfTable(bad.ctypes.data, 2, 5, lens.ctypes.data, offs.ctypes.data, forged_lut.ctypes.data)
    L = N.lib()
    h = ctypes_void = __import__("ctypes").c_void_p()
    assert L.dsvc_rans_encoder_create(__import__("ctypes").byref(h)) == 0
    sym, idx = np.array([0, 1], dtype=np.int32), np.array([0, 1], dtype=np.int32)
    assert L.dsvc_rans_encode_with_indexes(h, sym.ctypes.data, idx.ctypes.data, 2, __import__("ctypes").byref(forged)) < 0, "forged LUT was trusted"
    L.dsvc_rans_encoder_destroy(h)
    # ... and editing a table after its LUT was built revokes the trust
    t2 = ops.CdfTables(good.copy(), lens, offs)
    t2.cdf[0, 2] = 10                                  # now a zero-frequency symbol (also the fingerprinted mid entry)
    with pytest.raises(RuntimeError):
        ops.RansEncoder().encode_with_indexes(sym, idx, t2)


def test_scan_op_ctypes_mirrors_match_the_header(tmp_path):
    """dsvc_scan_op and its union members (dsvc_entropy_scan_step's op list) must lay out exactly like the C structs: sizes and
    field offsets from gcc against the ctypes mirrors in deepsvc_amd/_native.py."""
    import ctypes
    import subprocess
    structs = {"dsvc_scan_gc": N.ScanGc, "dsvc_scan_eb": N.ScanEb, "dsvc_scan_ew": N.ScanEw, "dsvc_scan_call": N.ScanCall, "dsvc_scan_op": N.ScanOp}
    lines = []
    for cname, cls in structs.items():
        for f, _ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
    src = tmp_path / "scan_layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dsvc.h"\nint main(void){\n' + "\n".join(lines) + "\nreturn 0;}\n")
    exe = tmp_path / "scan_layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.rsplit(" ", 1) for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert ctypes.sizeof(cls) == int(got[cname]), f"sizeof({cname}): ctypes {ctypes.sizeof(cls)} != C {got[cname]}"
        for f, _ in cls._fields_:
            assert getattr(cls, f).offset == int(got[f"{cname}.{f}"]), f"{cname}.{f}: ctypes offset {getattr(cls, f).offset} != C {got[cname + '.' + f]}"
    # the union's members all start at the union's offset
    assert N._ScanU.conv.offset == N._ScanU.gc.offset == N._ScanU.eb.offset == N._ScanU.ew.offset == N._ScanU.call.offset == 0
    assert ctypes.sizeof(N._ScanU) == max(ctypes.sizeof(c) for c in (N.ConvDesc, N.ScanGc, N.ScanEb, N.ScanEw, N.ScanCall))
    # every op kind maps to a union member, and the pointer finder sees the pointers of each member
    assert set(N.SCAN_UNION_MEMBER.values()) == {"conv", "gc", "eb", "ew", "call"}
    assert len(N.pointer_offsets(N.ConvDesc)) == 11 and len(N.pointer_offsets(N.ScanCall)) == 6 and len(N.pointer_offsets(N.ScanEw)) == 6


def test_frame_program_packs_temporaries_by_lifetime_and_relocates_pointers():
    """ops.FrameProgram (host logic of dsvc_pframe_segment_run), on CPU tensors -- no launch: temporaries whose lifetimes (first ..
    last op touching them) overlap get disjoint arena ranges, those that do not may share; pointers into externals are patched per
    run, pointers into persistent tensors stay as recorded."""
    import ctypes
    import torch
    from deepsvc_amd import ops

    rec = ops.ScanRecorder()
    mk = lambda n: torch.zeros(n, dtype=torch.uint8)
    ext_in, persistent_w = mk(4096), mk(1024)
    t = [mk(1000), mk(3000), mk(2000), mk(500)]          # lifetimes (ops): t0 [0,1], t1 [1,2], t2 [2,3], t3 [3,3]
    out = ops.Act(torch.zeros((1, 2, 2, 8), dtype=torch.float32))
    rec.temps += t + [out.buf]

    def ew(a, b, o):                                       # an eltwise op: out = f(a, b)
        w = N.ScanEw()
        w.a.ptr, w.b.ptr, w.out = a, b, o
        rec.add(N.SCAN_ELTWISE, "ew", w, [])
    ew(ext_in.data_ptr() + 16, persistent_w.data_ptr(), t[0].data_ptr())
    ew(t[0].data_ptr(), persistent_w.data_ptr() + 8, t[1].data_ptr() + 32)
    ew(t[1].data_ptr(), None, t[2].data_ptr())
    ew(t[2].data_ptr(), t[3].data_ptr(), out.ptr)
    prog = ops.FrameProgram(rec, [ext_in], [out], [])
    arr, base = prog.arr, prog.arena.data_ptr()
    got = lambda i, f: getattr(arr[i].u.ew, f).ptr if f != "out" else arr[i].u.ew.out
    off = {0: got(0, "out") - base, 1: got(1, "out") - 32 - base, 2: got(2, "out") - base, 3: got(3, "b") - base}
    assert got(1, "a") - base == off[0] and got(2, "a") - base == off[1] and got(3, "a") - base == off[2]       # producer / consumer see one address
    size = {0: 1000, 1: 3000, 2: 2000, 3: 500}
    live = {0: (0, 1), 1: (1, 2), 2: (2, 3), 3: (3, 3)}
    for i in range(4):
        assert 0 <= off[i] and off[i] + size[i] <= prog.arena.numel() and off[i] % 512 == 0
        for j in range(i + 1, 4):
            if not (live[i][1] < live[j][0] or live[j][1] < live[i][0]):
                assert off[i] + size[i] <= off[j] or off[j] + size[j] <= off[i], f"temporaries {i} and {j} are live together but overlap"
    assert prog.arena_bytes < sum(-(-s // 512) * 512 for s in size.values()), "lifetime packing reused nothing"
    assert got(0, "b") == persistent_w.data_ptr() and got(1, "b") == persistent_w.data_ptr() + 8                 # persistent: untouched
    # externals: input slot 0 and the fresh output (slot 1) are patched by run(); emulate the patching without a launch
    new_in, new_out = mk(4096), torch.zeros((1, 2, 2, 8), dtype=torch.float32)
    bases = [new_in.data_ptr(), new_out.data_ptr()]
    for cell, slot, delta in prog.ext_rel:
        cell.value = bases[slot] + delta
    assert got(0, "a") == new_in.data_ptr() + 16 and got(3, "out") == new_out.data_ptr()


def test_no_kernel_of_the_built_library_uses_scratch_memory():
    """deepsvc_amd/build.py compiles every .hip file with -Rpass-analysis=kernel-resource-usage and keeps the per-kernel numbers next to the
    objects.  A kernel with scratch use has lost registers -- or a whole accumulator array -- to the stack: parity tests cannot see that,
    the frame rate can (the generic conv's NT >= 3 slab variants once ran 3x slower that way).  The one tolerated entry is listed in
    build.SCRATCH_ALLOWED."""
    from deepsvc_amd import build
    build.build()
    usage = build.resource_usage()
    assert len(usage) > 150, "resource report missing: run `python -m deepsvc_amd.build --force`"
    bad = {k: v["scratch"] for k, v in usage.items() if v["scratch"] > build.SCRATCH_ALLOWED.get(k, 0)}
    assert not bad, bad
    # the conv kernels are written for two workgroups (waves per SIMD) per CU, the two-group 3x3 kernel for one 512-thread workgroup
    for k, v in usage.items():
        if "conv_mfma_kernel" in k:
            assert v["occupancy"] >= 2, (k, v)
        if "conv3x3_pp_kernel" in k:
            assert v["vgprs"] <= 128 or v["occupancy"] >= 1, (k, v)


def test_no_wide_buffer_store_of_the_built_library_has_a_register_scalar_offset():
    """Round 5: `buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen` (tile offset in the scalar-offset REGISTER) followed in the next issue slot
    by a vector instruction that rewrites v[a] -- the compiler inserts the wait state only when the store has NO register offset -- made the
    two-group conv's epilogue store the new register contents whenever the memory pipeline was backed up: decoded 1080p frames differed from run to
    run, while every isolated launch was bit-exact (csrc/conv3x3_pp.hip, PP_EPI_ST_SOFF; two pinned wait states cure it).  No parity test on
    an idle GPU can see that, so the build disassembles every gfx950 code object and refuses the instruction form; this test reads the same
    report on the CPU and checks that the scanner itself still recognises the form."""
    import glob
    import json
    from deepsvc_amd import build
    build.build()
    reports = sorted(glob.glob(os.path.join(os.path.dirname(build.LIB), "obj", "*.stores.json")))
    hip_sources = [s for s in build.SOURCES if s.endswith(".hip")]
    assert len(reports) == len(hip_sources), (reports, hip_sources)
    for r in reports:
        with open(r) as f:
            assert json.load(f)["wide_stores_with_register_soffset"] == 0, r
    rx = build._WIDE_STORE_SOFF
    assert rx.search("\tbuffer_store_dwordx4 v[52:55], v56, s[16:19], s4 offen offset:64// 0000")
    assert rx.search("buffer_store_dwordx3 v[1:3], off, s[0:3], s101")
    assert not rx.search("buffer_store_dwordx4 v[52:55], v56, s[16:19], 0 offen offset:64")
    assert not rx.search("buffer_store_dwordx2 v[52:53], v56, s[16:19], s4 offen")          # 64 bits of data: no hazard
    assert not rx.search("buffer_load_dwordx4 v[52:55], v56, s[16:19], s4 offen")


def test_host_coder_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY section 5 / VERDICT round 3 item 9: the host range coder parses untrusted byte strings and takes its tables from checkpoint
    buffers.  The PRODUCT sources (csrc/rans_host.cpp, csrc/host_misc.cpp, -DDSVC_HOST_ONLY: host functions only) are compiled together
    with tests/native/fuzz_host_coder.cpp under -fsanitize=address,undefined (-fno-sanitize-recover) and run on the CPU: round trips with
    escapes up to the int32 extremes, every truncation of short streams and random cuts of long ones, bit-flipped and all-ones streams,
    LUTs the library did not build, tables edited after their LUT was built, bad row lengths, out-of-range indexes, a LUT registry
    overflow, NaN / negative / zero pmfs, ragged weight packing.  Any sanitizer report aborts the process (non-zero exit)."""
    import shutil
    import subprocess
    cxx = next((c for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")) if c and os.path.exists(c)), None)
    if cxx is None:
        pytest.skip("no clang++ with sanitizer runtimes (g++ 11 has no _Float16 in C++)")
    csrc = os.path.join(ROOT, "deepsvc_amd", "csrc")
    exe = tmp_path / "fuzz_host_coder"
    subprocess.check_call([cxx, "-std=c++17", "-DDSVC_HOST_ONLY", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1",
                           os.path.join(ROOT, "tests", "native", "fuzz_host_coder.cpp"), os.path.join(csrc, "rans_host.cpp"),
                           os.path.join(csrc, "host_misc.cpp"), "-o", str(exe)])
    for seed in (1, 2, 3):
        r = subprocess.run([str(exe), str(seed)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0, f"seed {seed}:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        assert "no sanitizer report" in r.stdout


def test_lut_registry_forgets_and_counts():
    """dsvc_cdf_forget_lut removes a registration (CdfTables does so when it dies), a forgotten LUT is no longer trusted -- and no longer USED:
    the decoder binary-searches instead of indexing rows with foreign LUT entries -- and results do not change."""
    import ctypes
    good = np.array([[0, 10, 40000, 65536, 0], [0, 1, 2, 3, 65536]], dtype=np.int32)
    lens, offs = np.array([4, 5], dtype=np.int32), np.array([-1, -2], dtype=np.int32)
    t = ops.CdfTables(good, lens, offs)
    sym, idx = np.array([0, -1, 1, 5, -7], dtype=np.int32), np.array([0, 0, 1, 1, 0], dtype=np.int32)
    enc = ops.RansEncoder()
    enc.encode_with_indexes(sym, idx, t)
    data = enc.flush()
    L = N.lib()
    assert L.dsvc_cdf_forget_lut(t.lut.ctypes.data) == 1 and L.dsvc_cdf_forget_lut(t.lut.ctypes.data) == 0
    t.lut[:] = 65535                                  # a LUT nobody vouches for any more, full of out-of-range entries
    assert list(ops.RansDecoder(data).decode_stream(idx, t)) == list(sym)
    assert L.dsvc_get_counter(b"lut_registry_overflow") >= 0
