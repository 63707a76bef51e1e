"""Frame-level parity checks shared by the GPU tests and `__graft_entry__.smoke()`.

Everything here compares tensors the caller obtained from the HIP path (through the product API) with tensors from the
checker (CPU oracle or golden vectors produced by the reference); nothing here computes a reference result itself.

The checks are written so that they can FAIL: they look at un-clamped tensors (decoded feature, the reconstruction in front
of the clamp, latents, symbol planes) with relative-L2 / max-abs bounds a few times above what the fp32 HIP path shows,
and `tests/test_gpu_parity.py::test_parity_checks_catch_mutations` proves that a decoder returning zeros, a skipped UNet
and a single zeroed filter tap each turn them red.
"""
import numpy as np
import torch


def rel_l2(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def max_abs(got, want):
    return float((got.detach().double().cpu() - want.detach().double().cpu()).abs().max())


def psnr(a, b):
    mse = float(((a.detach().double().cpu() - b.detach().double().cpu()) ** 2).mean())
    return 999.0 if mse == 0 else -10.0 * np.log10(mse)


# fp32 mode vs the CPU oracle / the reference's golden tensors: both sides evaluate the same fp32 graph, only the summation
# order inside the convolutions differs.  Observed on MI355X (see the printed table of each test): rel-L2 <= 3e-6 everywhere.
FP32_BOUNDS = dict(feature=1e-5, recon=1e-5, warped=1e-5, predict=1e-5, latent=1e-5)
# f16 mode (IEEE binary16 MFMA operands + activation storage, fp32 accumulation, fp32 latents/heads) vs fp32: every stored
# activation carries a 2^-12 relative rounding error.  (bfloat16 storage showed 3.6e-3..9e-3 here and drifted 0.05-0.1 dB in
# PSNR over three chained frames -- outside the north_star tolerance -- which is why the fast mode is binary16.)
F16_BOUNDS = dict(feature=4e-3, recon=4e-3, warped=4e-3, predict=4e-3, latent=4e-3)
# ... over a WHOLE GOP chained on f16 mode's own decoded state (6 / 11 P-frames): the error of frame t-1 is an input of frame t
F16_GOP_BOUNDS = dict(feature=8e-3, recon=8e-3, warped=8e-3, predict=8e-3)


def check_frame(tag, got, want, bounds=FP32_BOUNDS, report=None):
    """got / want: dicts with `feature` [1,64,H,W] and the UN-CLAMPED `recon`, `warped`, `predict` [1,3,H,W].
    Asserts relative-L2 error per tensor; returns the measured numbers."""
    out = {}
    for name in ("feature", "recon", "warped", "predict"):
        g, w = got[name], want[name]
        assert tuple(g.shape) == tuple(w.shape), f"{tag}/{name}: shape {tuple(g.shape)} vs {tuple(w.shape)}"
        assert bool(torch.isfinite(g).all()), f"{tag}/{name}: non-finite values"
        out[name] = rel_l2(g, w)
    if report is not None:
        report.append((tag, out))
    print(f"[parity] {tag}: " + "  ".join(f"{k} rel-L2 {v:.2e}" for k, v in out.items()))
    for name, v in out.items():
        assert v <= bounds[name], f"{tag}/{name}: relative L2 error {v:.3e} > {bounds[name]:.1e}"
    return out


def check_golden(g, key, t, rel=1e-5, stride=4):
    """Tensor `t` against a golden entry written by tests/golden/make_golden.py::compact (sub-sampled data + fp64 sum and
    sum of squares of the full tensor)."""
    data, shape, stats = g[key + "/data"], tuple(g[key + "/shape"]), g[key + "/stats"]
    a = t.detach().float().cpu().numpy()
    assert tuple(a.shape) == shape, f"{key}: shape {a.shape} vs golden {shape}"
    sub = a if a.size <= 16384 else a[..., ::stride, ::stride]
    err = float(np.linalg.norm((sub - data).astype(np.float64)) / max(np.linalg.norm(data.astype(np.float64)), 1e-30))
    assert err <= rel, f"{key}: relative L2 error {err:.3e} vs the reference's golden tensor (> {rel:.1e})"
    s2 = float((a.astype(np.float64) ** 2).sum())
    assert abs(s2 - stats[1]) <= 4 * rel * max(stats[1], 1e-12) + 1e-9, f"{key}: sum of squares {s2} vs golden {stats[1]}"
    return err


def symbol_mismatch(got, want):
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    assert got.shape == want.shape, f"symbol planes: {got.shape} vs {want.shape}"
    return float((got != want).mean())


def product_symbol_planes(codec, h, w, device):
    """(z symbols, y symbols, y indexes) the product's encoder left in its staging vector."""
    ws = codec._plan(1, h, w, torch.device(device))
    st = ws.staging.cpu().numpy()
    return st[:ws.nz], st[ws.nz:ws.nz + ws.ny], st[ws.nz + ws.ny:]


def product_latent(codec, h, w, device):
    """y_hat [1,M,h,w] the last encode/decode of `codec` left in its scan workspace."""
    ws = codec._plan(1, h, w, torch.device(device))
    return ws.y_hat.nchw().float().cpu().clone()


# ------------------------------------------------------------------------------------------------------------
# full-size fixtures (tests/golden/make_golden_full.py): strided samples + block means (cover every pixel) + fp64 sums
# ------------------------------------------------------------------------------------------------------------
def line_crcs(plane):
    """int array [C, h, w] -> uint32 [C, h]: CRC-32 of every line as little-endian int32 (as make_golden_full.py stores them)."""
    import zlib
    a = np.ascontiguousarray(np.asarray(plane).astype("<i4"))
    C, h, _ = a.shape
    return np.asarray([[zlib.crc32(a[c, y].tobytes()) for y in range(h)] for c in range(C)], dtype=np.uint32)


def golden_full_errors(g, key, t):
    """Tensor `t` against a make_golden_full.py entry -> (rel-L2 on the strided samples, rel-L2 on the block means, relative error of
    the sum of squares).  The block means see every element of the tensor."""
    a = t.detach().float().cpu().numpy()
    shape = tuple(int(v) for v in g[key + "/shape"])
    assert tuple(a.shape) == shape, f"{key}: shape {a.shape} vs golden {shape}"
    stride, blk = int(g[key + "/stride"][0]), int(g[key + "/blk"][0])
    rel = lambda x, y: float(np.linalg.norm((x.astype(np.float64) - y.astype(np.float64))) / max(np.linalg.norm(y.astype(np.float64)), 1e-30))
    e_s = rel(a[..., ::stride, ::stride], g[key + "/data"])
    *lead, H, W = a.shape
    bm = a.astype(np.float64).reshape(*lead, H // blk, blk, W // blk, blk).mean(axis=(-3, -1))
    e_b = rel(bm, g[key + "/blockmean"])
    s2 = float((a.astype(np.float64) ** 2).sum())
    e_q = abs(s2 - float(g[key + "/stats"][1])) / max(float(g[key + "/stats"][1]), 1e-30)
    return e_s, e_b, e_q


def check_golden_full(g, key, t, rel, label=None):
    e_s, e_b, e_q = golden_full_errors(g, key, t)
    print(f"[parity] {label or key}: samples rel-L2 {e_s:.2e}, block means rel-L2 {e_b:.2e}, sum of squares {e_q:.2e}")
    assert e_s <= rel, f"{key}: strided samples off the reference's by {e_s:.3e} (> {rel:.1e})"
    assert e_b <= rel, f"{key}: block means off the reference's by {e_b:.3e} (> {rel:.1e})"
    assert e_q <= 4 * rel, f"{key}: sum of squares off the reference's by {e_q:.3e}"
    return e_s


# ------------------------------------------------------------------------------------------------------------------------
# hard decisions at full size: where a conforming fp32 implementation may legitimately differ from the reference
# ------------------------------------------------------------------------------------------------------------------------
def _dilate(mask, r):
    if r <= 0 or not mask.any():
        return mask.copy()
    t = torch.from_numpy(mask[None, None].astype(np.float32))
    return torch.nn.functional.max_pool2d(t, 2 * r + 1, stride=1, padding=r)[0, 0].numpy() > 0


def explain_plane_mismatches(g, pre, sym, idx, slice_depth, max_support, sym_tol=2e-5, idx_tol=2e-5, radius=7, seed=None, param_radius=4):
    """The GPU encoder's symbol / index planes [M, h, w] against the REFERENCE's (full planes + list of fragile decisions stored by
    tests/golden/make_golden_full.py).  The channel-conditional scan rounds `y - mu` and looks sigma up in a 64-level table, slice after
    slice; two fp32 implementations with different summation orders (oneDNN on the CPU, MFMA on the GPU: ~1e-6 relative apart) can
    only disagree where the reference's own decision was within that distance of flipping (a ROOT), and then wherever a root's
    altered y_hat reaches through the 3x3 stacks of the later slices.  The receptive fields are exact, not generous: mu / sigma of a
    slice come from FOUR 3x3 convs over the support slices' y_hat (image_model.py:172-181), so its symbols / indexes can differ only
    within `param_radius` = 4 latent pixels of a support-slice difference; its own y_hat additionally passes THREE 3x3 convs of the
    latent residual prediction (:185-188), so the set of differing y_hat pixels grows by <= `radius` = 4 + 3 per slice level; only the
    first `max_support` slices feed others (:227).  `seed`: latent
    pixels [h, w] whose INPUT already differs (the residual codec codes a picture predicted with the motion codec's output, so the
    motion codec's mismatches seed it).
    Walks the slices in dependency order and classifies every mismatch as root (listed as fragile with margin <= tol), downstream
    of a root, or UNEXPLAINED.  -> dict(symbols, indexes, roots, root_margins, unexplained, lines)"""
    ref_sym, ref_idx = g[pre + "_sym"].astype(np.int32), g[pre + "_idx"].astype(np.int32)
    sym, idx = np.asarray(sym, dtype=np.int32).reshape(ref_sym.shape), np.asarray(idx, dtype=np.int32).reshape(ref_idx.shape)
    M, h, w = ref_sym.shape
    fs, fi = g[pre + "_sym_fragile"], g[pre + "_idx_fragile"]
    frag_s = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) / (1.0 + float(r[4])) for r in fs}       # margin relative to 1 + |y| + |mu|
    frag_i = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) for r in fi}
    contaminated = np.zeros((h, w), dtype=bool) if seed is None else seed.copy()   # latent pixels where a support slice's y_hat (or the input) differs
    roots, unexplained, n_sym, n_idx, checked_by_slice = [], [], 0, 0, []
    for i in range(M // slice_depth):
        c0 = i * slice_depth
        ms, mi = sym[c0:c0 + slice_depth] != ref_sym[c0:c0 + slice_depth], idx[c0:c0 + slice_depth] != ref_idx[c0:c0 + slice_depth]
        n_sym, n_idx = n_sym + int(ms.sum()), n_idx + int(mi.sum())
        both = ms | mi
        checked_by_slice.append(1.0 - float(_dilate(contaminated, param_radius).mean()))      # share of the map held to "identical, or a listed root" in this slice
        if both.any():
            zone = _dilate(contaminated, param_radius)
            for c, y, x in zip(*np.nonzero(both & ~zone[None])):
                key = (c0 + int(c), int(y), int(x))
                m_s = frag_s.get(key) if ms[c, y, x] else None
                m_i = frag_i.get(key) if mi[c, y, x] else None
                if (m_s is not None and m_s <= sym_tol) or (m_i is not None and m_i <= idx_tol):
                    roots.append((key, "symbol" if m_s is not None and m_s <= sym_tol else "index", m_s if m_s is not None else m_i))
                else:
                    unexplained.append((key, bool(ms[c, y, x]), bool(mi[c, y, x]), m_s, m_i))
        if i < max_support:
            contaminated = _dilate(contaminated, radius) | both.any(0)
    lines = int((line_crcs(sym) != line_crcs(ref_sym)).sum()), int((line_crcs(idx) != line_crcs(ref_idx)).sum())
    where = ((sym != ref_sym) | (idx != ref_idx)).any(0)
    # how much of the latent map the walk could still hold to "identical or a listed root" at the LAST level (the power of the check)
    return dict(symbols=n_sym, indexes=n_idx, total=int(sym.size), roots=roots, unexplained=unexplained, lines=lines, where=where,
                strictly_checked=checked_by_slice)


def explain_hyper_mismatches(g, pre, zsym, y_shape, tol=2e-5, reach=8):
    """The GPU encoder's HYPER-latent symbols [N, h/4, w/4] against the reference's (`<pre>z_sym` + `<pre>z_fragile` of
    tests/golden/make_golden_full.py): a differing z symbol must be a rounding decision the reference took within `tol` (relative to
    1 + |z|) of flipping -- a z ROOT.  Its z_hat feeds both hyper-synthesis nets (two transposed 5x5 stride-2 convs and a 3x3 conv,
    image_model.py:384-398: a z pixel reaches +-7 latent pixels around its 4x4 footprint), so it seeds y-plane differences there.
    -> dict(roots [(key, margin)], unexplained [...], seed bool [h, w] for explain_plane_mismatches)"""
    ref = g[pre + "z_sym"].astype(np.int32)
    zsym = np.asarray(zsym, dtype=np.int32).reshape(ref.shape)
    frag = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) / (1.0 + float(r[4])) for r in g[pre + "z_fragile"]}
    h, w = y_shape
    seed = np.zeros((h, w), dtype=bool)
    roots, unexplained = [], []
    for c, y, x in zip(*np.nonzero(zsym != ref)):
        key = (int(c), int(y), int(x))
        m = frag.get(key)
        (roots if m is not None and m <= tol else unexplained).append((key, m))
        seed[max(4 * y - reach, 0):4 * y + 4 + reach, max(4 * x - reach, 0):4 * x + 4 + reach] = True
    return dict(roots=roots, unexplained=unexplained, seed=seed, differing=int((zsym != ref).sum()), total=int(ref.size))


def first_decoder_index_mismatch(g, pre, idx_taps, groups, slice_depth, h, w):
    """The index planes the GPU decoder used (level by level: ChannelSplitICIP2020ResB.debug_idx_taps) while decoding the REFERENCE's
    string, against the reference's index planes, in bitstream order.  -> None when every CDF lookup matched (the decode is then the
    reference's decode, symbol for symbol), else ((channel, y, x), reference margin of that decision or None, symbols decoded correctly
    before it): the range decoder is desynchronised from there on -- one wrong CDF shifts its state for the rest of the stream."""
    ref_idx = g[pre + "_idx"].astype(np.int32)
    frag = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) for r in g[pre + "_idx_fragile"]}
    done = 0
    for level, group in zip(idx_taps, groups):
        got = level.numpy().reshape(len(group) * slice_depth, h, w)
        want = np.concatenate([ref_idx[i * slice_depth:(i + 1) * slice_depth] for i in group], 0)
        bad = np.nonzero((got != want).reshape(-1))[0]
        if bad.size:
            c, y, x = np.unravel_index(int(bad[0]), got.shape)
            ch = group[int(c) // slice_depth] * slice_depth + int(c) % slice_depth
            return (ch, int(y), int(x)), frag.get((ch, int(y), int(x))), done + int(bad[0])
        done += got.size
    return None
