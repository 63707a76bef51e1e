"""GOP loop, metrics and multi-GPU sharding around the P-frame codec -- the build's counterpart of the part of
the reference's evaluation script that drives the hot path (test_video.py:290-369, 353-366).

One process per GPU; the unit of work is a (sequence, GOP): P-frame t needs `ref_frame` and `feature` of t-1
(test_video.py:368-369) and every key frame resets both (:296-297,332), so GOPs are independent and shard across
ranks with NO data-path collective.  The only communication is one gather of per-frame metric rows at the end
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
"""
import math
import time

import torch
import torch.nn.functional as F

from .modules import crop, pad

METRIC_COLUMNS = ("unit", "frame", "bpp_mv", "bpp_res", "psnr", "psnr_mc", "psnr_warp", "t_enc", "t_dec", "msssim", "msssim_mc", "msssim_warp")
# every key of the reference's per-run JSON (test_video.py:442-460), in its order
REFERENCE_JSON_KEYS = ("psnr", "bpp", "msssim", "ipsnr", "ibpp", "imsssim", "ppsnr", "pbpp", "porposed_pbpp_wosm", "pmsssim", "mcpsnr", "warppsnr",
                       "mvbpp", "resbpp", "mcmsssim", "warmsssim", "ienc", "idec", "pent", "pdec", "ent", "dec", "smbpp", "sment", "smdec", "bpp2l")


def shard_units(n_units, rank, world_size):
    """Static round-robin assignment of GOP indices to ranks."""
    return list(range(rank, n_units, world_size))


def strings_bytes(enc):
    """test_video.py:364-365."""
    return sum(len(s[0]) for s in enc["strings"])


def _psnr(a, b):
    mse = F.mse_loss(a, b).item()
    return 99.0 if mse <= 0 else -10.0 * math.log10(mse)


def _msssim(a, b):
    from .metrics import ms_ssim
    return float(ms_ssim(a.contiguous(), b.contiguous(), 1.0))


def process(frame255, inverse=False):
    """test_video.py:176-190 `Process`: per-channel (v - mean) / std of a 0..255 RGB frame, channels returned in reversed order
    (the semantic layer's input normalisation)."""
    mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    ch = [frame255[:, i:i + 1] for i in range(3)]
    out = [c * s_ + m if inverse else (c - m) / s_ for c, m, s_ in zip(ch, mean, std)]
    return torch.cat([out[2], out[1], out[0]], 1)


def code_gop(model, frames, sm_feas, unit=0, sync=None, metrics=True, i_model=None, msssim=False, overlap_coding=False,
             sm_layer=None, supp=None, sm_log=None):
    """Encode+decode one GOP.  frames[1:] are P-frames; frames[0] is the key frame: taken as already decoded when
    `i_model` is None (BASELINE.json configs 1-2), else coded with the I-frame codec `i_model` (`ICIP2020ResB`,
    test_video.py:296-312: compress -> decompress, the clamped reconstruction becomes the first reference) and reported
    as a row with frame index 0 whose `bpp_res` column holds the I-frame bpp.  All frames are [1,3,h,w] tensors on the
    model's device, un-padded.  Returns one metric row per coded frame.  `msssim=True` fills the last column with MS-SSIM of
    the cropped reconstruction (test_video.py:311,359-363; GPU kernel, frames must exceed 160 px on the smaller side).

    Timing: `DeepSVC.compress` returns while the host range coder is still writing the strings (PendingStrings).  By default
    the encode clock `t_enc` stops only after both codings have been JOINED, so t_enc / t_dec mean what the reference's
    timers mean (test_video.py:340-351: strings exist when compress returns).  `overlap_coding=True` is the throughput mode:
    the strings are joined by whoever reads them first (the decoder), which overlaps the residual range coding with the
    decoder's motion path -- t_enc then excludes, and t_dec includes, that coding time; t_enc + t_dec is the same either way.

    `sm_layer` (a `semantic_layer.StudentLayer1`) makes the loop self-contained the way test_video.py:336-338 is: per P-frame
    `sm, y = sm_layer(process(cur * 255), supp, mean=True, encode=True)` produces `sm_fea` (instead of `sm_feas[t]`), and the
    support stack advances `supp = cat([supp, sm])[1:]`; `supp` is the initial support ([n,256,H/4,W/4] on the padded frame: in the
    reference the teacher network's feature of the key frame, out of scope here).  Its (bpp, enct, dect) per frame are appended
    to `sm_log` -- the `sm_rows` argument of `reference_results`."""
    sync = sync or (lambda: None)
    h, w = frames[0].shape[-2:]
    rows = []
    if i_model is None:
        ref = pad(frames[0], 64)
    else:
        key = pad(frames[0], 64)
        sync(); t0 = time.perf_counter()
        i_enc = i_model.compress(key)
        sync(); t1 = time.perf_counter()
        ref = i_model.decompress(i_enc["strings"], i_enc["shape"])["x_hat"]
        sync(); t2 = time.perf_counter()
        i_psnr = _psnr(crop(ref, (h, w)), frames[0]) if metrics else 0.0
        rows.append([float(unit), 0.0, 0.0, strings_bytes(i_enc) * 8.0 / (h * w), i_psnr, 0.0, 0.0, t1 - t0, t2 - t1,
                     _msssim(crop(ref, (h, w)), frames[0]) if msssim else 0.0, 0.0, 0.0])
    feature = None
    for t in range(1, len(frames)):
        cur = pad(frames[t], 64)
        if sm_layer is not None:
            sm, _ = sm_layer(process(cur * 255.0), supp, mean=True, encode=True)            # test_video.py:336
            supp = torch.cat([supp, sm], 0)[1:]                                             # :338
            if sm_log is not None:
                sm_log.append((float(sm_layer.bpp_loss), sm_layer.enct, sm_layer.dect))     # as the reference takes it (test_video.py:337,366): per PADDED pixel
        else:
            sm = sm_feas[t]
        sync(); t0 = time.perf_counter()
        mv_enc, res_enc = model.compress(ref, cur, sm, feature)
        if not overlap_coding:
            for enc in (mv_enc, res_enc):
                if hasattr(enc, "result"):
                    enc.result()
        sync(); t1 = time.perf_counter()
        feature, recon, warped, predict = model.decompress(ref, mv_enc, res_enc, sm, feature)
        sync(); t2 = time.perf_counter()
        row = [float(unit), float(t), strings_bytes(mv_enc) * 8.0 / (h * w), strings_bytes(res_enc) * 8.0 / (h * w)]
        if metrics:
            org = frames[t]
            row += [_psnr(crop(recon, (h, w)), org), _psnr(crop(predict, (h, w)), org), _psnr(crop(warped, (h, w)), org)]
        else:
            row += [0.0, 0.0, 0.0]
        ms = [_msssim(crop(x, (h, w)), frames[t]) for x in (recon, predict, warped)] if msssim else [0.0, 0.0, 0.0]
        rows.append(row + [t1 - t0, t2 - t1] + ms)
        ref = recon
    return rows


def code_gops_in_lanes(models, gops, device, metrics=True, i_models=None, overlap_coding=True):
    """Throughput mode: code the independent GOPs `gops` = [(frames, sm_feas), ...] on len(models) concurrent LANES of one
    GPU.  A lane is one model instance (own weights, scan workspaces and HIP graphs), one stream and one host thread; GOP u
    goes to lane u % L.  One GOP alone leaves the GPU idle ~20 % of a frame while the host range-decodes each scan level
    (sequential by bitstream format); other lanes' kernels fill those gaps.  Results are identical to
    coding the GOPs one after the other with code_gop (every kernel is deterministic); rows come back ordered by GOP.

    Weight packing and HIP-graph capture need a quiescent device, so each lane is first warmed up alone on the P-frames of its
    first GOP (results discarded) with DeepSVC.warm_up, which returns when that instance's graphs and frame programs are in place."""
    import threading
    L = len(models)
    assert L >= 1 and (i_models is None or len(i_models) == L)
    for m in models:                         # several GOPs in flight: ONE stream per lane (video_model.DeepSVC.use_side_stream / single_stream)
        m.use_side_stream = L == 1
        if L > 1:
            m.single_stream = False
    streams = [torch.cuda.Stream(device) for _ in range(L)]
    mine = [[u for u in range(len(gops)) if u % L == ln] for ln in range(L)]

    for ln in range(L):                      # one lane at a time, each to completion (DeepSVC.warm_up: deterministic, no cross-lane retries)
        if not mine[ln]:
            continue
        # warm up on the lane's first GOP that HAS P-frames (a lone key frame -- e.g. the tail GOP of a sequence -- has nothing to code);
        # a lane that holds only such GOPs borrows the P-frames of any other GOP: the recordings depend on the shape, not on the content
        own = [u for u in mine[ln] if len(gops[u][0]) >= 2]
        anyone = own or [u for u in range(len(gops)) if len(gops[u][0]) >= 2]
        if not anyone:
            continue                         # nothing but key frames anywhere: no P-frame will be coded, nothing to warm up
        frames, sms = gops[anyone[0]]
        with torch.cuda.stream(streams[ln]):
            ref = pad(frames[0], 64)
            if i_models is not None:
                enc = i_models[ln].compress(ref)
                ref = i_models[ln].decompress(enc["strings"], enc["shape"])["x_hat"]
            models[ln].warm_up(ref, [pad(f, 64) for f in frames[1:]], list(sms[1:]))
    out, errors = {}, []

    def lane(ln):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[ln]):
                for u in mine[ln]:
                    frames, sms = gops[u]
                    out[u] = code_gop(models[ln], frames, sms, unit=u, sync=streams[ln].synchronize, metrics=metrics,
                                      i_model=None if i_models is None else i_models[ln], overlap_coding=overlap_coding)
                streams[ln].synchronize()
        except BaseException as e:          # surfaced in the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=lane, args=(ln,)) for ln in range(L)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return [row for u in range(len(gops)) for row in out[u]]


def gather_rows(rows, device, world_size):
    """All ranks' metric rows -> one [n, len(METRIC_COLUMNS)] fp64 CPU tensor on every rank."""
    local = torch.tensor(rows, dtype=torch.float64, device=device).reshape(-1, len(METRIC_COLUMNS))
    import torch.distributed as dist
    if world_size == 1 and not (dist.is_available() and dist.is_initialized()):
        return local.cpu()            # (a one-rank process group still takes the collective path: bench.py DSVC_BENCH_FORCE_PG)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world_size)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=device))
    n_max = int(max(int(c) for c in counts))
    padded = torch.zeros((n_max, local.shape[1]), dtype=torch.float64, device=device)
    padded[: local.shape[0]] = local
    parts = [torch.zeros_like(padded) for _ in range(world_size)]
    dist.all_gather(parts, padded)
    return torch.cat([p[: int(c)] for p, c in zip(parts, counts)], 0).cpu()


def summarise(table):
    cols = {n: table[:, i] for i, n in enumerate(METRIC_COLUMNS)}
    n = table.shape[0]
    return {"frames": int(n), "bpp": float((cols["bpp_mv"] + cols["bpp_res"]).mean()) if n else 0.0,
            "mvbpp": float(cols["bpp_mv"].mean()) if n else 0.0, "resbpp": float(cols["bpp_res"].mean()) if n else 0.0,
            "ppsnr": float(cols["psnr"].mean()) if n else 0.0, "mcpsnr": float(cols["psnr_mc"].mean()) if n else 0.0,
            "warppsnr": float(cols["psnr_warp"].mean()) if n else 0.0,
            "pent": float(cols["t_enc"].mean()) if n else 0.0, "pdec": float(cols["t_dec"].mean()) if n else 0.0,
            "pmsssim": float(cols["msssim"].mean()) if n else 0.0}


def reference_results(table, sm_rows=None):
    """The `results` dict of the reference's per-run JSON (test_video.py:353-460): all 26 keys, each a list
    with one entry per SEQUENCE (= `unit` of the metric rows), computed exactly as the reference averages them:
      * I-frame rows (frame index 0 with bpp_mv == 0 and a key-frame bpp in `bpp_res`) feed ipsnr / ibpp / imsssim / ienc / idec;
      * P-frame rows feed ppsnr / pbpp / pmsssim / mcpsnr / warppsnr / mvbpp / resbpp / mcmsssim / warmsssim / pent / pdec;
      * psnr / bpp / msssim / ent / dec / porposed_pbpp_wosm / bpp2l average over ALL frames of the sequence (:370-391);
      * the semantic layer's rate and time (`smbpp`, `sment`, `smdec`; test_video.py:336-337,386-391) come from `sm_rows`
        = {unit: [(sm_bpp, enct, dect) per P-frame]} when a semantic-layer codec produced `sm_fea`, else they are 0 --
        p_bpp = mv + res + sm (:366), bpp2l = mv + sm for P-frames and the I-frame bpp for I-frames (:318,384), and the
        reference ADDS the semantic layer's times to pent / pdec / ent / dec (:388-391)."""
    import numpy as np
    cols = {n: i for i, n in enumerate(METRIC_COLUMNS)}
    t = table.numpy() if hasattr(table, "numpy") else np.asarray(table)
    out = {k: [] for k in REFERENCE_JSON_KEYS}
    avg = lambda v: float(np.average(v)) if len(v) else 0.0      # no I-frame rows (key frame given): 0, so the file stays strict JSON
    for unit in sorted(set(t[:, cols["unit"]].tolist())):
        rows = t[t[:, cols["unit"]] == unit]
        rows = rows[np.argsort(rows[:, cols["frame"]], kind="stable")]
        is_i = (rows[:, cols["frame"]] == 0)
        I, P = rows[is_i], rows[~is_i]
        sm = np.asarray(sm_rows[unit], dtype=np.float64).reshape(-1, 3) if sm_rows and unit in sm_rows else np.zeros((len(P), 3))
        assert len(sm) == len(P), "one (sm_bpp, enct, dect) triple per P-frame"
        g = lambda R, n: R[:, cols[n]]
        i_bpp = g(I, "bpp_res")
        p_wosm = g(P, "bpp_mv") + g(P, "bpp_res")
        p_bpp = p_wosm + sm[:, 0]
        p_enc, p_dec = g(P, "t_enc") + sm[:, 1], g(P, "t_dec") + sm[:, 2]
        out["psnr"].append(avg(np.concatenate([g(I, "psnr"), g(P, "psnr")])))
        out["bpp"].append(avg(np.concatenate([i_bpp, p_bpp])))
        out["msssim"].append(avg(np.concatenate([g(I, "msssim"), g(P, "msssim")])))
        out["ipsnr"].append(avg(g(I, "psnr"))); out["ibpp"].append(avg(i_bpp)); out["imsssim"].append(avg(g(I, "msssim")))
        out["ppsnr"].append(avg(g(P, "psnr"))); out["pbpp"].append(avg(p_bpp))
        out["porposed_pbpp_wosm"].append(avg(np.concatenate([i_bpp, p_wosm])))
        out["pmsssim"].append(avg(g(P, "msssim")))
        out["mcpsnr"].append(avg(g(P, "psnr_mc"))); out["warppsnr"].append(avg(g(P, "psnr_warp")))
        out["mvbpp"].append(avg(g(P, "bpp_mv"))); out["resbpp"].append(avg(g(P, "bpp_res")))
        out["mcmsssim"].append(avg(g(P, "msssim_mc"))); out["warmsssim"].append(avg(g(P, "msssim_warp")))
        out["ienc"].append(avg(g(I, "t_enc"))); out["idec"].append(avg(g(I, "t_dec")))
        out["pent"].append(avg(p_enc)); out["pdec"].append(avg(p_dec))
        out["ent"].append(avg(np.concatenate([g(I, "t_enc"), p_enc]))); out["dec"].append(avg(np.concatenate([g(I, "t_dec"), p_dec])))
        out["smbpp"].append(avg(sm[:, 0])); out["sment"].append(avg(sm[:, 1])); out["smdec"].append(avg(sm[:, 2]))
        out["bpp2l"].append(avg(np.concatenate([i_bpp, g(P, "bpp_mv") + sm[:, 0]])))
    return out


def write_reference_json(path, table, name, sm_rows=None):
    """test_video.py:452-460: {"name", "description": "Inference (ans)", "results": {...}} with indent 2."""
    import json
    output = {"name": name, "description": "Inference (ans)", "results": reference_results(table, sm_rows)}
    with open(path, "w", encoding="utf-8") as f:
        json.dump(output, f, indent=2)
    return output
