#!/usr/bin/env python3
"""bench.py -- encode+decode frames/s of the DeepSVC P-frame hot path at 1080p (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f16|fp32] [--size HxW] [--config 2|3] [--lanes L]

`--gpus N` with N > 1 launches N ranks itself (one fresh child process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, before
the parent touches the GPU; the parent only relays rank 0's line); under `python -m torch.distributed.run ... bench.py --gpus N`
the ranks are already there and WORLD_SIZE must equal N.

A "step" = one frame on every GOP lane of a rank: `DeepSVC.compress` + `DeepSVC.decompress` on a 1920x1080 frame (padded to
1088x1920), frames chained inside GOPs of 7 (1 given key frame + 6 P-frames, BASELINE.json configs[1]); `--config 3`: GOP = 12
whose key frame is coded by the I-frame codec `ICIP2020ResB` (configs[2]).  Synthetic septuplets and the W0 weight sets
(deepsvc_amd/synth.py) are resident in HBM before the timed region.  Each rank codes its own GOPs (weak scaling, no data-path
collective); one metrics gather at the end.

Prints ONE JSON line on rank 0 (contract in the task description): value = frames of all ranks / max-over-ranks time of the
headline leg (f16 MFMA mode, `--lanes` GOP pipelines in flight per GPU: default 6, ONE stream per lane since round 5 -- HIP maps streams onto
4 hardware queues and a second stream per lane cost 6 % at six lanes).  Further legs of the same run, each barrier-bracketed
and timed the same way, are reported inside the line:
  "lanes1"      the same mode with ONE GOP in flight (strictly sequential frames): fps, and the source of `roofline` -- HIP events
                around every launch of the dominant conv shape in that region are the kernel's own durations (with several lanes
                in flight a launch time-shares the GPU with the other lanes' kernels; that figure is kept as `roofline.timeshared_*`);
  "parity_mode" fp32 mode (fp32 storage, fp32 MFMA): the reference's arithmetic; bitstreams byte-identical to the reference's
                at the golden sizes (tests/test_gpu_parity.py); fps + its own roofline against the 157.3 TFLOP/s fp32 MFMA peak;
  "f16_vs_f32" rate / PSNR / un-clamped tensor deltas of f16 mode against fp32 mode on the same frames, measured in this run;
  "cpu_baseline" (+ "cpu_baseline_1thread") the CPU oracle timed on this host (N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "f16": 2500.0}      # MI355X_MICROARCH.md: dense f32 / f16 MFMA peaks
HBM_PEAK = 8.0e12                                    # B/s
# conv FLOP per padded pixel, reference structure (incl. its encoder-side recomputation)
FLOP_PER_PX = {2: 8.571e6,      # GOP = 7: mean over the 6 P-frames (first P 8.453e6, steady-state 8.594e6)
               3: 8.076e6}      # GOP = 12 with the intra-coded key frame: mean over all 12 frames
GOP_OF = {2: 7, 3: 12}
PMC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_dominant_kernel_pmc.json") for r in (5, 4, 3, 2)) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r05_dominant_kernel_pmc.json"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=6, help="independent GOP pipelines in flight per GPU (1 = strictly sequential frames)")
    ap.add_argument("--precision", default=os.environ.get("DSVC_PRECISION", "f16"), choices=["fp32", "f16"])
    ap.add_argument("--size", default="1080x1920")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3], help="BASELINE.json configs[1] (GOP=7, key frame given) or configs[2] (GOP=12, I-frame codec)")
    ap.add_argument("--coder", default="rans", choices=["rans", "rans_ms"],
                    help="range coder of the y strings: rans = the reference's format (default, the headline); rans_ms = the GPU-decodable multi-stream "
                         "coder, a LABELLED non-reference format (+3-4 %% bytes, no host round trips in the decode chain)")
    ap.add_argument("--one-lane-mv-stream", type=int, default=int(os.environ.get("DSVC_BENCH_ONE_LANE_MV", "0")),
                    help="1: the one-GOP-in-flight legs run the motion decode chain on its own stream (DeepSVC.single_stream); never the multi-lane legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="headline leg only (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=256)
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--no-cpu-full-frame", action="store_true", help="skip the one full-size oracle P-frame of the cpu_baseline leg (~15-40 s of CPU)")
    ap.add_argument("--switch-interval", type=float, default=float(os.environ.get("DSVC_SWITCH_INTERVAL", "0.005")),
                    help="sys.setswitchinterval for the lane / coder threads (seconds)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Spawn args.gpus fresh child processes (this interpreter, this script, same flags), one per GPU, before this process makes
    any GPU call; relay rank 0's JSON line; exit non-zero if any rank fails.  Never re-execs a process that touched the GPU."""
    import torch                                   # device_count() does not initialise the GPU on this image
    n = args.gpus
    rehearsal = bool(os.environ.get("DSVC_BENCH_REHEARSAL"))
    have = torch.cuda.device_count()
    if have < n and not rehearsal:
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (DSVC_BENCH_REHEARSAL=1 runs the ranks on one GPU over gloo)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))       # only rank 0 may write the line to stdout
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out.decode().splitlines():          # stdout carries exactly the JSON line; anything else a library printed goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if any(codes):
        sys.exit(f"bench.py: rank exit codes {codes}")


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle on the host cores
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sample_hw, n_frames, threads, H_pad, W_pad):
    """The oracle (CPU restatement of the reference path, oracle/) timed on this host: `n_frames` chained P-frames
    (compress + decompress each; state chaining of test_video.py:342-369) at sample_hw = (h, w) with `threads` torch threads
    (1 = the reference harness's own setting, test_video.py:16); scaled to the padded frame by the pixel ratio when the sample
    is smaller (the cost is per-pixel: SURVEY section 8d).  Baseline only -- never the measured path."""
    import torch
    from deepsvc_amd import synth
    from oracle.codec import DeepSVCOracle
    sh, sw = sample_hw
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        orc = DeepSVCOracle(synth.make_model("W0").state_dict())
        orc.update()
        fr = synth.make_frames(sh, sw, n_frames + 1)
        ref, feat, per, split = fr[0], None, [], []
        for t in range(1, n_frames + 1):
            sm = synth.make_sm_fea(sh, sw, t)
            t0 = time.perf_counter()
            mv, res = orc.compress(ref, fr[t], sm, feat)
            t1 = time.perf_counter()
            feat, ref, _, _ = orc.decompress(ref, mv, res, sm, feat)
            per.append(time.perf_counter() - t0)
            split.append((t1 - t0, per[-1] - (t1 - t0)))
    finally:
        torch.set_num_threads(prev)
    t_tot = sum(per)
    scale = (sh * sw) / float(H_pad * W_pad)
    what = "scaled by pixel ratio to" if scale != 1.0 else "the full padded frame"
    return {"value": n_frames / t_tot * scale, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n_frames} chained P-frame(s) (compress+decompress) at {sh}x{sw}, {t_tot:.1f} s of CPU "
                      f"(per frame min {min(per):.2f} / max {max(per):.2f} s; encode {sum(a for a, _ in split):.1f} s, decode {sum(b for _, b in split):.1f} s), "
                      f"{what} {H_pad}x{W_pad}; oracle = PyTorch-CPU restatement of the reference, pinned byte for byte by the reference's "
                      f"golden vectors (incl. {H_pad}x{W_pad} when that is 1088x1920)"}


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # N ranks share one host: each pins itself to ITS share of the cores this job may use (local ranks take consecutive slices) before
    # any worker thread exists -- lane threads, the range-coder / decode pools and torch's own pool are created later and inherit the
    # mask -- so that 8 ranks x (lane threads + coder threads) do not migrate over each other's cores.  DSVC_BENCH_NO_PIN=1 turns it off.
    host = {"cores_visible": None, "cores_pinned": None}
    if hasattr(os, "sched_getaffinity"):
        allowed = sorted(os.sched_getaffinity(0))
        host["cores_visible"] = len(allowed)
        if world > 1 and not os.environ.get("DSVC_BENCH_NO_PIN") and len(allowed) >= world:
            lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            lr = local_rank % lw
            share = allowed[lr * len(allowed) // lw:(lr + 1) * len(allowed) // lw]
            os.sched_setaffinity(0, share)
            host["cores_pinned"] = len(share)
    import torch
    sys.setswitchinterval(args.switch_interval)
    ranks_seen = 1
    # DSVC_BENCH_FORCE_PG=1: build the process group even for ONE rank, so that init_process_group(nccl, device_id), barrier,
    # all_reduce(MAX) and the all_gather of the metric rows execute on RCCL on a 1-GPU box (tests/test_gpu_dropin.py)
    use_pg = world > 1 or bool(os.environ.get("DSVC_BENCH_FORCE_PG"))
    if use_pg:
        import torch.distributed as dist
        if os.environ.get("DSVC_BENCH_REHEARSAL"):      # N ranks sharing the ONE GPU of a development box: gloo, all on cuda:0
            dist.init_process_group(backend="gloo")        # (exercises the rank / barrier / gather plumbing, not RCCL)
            local_rank = 0
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        ranks_seen = dist.get_world_size()
        assert ranks_seen == args.gpus, f"process group has {ranks_seen} ranks, --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:           # keep each rank's torch CPU thread pool to its share of the cores (the mask set above, or an even split)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 16)
        torch.set_num_threads(max(1, cores if host["cores_pinned"] else cores // world))

    from deepsvc_amd import harness, ops, synth
    from deepsvc_amd.modules import pad
    h, w = (int(v) for v in args.size.lower().split("x"))
    H_pad, W_pad = -(-h // 64) * 64, -(-w // 64) * 64
    GOP = GOP_OF[args.config]
    intra = args.config == 3
    conv_flop = 2.0 * H_pad * W_pad * 64 * 64 * 9

    def barrier():
        if use_pg:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if use_pg:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    # L independent GOP pipelines ("lanes") per GPU: GOPs are independent units of work (harness.py), and one GOP alone
    # leaves the GPU idle ~20 % of a frame while the host range-decodes each scan level (sequential by bitstream format).
    # Each lane = its own model instance (weights, workspaces, HIP graphs), stream and thread, coding its own GOPs; a
    # step codes one frame on EVERY lane (a batch of L frames), so the line reports L*K frames per rank.
    def make_lane(ln, precision):
        ops.set_precision(precision)
        m = synth.make_model("W0")
        m.update(force=True)
        m.set_coder(args.coder)
        im = None
        if intra:
            im = synth.make_icodec("W0")
            im.update(force=True)
            im.coder = args.coder
            im = im.to(dev)
        seed = 1234 + rank * 16 + ln
        return {"model": m.to(dev), "imodel": im, "stream": torch.cuda.Stream(dev), "t": 0, "ref": None, "feature": None, "rows": [],
                "id": ln, "frames": [f.to(dev) for f in synth.make_frames(h, w, GOP, seed=seed)],
                "sm": [None] + [synth.make_sm_fea(H_pad, W_pad, t, seed=seed).to(dev) for t in range(1, GOP)]}

    def step(L, record):
        """One frame of lane L's GOP chain.  config 2: P-frames 1..6 (key frame given); config 3: frame 0 is coded by the
        I-frame codec and counts as a step, then P-frames 1..11."""
        model, frames, sm_feas = L["model"], L["frames"], L["sm"]
        per_gop = GOP if intra else GOP - 1
        t = L["t"] % per_gop + (0 if intra else 1)
        if t == 0:
            key = pad(frames[0], 64)
            enc = L["imodel"].compress(key)
            recon = L["imodel"].decompress(enc["strings"], enc["shape"])["x_hat"]
            if record:
                keep = L["t"] < 2 * per_gop                     # (see the P-frame branch)
                L["rows"].append([float(rank * 1000 + L["id"] * 100 + L["t"] // per_gop), 0.0, 0.0,
                                  harness.strings_bytes(enc) * 8.0 / (h * w), recon if keep else None, key if keep else None])
            L["ref"], L["feature"] = recon, None
            L["t"] += 1
            return
        if t == 1 and not intra:
            L["ref"], L["feature"] = pad(frames[0], 64), None
        cur = pad(frames[t], 64)
        mv_enc, res_enc = model.compress(L["ref"], cur, sm_feas[t], L["feature"])
        feature, recon, warped, predict = model.decompress(L["ref"], mv_enc, res_enc, sm_feas[t], L["feature"])
        if record:
            # decoded / original frames are kept for the PSNR column (computed outside the timed region) for the lane's first two GOPs only:
            # every later GOP of a lane codes the same frames to the same bits (deterministic), and 50 MB per frame would otherwise pile up
            # (+0.19 GB per step at 4 lanes: a 180-step run held 54 GB)
            keep = L["t"] < 2 * per_gop
            L["rows"].append([float(rank * 1000 + L["id"] * 100 + L["t"] // per_gop), float(t),
                              harness.strings_bytes(mv_enc) * 8.0 / (h * w), harness.strings_bytes(res_enc) * 8.0 / (h * w),
                              recon if keep else None, cur if keep else None])
        L["ref"], L["feature"] = recon, feature
        L["t"] += 1

    def run_lane(L, n, record):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(L["stream"]):
            for _ in range(n):
                step(L, record)
            L["stream"].synchronize()

    def warm(lanes, n):
        """One lane at a time, each to completion: DeepSVC.warm_up codes chained P-frames of the lane's GOP until that instance's scan graphs
        and frame programs (both variants) are in place -- everything a recording depends on belongs to the instance, so there is nothing to
        retry across lanes.  `n` (--warmup) further untimed steps per lane follow, one lane at a time."""
        for L in lanes:
            L["model"].use_side_stream = len(lanes) == 1
        for L in lanes:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(L["stream"]):
                if intra:
                    key = pad(L["frames"][0], 64)
                    enc = L["imodel"].compress(key)
                    ref = L["imodel"].decompress(enc["strings"], enc["shape"])["x_hat"]
                else:
                    ref = pad(L["frames"][0], 64)
                took = L["model"].warm_up(ref, [pad(f, 64) for f in L["frames"][1:]], L["sm"][1:])
                L["stream"].synchronize()
            L["warm_frames"] = took
        for _ in range(max(n, 0)):
            for L in lanes:
                run_lane(L, 1, False)
        assert all(L["model"].graphs_ready(1, H_pad, W_pad, dev) for L in lanes)
        for L in lanes:                      # every timed region starts at a GOP boundary
            L["t"] = 0

    def timed(lanes, n):
        """Barrier + synchronize on both sides; every launch of the dominant conv shape inside is bracketed by HIP events on
        its launch stream.  -> (max-over-ranks seconds, [ms per dominant launch on this rank], {residual operands: launches})"""
        for L in lanes:                      # GOP lanes: one stream per lane; one GOP in flight: residual chain on its side stream
            L["model"].single_stream = bool(args.one_lane_mv_stream) and len(lanes) == 1
            L["model"].use_side_stream = len(lanes) == 1
        ops.profile_begin(3, 1, 64, 64, H_pad, W_pad)        # HIP events around every launch of the dominant conv shape, recorded in the C library
        barrier()
        t0 = time.perf_counter()
        if len(lanes) == 1:
            run_lane(lanes[0], n, True)
        else:
            import threading
            threads = [threading.Thread(target=run_lane, args=(L, n, True)) for L in lanes]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
        barrier()
        elapsed = time.perf_counter() - t0
        durs, nres = ops.profile_end()
        hist = {k: nres.count(k) for k in (0, 1, 2)}
        return max_over_ranks(elapsed), durs, hist

    def pmc_traffic(section, hist):
        """HBM bytes per launch from the committed rocprofv3 --pmc passes (they cannot run in-process), weighted by THIS region's
        launch mix: the PMC file holds bytes per launch for launches with 0 / 1 / 2 residual operands."""
        if not os.path.exists(PMC_FILE):
            return None, {}
        with open(PMC_FILE) as f:
            pmc = json.load(f)
        sec = pmc.get(section) or (pmc if section == "f16" and "variants" in pmc else None)
        if not sec:
            return None, {}
        by = sec.get("by_residual_operands")
        n = sum(hist.values())
        if by and n and all(str(k) in by for k, c in hist.items() if c):
            tr = sum(by[str(k)]["hbm_bytes_per_launch_corrected"] * c for k, c in hist.items() if c) / n
            al = sum(by[str(k)]["algorithmic_bytes_per_launch"] * c for k, c in hist.items() if c) / n
        else:
            tr, al = sec.get("hbm_bytes_per_launch_corrected"), sec.get("algorithmic_bytes_per_launch")
        src_fp = hashlib.sha256(b"".join(open(os.path.join(ROOT, "deepsvc_amd", "csrc", f), "rb").read() for f in ("conv3x3_pp.hip", "conv3x3_pps.hip", "conv_mfma.hip", "conv_common.h"))).hexdigest()[:16]
        stale = pmc.get("kernel_sources_sha256_16") not in (None, src_fp)
        return tr, {"traffic_over_algorithmic": tr / al if tr and al else None, "traffic_algorithmic_bytes": al,
                    "traffic_pmc_collected_on": {"commit": pmc.get("commit"), "kernel_sources_sha256_16": pmc.get("kernel_sources_sha256_16")},
                    "traffic_kernel_sources_now_sha256_16": src_fp, "traffic_pmc_is_stale": bool(stale) if pmc.get("kernel_sources_sha256_16") else None,
                    "traffic_source": f"{os.path.relpath(PMC_FILE, ROOT)} [{section}]: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --lanes 1` "
                                      "(FETCH_SIZE x2 per MI355X_MICROARCH.md), bytes per launch by number of residual operands, weighted by this region's launch mix"}

    def roofline_of(durs, precision, hist, extra=None):
        if not durs:
            return None
        avg_ms = sum(durs) / len(durs)
        nres = [sum(k * c for k, c in hist.items()) / max(sum(hist.values()), 1)]
        achieved = conv_flop / (avg_ms * 1e-3) / 1e12
        esz = 2 if precision == "f16" else 4
        alg_bytes = (2.0 + nres[0]) * H_pad * W_pad * 64 * esz       # input + output + the residual operand(s), launch-weighted mean
        r = {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[precision],
             "traffic": None,
             "kernel": f"3x3 s1 64->64 conv @{H_pad}x{W_pad}, all epilogue variants ({'conv3x3_pp_kernel' if precision == 'f16' else 'conv3x3_pps_kernel<float> (two residual operands: conv_mfma_kernel)'})",
             "launches": len(durs), "avg_ms": avg_ms, "min_ms": min(durs),
             "algorithmic_flop": conv_flop, "algorithmic_bytes": alg_bytes, "mean_residual_operands": nres[0],
             "hbm_view": {"achieved_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK / 1e9, "frac": alg_bytes / (avg_ms * 1e-3) / HBM_PEAK,
                          "note": "algorithmic bytes = input + output + residual operand(s), launch-weighted over the epilogue variants; the kernel is "
                                  "not HBM-bound: a producer->consumer chain runs at the same 0.9 PFLOP/s on Infinity-Cache-resident 67 MB tensors "
                                  "(tools/bench_chain.py, DESIGN.md section 4.2)"}}
        if (h, w) == (1080, 1920) and rank == 0:
            r["traffic"], more = pmc_traffic(precision, hist)
            r.update(more)
        r.update(extra or {})
        return r

    # ---- headline leg ------------------------------------------------------------------------------------------------
    lanes = [make_lane(ln, args.precision) for ln in range(args.lanes)]
    warm(lanes, args.warmup)
    t_max, durs_shared, hist_shared = timed(lanes, args.steps)
    rows = [r for L in lanes for r in L["rows"]]
    hbm_peak_gb = torch.cuda.max_memory_allocated(dev) / 2**30       # incl. the one-off recording pass of the frame segments (all temporaries alive at once)
    hbm_steady_gb = torch.cuda.memory_allocated(dev) / 2**30       # what the lanes hold between frames: weights, workspaces, arenas, frames

    side = not args.no_side_legs
    lanes1 = parity = deltas = None
    durs_own, hist_own = (durs_shared, hist_shared) if (args.lanes == 1 or not side) else (None, {})     # without the lanes1 leg only the time-shared durations exist
    if side and args.lanes > 1:
        # ---- the same mode, ONE GOP in flight: the kernel's own durations ----
        for L in lanes:
            L["rows"] = []
        n1 = GOP if intra else GOP - 1
        t1, durs_own, hist_own = timed(lanes[:1], n1)
        lanes1 = {"value": n1 * world / t1, "unit": "frames/s", "steps": n1, "ms_per_step": 1000.0 * t1 / n1, "gop_lanes": 1}
    if side and args.precision == "f16" and not intra:
        # ---- reference precision: fp32 mode, one GOP in flight ----
        ops.set_precision("fp32")
        try:
            f32 = make_lane(0, "fp32")
            warm([f32], 2)
            n32 = GOP - 1
            t32, durs32, hist32 = timed([f32], n32)
            parity = {"dtype": "f32", "value": n32 * world / t32, "unit": "frames/s", "steps": n32, "ms_per_step": 1000.0 * t32 / n32,
                      "gop_lanes": 1, "roofline": roofline_of(durs32, "fp32", hist32),
                      "whole_frame_tflops": n32 / t32 * FLOP_PER_PX[2] * H_pad * W_pad / 1e12,
                      "note": "fp32 storage + v_mfma_f32_32x32x2_f32: the reference's arithmetic (PyTorch fp32); bitstreams byte-identical "
                              "to the reference's golden strings at 64/128/256 px (tests/test_gpu_parity.py)"}
            if args.lanes > 1:
                # ---- ... and with as many GOPs in flight as the headline leg (whole-job fp32 throughput of this GPU) ----
                more = [make_lane(ln, "fp32") for ln in range(1, args.lanes)]
                f32["rows"] = []
                warm([f32] + more, 2)
                t32n, _, _ = timed([f32] + more, n32)
                parity["at_headline_lanes"] = {"value": n32 * args.lanes * world / t32n, "unit": "frames/s", "steps": n32, "gop_lanes": args.lanes,
                                               "ms_per_step": 1000.0 * t32n / n32,
                                               "whole_frame_tflops": n32 * args.lanes / t32n * FLOP_PER_PX[2] * H_pad * W_pad / 1e12}
                del more
                f32["rows"] = []
                torch.cuda.empty_cache()
            if rank == 0:
                # ---- f16 mode vs fp32 mode on the same frames (each chained on its own decoded state), un-clamped tensors ----
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import parity_util as PU
                h16 = lanes[0]
                per = []
                st = {"f16": (pad(h16["frames"][0], 64), None), "fp32": (pad(h16["frames"][0], 64), None)}
                for t in range(1, GOP):               # the WHOLE GOP: the error of frame t - 1 is an input of frame t (test_video.py:368-369)
                    cur, sm = pad(h16["frames"][t], 64), h16["sm"][t]
                    got = {}
                    for mode, L in (("f16", h16), ("fp32", f32)):
                        ops.set_precision(mode)
                        ref, feat = st[mode]
                        L["model"].debug_taps = {}
                        mv, res = L["model"].compress(ref, cur, sm, feat)
                        feat2, recon, _, _ = L["model"].decompress(ref, mv, res, sm, feat)
                        taps, L["model"].debug_taps = L["model"].debug_taps, None
                        got[mode] = (harness.strings_bytes(mv) + harness.strings_bytes(res), PU.psnr(harness.crop(recon, (h, w)), h16["frames"][t]),
                                     taps["recon"].float(), feat2.float())
                        st[mode] = (recon, feat2)
                    (nb, pb, rb, fb), (nf, pf, rf, ff) = got["f16"], got["fp32"]
                    per.append({"frame": t, "bytes_f16": nb, "bytes_f32": nf, "dbpp_pct": 100.0 * (nb - nf) / nf, "psnr_f32_db": pf,
                                "dpsnr_db": pb - pf, "recon_rel_l2": float((rb - rf).norm() / rf.norm()),
                                "feature_rel_l2": float((fb - ff).norm() / ff.norm()),
                                "psnr_f16_vs_f32_recon_db": PU.psnr(rb, rf),
                                "clamped_frac_f32": float(((rf < 0) | (rf > 1)).float().mean())})
                worst = lambda k: max((abs(p[k]) for p in per))
                deltas = {"frames": per, "dbpp_pct": worst("dbpp_pct"), "dpsnr_db": worst("dpsnr_db"), "feature_rel_l2": worst("feature_rel_l2"),
                          "recon_rel_l2": worst("recon_rel_l2"),
                          "note": f"worst |delta| over the {GOP - 1} chained P-frames of lane 0's GOP, each mode chained on its own decoded state; "
                                  "un-clamped reconstruction and decoded 64-channel feature; north_star tolerance: 1 % bpp, 0.05 dB"}
                # ---- both modes against the REFERENCE's own run at this size (tests/golden/deepsvc_ref_1080p.npz: the reference's
                #      compress()/decompress() on synth.make_frames(1088, 1920, 3), made in the build container) ----
                gpath = os.path.join(ROOT, "tests", "golden", "deepsvc_ref_gop1080p.npz")
                if (H_pad, W_pad) == (1088, 1920) and os.path.exists(gpath):
                    import numpy as np
                    g = np.load(gpath)
                    n_ref = int(g["g1080/psnr"].shape[0])
                    gfr = [f.to(dev) for f in synth.make_frames(1088, 1920, n_ref + 1)]
                    vs = {}
                    for mode, L in (("fp32", f32), ("f16", h16)):
                        ops.set_precision(mode)
                        ref, feat, fr_rows = gfr[0], None, []
                        for t in range(1, n_ref + 1):
                            sm = synth.make_sm_fea(1088, 1920, t).to(dev)
                            mv, res = L["model"].compress(ref, gfr[t], sm, feat)
                            feat, recon, _, _ = L["model"].decompress(ref, mv, res, sm, feat)
                            mine = [e["strings"][i][0] for e in (mv, res) for i in (0, 1)]
                            nb, ng = sum(len(x) for x in mine), int(g["g1080/string_bytes"][t - 1].sum())
                            same = all(len(x) == int(g["g1080/string_bytes"][t - 1][k]) and zlib.crc32(x) == int(g["g1080/string_crc32"][t - 1][k]) for k, x in enumerate(mine))
                            e_s, e_b, _ = PU.golden_full_errors(g, f"g1080/f{t}/dec/recon", recon)
                            fr_rows.append({"frame": t, "bytes": nb, "bytes_reference": ng, "bytes_identical": same, "dbpp_pct": 100.0 * (nb - ng) / ng,
                                            "dpsnr_db": PU.psnr(recon, gfr[t]) - float(g["g1080/psnr"][t - 1]), "psnr_reference_db": float(g["g1080/psnr"][t - 1]),
                                            "recon_rel_l2_samples": e_s, "recon_rel_l2_block_means": e_b})
                            ref = recon
                        vs[mode] = fr_rows
                    deltas["vs_reference_golden"] = dict(vs, worst_f16_dbpp_pct=max(abs(r["dbpp_pct"]) for r in vs["f16"]),
                                                         worst_f16_dpsnr_db=max(abs(r["dpsnr_db"]) for r in vs["f16"]),
                                                         note=f"all {n_ref} P-frames of the reference's own 1088x1920 GOP (tests/golden/deepsvc_ref_gop1080p.npz: its "
                                                              "compress()/decompress() chained on its own decoded state, test_video.py:332-369), each mode chained on ITS own "
                                                              "decoded state; clamped decoded frames, strided samples and block means; asserted in tests/test_gpu_fullsize.py")
        finally:
            ops.set_precision(args.precision)

    # ---- metrics (outside the timed regions) -------------------------------------------------------------------------
    metric_rows, psnr_of = [], {}
    for unit, t, bmv, bres, recon, cur in rows:
        key = (int(unit) // 100, int(t))                       # (rank, lane), frame index inside the GOP
        if recon is not None:
            mse = float(((harness.crop(recon, (h, w)) - harness.crop(cur, (h, w))) ** 2).mean())
            psnr_of.setdefault(key, -10 * torch.log10(torch.tensor(mse)).item() if mse > 0 else 99.0)
        metric_rows.append([unit, t, bmv, bres, psnr_of[key], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    table = harness.gather_rows(metric_rows, dev, world)

    if rank == 0:
        total_frames = args.steps * world * args.lanes
        fps = total_frames / t_max
        roof = roofline_of(durs_own, args.precision, hist_own)
        if roof is not None:
            roof["measured_in"] = "lanes1 leg (one GOP in flight): HIP events on the launch stream around every launch of this shape" if args.lanes > 1 and side \
                else ("timed region (one GOP in flight): HIP events on the launch stream around every launch of this shape" if args.lanes == 1 else
                      f"timed region with {args.lanes} lanes in flight (--no-side-legs): durations are TIME-SHARED with the other lanes' kernels, not the kernel's own cost")
            if args.lanes > 1 and side and durs_shared:
                ts = sum(durs_shared) / len(durs_shared)
                roof["timeshared_avg_ms"] = ts
                roof["timeshared_frac"] = conv_flop / (ts * 1e-3) / 1e12 / PEAK_TFLOPS[args.precision]
                roof["timeshared_note"] = f"the same launches inside the {args.lanes}-lane headline region: a launch shares the GPU with the other lanes' kernels, so this is not the kernel's cost"
        summ = harness.summarise(table)
        name = "encode+decode fps @1080p" if (h, w) == (1080, 1920) else f"encode+decode fps @{w}x{h}"
        line = {
            "metric": name, "value": fps, "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.precision == "f16" else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{args.config - 1}]: {w}x{h} (padded {W_pad}x{H_pad}) GOP={GOP} "
                                   + ("(key frame coded by ICIP2020ResB + 11 P-frames), " if intra else "(key frame given + 6 P-frames), ")
                                   + f"W0 synthetic PSNR weight set, compress+decompress per frame, GOPs sharded across ranks, {args.lanes} independent "
                                     f"GOP lane(s) in flight per GPU (one step = one frame on every lane)",
                       "precision": args.precision, "coder": args.coder + (" (the reference's bitstream format)" if args.coder == "rans" else " (NON-reference multi-stream format, labelled)"),
                       "gop_lanes": args.lanes, "frames_per_step": args.lanes * world,
                       "hbm_peak_gb": hbm_peak_gb, "hbm_steady_gb": hbm_steady_gb,
                       "host": dict(host, torch_threads=torch.get_num_threads(), python_threads_alive=__import__("threading").active_count()), "bpp": summ["bpp"], "psnr_db": summ["ppsnr"],
                       "whole_frame_tflops": fps * FLOP_PER_PX[args.config] * H_pad * W_pad / 1e12 / world,
                       # the other legs of this run as scalars (details in the objects of the same names below)
                       "lanes1_fps": None if lanes1 is None else lanes1["value"],
                       "lanes1_ms_per_frame": None if lanes1 is None else lanes1["ms_per_step"],
                       "parity_mode_fp32_fps": None if parity is None else parity["value"],
                       "parity_mode_fp32_roofline_frac": None if parity is None or parity["roofline"] is None else parity["roofline"]["frac"],
                       "parity_mode_fp32_fps_at_headline_lanes": None if parity is None or "at_headline_lanes" not in parity else parity["at_headline_lanes"]["value"],
                       "roofline_frac": None if roof is None else roof["frac"]},
            "roofline": roof,
        }
        if lanes1 is not None:
            line["lanes1"] = lanes1
        if parity is not None:
            line["parity_mode"] = parity
        if deltas is not None:
            line["f16_vs_f32"] = deltas
        if not args.no_cpu_baseline and world == 1:
            import torch as _t
            # all-core leg: this process's CPU share on a 1-GPU box is 16 cores; more threads than that on a 256x256 image only
            # add synchronisation cost (128 threads measured slower than 1)
            cores = min(_t.get_num_threads(), 16)
            small = cpu_baseline((args.cpu_sample, args.cpu_sample), args.cpu_frames, cores, H_pad, W_pad)
            if args.no_cpu_full_frame:
                line["cpu_baseline"] = small
            else:         # one measured P-frame at the benchmark's own size, all cores of this process's share
                line["cpu_baseline"] = cpu_baseline((H_pad, W_pad), 1, cores, H_pad, W_pad)
                line["cpu_baseline_small_sample"] = small
            line["cpu_baseline_1thread"] = cpu_baseline((args.cpu_sample, args.cpu_sample), args.cpu_frames, 1, H_pad, W_pad)
        print(json.dumps(line), flush=True)
    # every rank says what it ran on (stderr: stdout carries only rank 0's JSON line)
    import threading
    sys.stderr.write(f"[bench rank {rank}/{world}] cores visible {host['cores_visible']}, pinned to {host['cores_pinned']}, torch threads {torch.get_num_threads()}, "
                     f"python threads alive {threading.active_count()}, lanes {args.lanes}\n")
    if use_pg:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
