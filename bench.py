#!/usr/bin/env python3
"""bench.py -- encode+decode frames/s of the DeepSVC P-frame hot path at 1080p (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|bf16] [--size HxW]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

A "step" = one P-frame: `DeepSVC.compress` + `DeepSVC.decompress` on a 1920x1080 frame (padded to 1088x1920),
frames chained inside GOPs of 7 (1 given key frame + 6 P-frames, BASELINE.json configs[1]); synthetic septuplets
and the W0 weight set (deepsvc_amd/synth.py) are resident in HBM before the timed region.  Each rank codes its
own GOPs (weak scaling, no data-path collective); one metrics gather at the end.
Prints ONE JSON line on rank 0 (contract in the task description): value = frames of all ranks / max-over-ranks time.
Extra objects: "roofline" (dominant kernel = 3x3 64->64 conv at full resolution, per-launch HIP-event timing
inside the timed region) and "cpu_baseline" (the CPU oracle timed on this host on a bounded 256x256 sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOP = 7
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}      # MI355X_MICROARCH.md: dense f32 / bf16 MFMA peaks
FLOP_PER_PX_GOP7 = 8.571e6                           # GOP=7 mean conv FLOP per padded pixel per P-frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=4, help="independent GOP pipelines in flight per GPU (1 = strictly sequential frames)")
    ap.add_argument("--precision", default=os.environ.get("DSVC_PRECISION", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--size", default="1080x1920")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256)
    return ap.parse_args()


def cpu_baseline(sample, H_pad, W_pad):
    """The oracle (CPU restatement of the reference path, oracle/) timed on this host: one chained pair of
    P-frames at sample x sample, all host threads; scaled to 1080p by the pixel ratio (cost is per-pixel:
    SURVEY section 8d).  Baseline only -- never the measured path."""
    from deepsvc_amd import synth
    from oracle.codec import DeepSVCOracle
    model = synth.make_model("W0")
    orc = DeepSVCOracle(model.state_dict())
    orc.update()
    fr = synth.make_frames(sample, sample, 3)
    ref, feat = fr[0], None
    t_tot, n = 0.0, 0
    for t in (1, 2):
        sm = synth.make_sm_fea(sample, sample, t)
        t0 = time.perf_counter()
        mv, res = orc.compress(ref, fr[t], sm, feat)
        feat, ref, _, _ = orc.decompress(ref, mv, res, sm, feat)
        t_tot += time.perf_counter() - t0
        n += 1
    fps_sample = n / t_tot
    scale = (sample * sample) / float(H_pad * W_pad)
    return {"value": fps_sample * scale, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} chained P-frames (compress+decompress) at {sample}x{sample}, {t_tot:.1f} s of CPU, "
                      f"scaled by pixel ratio to {H_pad}x{W_pad}; oracle = PyTorch-CPU restatement of the reference"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("DSVC_BENCH_REHEARSAL"):      # N ranks sharing the ONE GPU of a development box: gloo, all on cuda:0
            dist.init_process_group(backend="gloo")        # (exercises the rank / barrier / gather plumbing, not RCCL)
            local_rank = 0
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from deepsvc_amd import harness, ops, synth
    from deepsvc_amd.modules import pad
    ops.set_precision(args.precision)
    h, w = (int(v) for v in args.size.lower().split("x"))
    H_pad, W_pad = -(-h // 64) * 64, -(-w // 64) * 64

    # L independent GOP pipelines ("lanes") per GPU: GOPs are independent units of work (harness.py), and one GOP alone
    # leaves the GPU idle ~20 % of a frame while the host range-decodes each scan level (sequential by bitstream format).
    # Each lane = its own model instance (weights, workspaces, HIP graphs), stream and thread, coding its own GOPs; a
    # step codes one P-frame on EVERY lane (a batch of L frames), so the line reports L*K frames per rank.
    lanes = []
    for ln in range(args.lanes):
        m = synth.make_model("W0")
        m.update(force=True)
        seed = 1234 + rank * 16 + ln
        lanes.append({
            "model": m.to(dev), "stream": torch.cuda.Stream(dev), "t": 0, "ref": None, "feature": None, "rows": [], "id": ln,
            "frames": [f.to(dev) for f in synth.make_frames(h, w, GOP, seed=seed)],
            "sm": [None] + [synth.make_sm_fea(H_pad, W_pad, t, seed=seed).to(dev) for t in range(1, GOP)]})

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def step(L, record):
        model, frames, sm_feas = L["model"], L["frames"], L["sm"]
        t = L["t"] % (GOP - 1) + 1                    # P-frame index inside the GOP: 1..6
        if t == 1:
            L["ref"], L["feature"] = pad(frames[0], 64), None
        cur = pad(frames[t], 64)
        mv_enc, res_enc = model.compress(L["ref"], cur, sm_feas[t], L["feature"])
        feature, recon, warped, predict = model.decompress(L["ref"], mv_enc, res_enc, sm_feas[t], L["feature"])
        if record:
            L["rows"].append([float(rank * 1000 + L["id"] * 100 + L["t"] // (GOP - 1)), float(t),
                              harness.strings_bytes(mv_enc) * 8.0 / (h * w), harness.strings_bytes(res_enc) * 8.0 / (h * w),
                              recon, cur])
        L["ref"], L["feature"] = recon, feature
        L["t"] += 1

    def run_lane(L, n, record):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(L["stream"]):
            for _ in range(n):
                step(L, record)
            L["stream"].synchronize()

    # Warm-up, one lane at a time: weight packing and HIP-graph capture need a quiescent device (no other thread
    # launching).  W steps per lane; with several lanes a lane's graphs are only final once EVERY lane has packed its
    # weights, so lanes whose graphs are not in place after W steps get the few extra untimed steps they need.
    for w_i in range(max(args.warmup, 1)):
        for L in lanes:
            run_lane(L, 1, False)
    for _ in range(4):
        todo = [L for L in lanes if not L["model"].graphs_ready(1, H_pad, W_pad, dev)]
        if not todo:
            break
        for L in todo:
            run_lane(L, 1, False)
    # the roofline leg: time every launch of the dominant conv shape inside the timed region
    full_res = lambda pc, x0: (pc.kh == 3 and pc.stride == 1 and pc.segC == (64,) and pc.Cout == 64 and x0.H == H_pad and x0.W == W_pad)
    ops.PROFILE = {"match": full_res, "pairs": []}
    barrier()
    t0 = time.perf_counter()
    if len(lanes) == 1:
        run_lane(lanes[0], args.steps, True)
    else:
        import threading
        threads = [threading.Thread(target=run_lane, args=(L, args.steps, True)) for L in lanes]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    rows = [r for L in lanes for r in L["rows"]]
    # the same launch on an otherwise idle GPU (after the timed region): with several lanes the in-region durations
    # include time-sharing with the other lanes' kernels, so both figures are reported
    iso_ms = None
    if rank == 0:
        from deepsvc_amd.modules import Conv2d
        st = ops.RT.storage
        conv = Conv2d(64, 64, 3, 1, 1).to(dev)
        xin = ops.Act(torch.randn((1, H_pad, W_pad, 64), device=dev).to(st))
        xout = ops.new_act(1, H_pad, W_pad, 64, st, dev)
        for _ in range(3):
            conv.run(xin, xout)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            conv.run(xin, xout)
        e1.record()
        torch.cuda.synchronize()
        iso_ms = e0.elapsed_time(e1) / 10

    # metrics (outside the timed region)
    metric_rows = []
    for unit, t, bmv, bres, recon, cur in rows:
        mse = float(((recon - cur) ** 2).mean())
        metric_rows.append([unit, t, bmv, bres, -10 * torch.log10(torch.tensor(mse)).item() if mse > 0 else 99.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    table = harness.gather_rows(metric_rows, dev, world)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    t_max = float(t_max)

    if rank == 0:
        total_frames = args.steps * world * args.lanes
        fps = total_frames / t_max
        durs = [a.elapsed_time(b) for a, b in prof["pairs"]]          # ms, HIP events on the launch stream
        conv_flop = 2.0 * H_pad * W_pad * 64 * 64 * 9
        roof = None
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_dominant_kernel_pmc.json")
        if args.precision == "bf16" and (h, w) == (1080, 1920) and os.path.exists(pmc_file):
            with open(pmc_file) as f:       # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (cannot run in-process)
                traffic = json.load(f)["hbm_bytes_per_launch_corrected"]
            traffic_src = "profiles/r01_dominant_kernel_pmc.json (2*FETCH_SIZE + WRITE_SIZE per launch, MI355X_MICROARCH.md HBM correction)"
        if durs:
            avg_ms = sum(durs) / len(durs)
            achieved = conv_flop / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                    "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes": 2.0 * H_pad * W_pad * 64 * (2 if args.precision == "bf16" else 4),
                    "kernel": "3x3 s1 64->64 conv @%dx%d (bf16: conv3x3_pp_kernel, fp32: conv_mfma_kernel)" % (H_pad, W_pad),
                    "launches": len(durs), "avg_ms": avg_ms, "isolated_avg_ms": iso_ms,
                    "isolated_achieved": conv_flop / (iso_ms * 1e-3) / 1e12 if iso_ms else None,
                    "isolated_frac": conv_flop / (iso_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.precision] if iso_ms else None,
                    "isolated_hbm_frac": (2.0 * H_pad * W_pad * 64 * (2 if args.precision == "bf16" else 4) / (iso_ms * 1e-3) / 8e12) if iso_ms else None,
                    "concurrent_lanes": args.lanes,
                    "note": "achieved/frac/avg_ms: HIP events around every launch of this shape inside the timed region -- with "
                            "concurrent_lanes > 1 the launch shares the GPU with the other lanes' kernels, so its duration (and the "
                            "rocprofv3 per-kernel average of the same command) is ~lanes x the kernel's own time; isolated_*: the same "
                            "launch on an idle GPU after the timed region = what `--lanes 1` and its rocprofv3 summary show "
                            "(arithmetic intensity 288 FLOP/B sits at the machine balance 2500/8 = 312, hence both fractions)"}
        summ = harness.summarise(table)
        line = {
            "metric": "encode+decode fps @1080p" if (h, w) == (1080, 1920) else f"encode+decode fps @{w}x{h}", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{w}x{h} (padded {W_pad}x{H_pad}) GOP={GOP} P-frames, W0 synthetic PSNR weight set, "
                                   f"compress+decompress per frame, GOPs sharded across ranks, {args.lanes} independent GOP lane(s) in flight per GPU "
                                   f"(one step = one P-frame on every lane)", "precision": args.precision, "gop_lanes": args.lanes,
                       "hbm_peak_gb": torch.cuda.max_memory_allocated(dev) / 2**30,
                       "bpp": summ["bpp"], "psnr_db": summ["ppsnr"],
                       "whole_frame_tflops": fps * FLOP_PER_PX_GOP7 * H_pad * W_pad / 1e12 / world},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, H_pad, W_pad)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
