"""Turn the rocprofv3 CSVs that tools/collect_profiles.sh leaves under gpurun_out/<tag>/ into the summaries committed under
profiles/ (round prefix r03):

  r03_bench_<run>_kernel_stats.csv          rocprofv3 --stats table of the run, verbatim
  r03_bench_<run>_by_kernel_and_grid.csv    the same trace grouped by (kernel name, grid, workgroup): one row per launch SHAPE,
                                            so that a per-shape duration (hence a roofline fraction) can be re-derived
  r03_bench_<run>_line.json                 the JSON line bench.py printed under the profiler
  r03_dominant_kernel_pmc.json              HBM traffic per launch of the dominant conv (FETCH_SIZE / WRITE_SIZE from separate
                                            --pmc passes, corrected as MI355X_MICROARCH.md prescribes) + MFMA-utilisation counters

    python tools/prof_summary.py gpurun_out/r2p
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
PRE = sys.argv[2] if len(sys.argv) > 2 else "r04"
OUT = os.path.join(ROOT, "profiles")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(dsvc_conv::ConvParams.*$", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_116conv_mfma_kernelI(DF16_|f)Li(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d)EEEvN9dsvc", n)
    if m:
        n = f"conv_mfma_kernel<{'f16' if m.group(1) == 'DF16_' else 'f32'}, MT={m.group(2)}, NT={m.group(3)}, FAST={m.group(4)}, REGEPI={m.group(5)}, PF={m.group(6)}>"
    m = re.match(r"_ZN12_GLOBAL__N_118conv3x3_pps_kernelI(DF16_|f)((?:Lb\dE)+)EEvN9dsvc", n)
    if m:
        flags = re.findall(r"Lb(\d)E", m.group(2)) + ["0", "0", "0"]
        n = f"conv3x3_pps_kernel<{'f16' if m.group(1) == 'DF16_' else 'float'}, ACT={flags[0]}, EXT={flags[1]}, R1P={flags[2]}>"
    return n


def by_shape(trace_csv, out_csv):
    agg = collections.OrderedDict()
    rows = []
    with open(trace_csv) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]),
                   int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
            rows.append((key, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    # conv3x3_pp_kernel is persistent: 256 workgroups at every image size, so name + grid do not separate its full-resolution
    # launches (1088x1920) from the half-resolution ones of the UNets (544x960, a quarter of the work).  Their durations are
    # bimodal (e.g. 41 vs 117 us); a launch shorter than 0.45 x the variant's 90th-percentile duration is labelled @1/2.
    import statistics
    pp = collections.defaultdict(list)
    for key, d in rows:
        if key[0].startswith("conv3x3_pp_kernel"):
            pp[key].append(d)
    thr = {k: 0.45 * statistics.quantiles(v, n=10)[-1] if len(v) >= 10 else 0.0 for k, v in pp.items()}
    for key, d in rows:
        if key in thr:
            key = (key[0] + (" @1/2 res (by duration)" if d < thr[key] else " @1/1 res (by duration)"),) + key[1:]
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroups_x", "grid_y", "workgroup_size", "lds_bytes", "vgprs", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow(list(k) + [a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.2f}", f"{a[2]:.2f}", f"{a[3]:.2f}", f"{100 * a[1] / tot:.2f}"])
    return agg


for run, name in (("l1", "f16_lanes1"), ("l4", "f16_headline_lanes"), ("f32", "fp32_lanes1")):
    d = os.path.join(src, run)
    if not os.path.isdir(d):
        continue
    shutil.copy(os.path.join(d, f"{run}_kernel_stats.csv"), os.path.join(OUT, f"{PRE}_bench_{name}_kernel_stats.csv"))
    by_shape(os.path.join(d, f"{run}_kernel_trace.csv"), os.path.join(OUT, f"{PRE}_bench_{name}_by_kernel_and_grid.csv"))
    line = [l for l in open(os.path.join(src, f"{run}.stdout")) if l.startswith("{")][-1]
    with open(os.path.join(OUT, f"{PRE}_bench_{name}_line.json"), "w") as f:
        json.dump(json.loads(line), f, indent=1)


def counters(path):
    """-> {(kernel short name, workgroups): {counter: [values per dispatch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            out[(short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


pf, pw, ps = (os.path.join(src, d, "p_counter_collection.csv") for d in ("pmc_fetch", "pmc_write", "pmc_sq"))
if all(os.path.exists(p) for p in (pf, pw, ps)):
    F, W, S = counters(pf), counters(pw), counters(ps)
    H, Wd = 1088, 1920
    esz = 2
    variants = {}
    for key in F:
        name, wgs = key
        if not name.startswith("conv3x3_pp_kernel<2, 2,") or wgs != 256 or key not in W:
            continue
        fv, wv = F[key]["FETCH_SIZE"], W[key]["WRITE_SIZE"]
        # the kernel is persistent (always 256 workgroups), so name + grid do not tell full-resolution launches from the
        # half-resolution ones of the UNets: the passes run the same command, so dispatch i of a kernel name is the same launch
        # in each pass -- keep the launches whose WRITE_SIZE is the full-resolution output (1088*1920*64*2 B = 261120 KB)
        full_kb = H * Wd * 64 * esz / 1024.0
        if len(fv) != len(wv):
            continue
        keep = [i for i, x in enumerate(wv) if abs(x - full_kb) < 0.03 * full_kb]
        if not keep:
            continue
        fv, wv = [fv[i] for i in keep], [wv[i] for i in keep]
        if key in S and all(len(v) == len(W[key]["WRITE_SIZE"]) for v in S[key].values()):
            S[key] = {c: [v[i] for i in keep] for c, v in S[key].items()}
        elif key in S and "SQ_INSTS_MFMA" in S[key]:
            # a pass with another step count: full-resolution launches are those issuing the full MFMA count
            # (2*1088*1920*64*64*9 FLOP / 32768 FLOP per 32x32x16 MFMA = 4 718 592; a fused 1x1 tail adds 1/9 of that)
            mf = S[key]["SQ_INSTS_MFMA"]
            k2 = [i for i, x in enumerate(mf) if x >= 4718592 * 0.99]
            S[key] = {c: [v[i] for i in k2] for c, v in S[key].items() if len(v) == len(mf)}
            if not k2:
                del S[key]
        n = len(fv)
        fkb, wkb = sum(fv) / n, sum(wv) / len(wv)
        flags = [x.strip() == "true" for x in name[name.index("<") + 1:name.index(">")].split(",")[2:]]   # HEAD, PRE, ACT, R1, R2, POST, M16
        nres = int(flags[3]) + int(flags[4])
        alg = (2 + nres) * H * Wd * 64 * esz
        ent = {"launches": n, "FETCH_SIZE_KB_avg": fkb, "WRITE_SIZE_KB_avg": wkb,
               "hbm_bytes_per_launch_corrected": 2 * fkb * 1024 + wkb * 1024, "algorithmic_bytes_per_launch": alg,
               "traffic_over_algorithmic": (2 * fkb * 1024 + wkb * 1024) / alg}
        if key in S:
            sq = {c: sum(v) / len(v) for c, v in S[key].items()}
            cyc = sq.get("GRBM_GUI_ACTIVE", 0) / 8.0
            ent["sq_counters"] = dict(sq, **{"shader_cycles_per_launch (GRBM_GUI_ACTIVE / 8 XCDs)": cyc,
                                             "mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs)": sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
                                             "wave_wait_fraction = SQ_WAIT_ANY / SQ_WAVE_CYCLES": sq.get("SQ_WAIT_ANY", 0) / sq["SQ_WAVE_CYCLES"] if sq.get("SQ_WAVE_CYCLES") else None})
        variants[name] = ent
    tot_l = sum(v["launches"] for v in variants.values())
    mean_traffic = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in variants.values()) / tot_l
    mean_alg = sum(v["algorithmic_bytes_per_launch"] * v["launches"] for v in variants.values()) / tot_l
    rec = {"kernel": "conv3x3_pp_kernel<NT=2, NCH=2, HEAD, PRE, ACT, R1, R2, POST>: every 3x3 s1 64->64 launch @1088x1920 of one P-frame, f16 storage",
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc GRBM_GUI_ACTIVE SQ_... (one pass each, with --kernel-trace only) -- python3 bench.py --lanes 1 --steps 4 --warmup 3 --no-side-legs --no-cpu-baseline",
           "correction": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> x2; WRITE_SIZE as reported; both in KB",
           "hbm_bytes_per_launch_corrected": mean_traffic, "algorithmic_bytes_per_launch": mean_alg, "launches": tot_l,
           "selection": "full-resolution launches only (WRITE_SIZE within 3 % of 1088*1920*64*2 B): the persistent kernel's name and grid are the same at 1/2 resolution",
           "note": "launch-weighted mean over the epilogue variants below (the `roofline` of the bench line averages the same launches); algorithmic = input + output (+ one tensor per residual operand), 2 bytes per element",
           "variants": variants}
    def by_nres(entries):
        """{residual operands: launch-weighted bytes per launch} -- what bench.py weights with its own launch mix"""
        out = {}
        for k in (0, 1, 2):
            es = [e for e in entries if e["nres"] == k]
            if es:
                n = sum(e["launches"] for e in es)
                out[str(k)] = {"launches": n,
                               "hbm_bytes_per_launch_corrected": sum(e["hbm_bytes_per_launch_corrected"] * e["launches"] for e in es) / n,
                               "algorithmic_bytes_per_launch": sum(e["algorithmic_bytes_per_launch"] * e["launches"] for e in es) / n}
        return out

    for name, v in variants.items():
        fl = [x.strip() == "true" for x in name[name.index("<") + 1:name.index(">")].split(",")[2:]]
        v["nres"] = int(fl[3]) + int(fl[4])
    rec["by_residual_operands"] = by_nres(list(variants.values()))
    full = {"f16": rec}
    # ---- fp32 mode: the same conv shape on the generic kernel (conv_mfma_kernel<f32, MT=2, NT=2>, 8160 workgroups); the epilogue variant is a
    # run-time parameter there, so launches are classed by how many 64-channel fp32 tensors they READ (1 + residual operands)
    pf32, pw32 = (os.path.join(src, d, "p_counter_collection.csv") for d in ("pmc_fetch_f32", "pmc_write_f32"))
    if os.path.exists(pf32) and os.path.exists(pw32):
        F32c, W32c = counters(pf32), counters(pw32)
        tensor = H * Wd * 64 * 4
        entries = []
        for key in F32c:
            name, wgs = key
            general = (name.startswith("conv_mfma_kernel<f32, MT=2, NT=2") or name.startswith("conv_mfma_kernel<float, 2, 2,")) and wgs == 8160
            streamed = name.startswith("conv3x3_pps_kernel<float") and wgs == 256         # round 4: the 64 -> 64 family's two-group kernel (persistent grid)
            if not (general or streamed) or key not in W32c:
                continue
            fv, wv = F32c[key]["FETCH_SIZE"], W32c[key]["WRITE_SIZE"]
            if len(fv) != len(wv):
                continue
            groups = collections.defaultdict(list)
            for f_kb, w_kb in zip(fv, wv):
                if abs(w_kb * 1024 - tensor) > 0.03 * tensor:
                    continue                               # another output width
                reads = 2 * f_kb * 1024 / tensor
                k = int(round(reads)) - 1
                if k in (0, 1, 2) and abs(reads - (k + 1)) < 0.35:
                    groups[k].append((2 * f_kb * 1024 + w_kb * 1024))
            for k, vals in groups.items():
                entries.append({"nres": k, "launches": len(vals), "hbm_bytes_per_launch_corrected": sum(vals) / len(vals),
                                "algorithmic_bytes_per_launch": (2 + k) * tensor})
        if entries:
            n = sum(e["launches"] for e in entries)
            full["fp32"] = {"kernel": "conv3x3_pps_kernel<float> (0 / 1 residual operands) + conv_mfma_kernel<f32, MT=2, NT=2, FAST=1> (2 residual operands): 3x3 s1 64->64 "
                                      "launches @1088x1920 of one P-frame, fp32 storage (launches reading 2 tensors include the few two-segment 64+64->64 convs)",
                            "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (one pass each) -- python3 bench.py --precision fp32 --lanes 1 --steps 2 --warmup 2 --no-side-legs --no-cpu-baseline",
                            "launches": n,
                            "hbm_bytes_per_launch_corrected": sum(e["hbm_bytes_per_launch_corrected"] * e["launches"] for e in entries) / n,
                            "algorithmic_bytes_per_launch": sum(e["algorithmic_bytes_per_launch"] * e["launches"] for e in entries) / n,
                            "by_residual_operands": by_nres(entries)}
    # what these counters were collected ON: bench.py compares the hash with the kernel sources of the run that quotes the file
    # (`roofline.traffic_pmc_is_stale`), so a kernel change without a re-collected PMC file shows in the bench line
    import hashlib
    import subprocess
    csrc = os.path.join(ROOT, "deepsvc_amd", "csrc")
    full["kernel_sources_sha256_16"] = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("conv3x3_pp.hip", "conv3x3_pps.hip", "conv_mfma.hip", "conv_common.h"))).hexdigest()[:16]
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "deepsvc_amd/csrc"]) != 0
        full["commit"] = head + ("+uncommitted csrc changes" if dirty else "")
    except Exception:
        full["commit"] = None
    with open(os.path.join(OUT, f"{PRE}_dominant_kernel_pmc.json"), "w") as f:
        json.dump(full, f, indent=1)
    # traffic of every kernel, for the pass-count table of DESIGN.md
    with open(os.path.join(OUT, f"{PRE}_bench_f16_pmc_traffic_by_kernel.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroups", "launches", "fetch_MB_per_launch_corrected", "write_MB_per_launch", "total_GB_all_launches"])
        rows = []
        for key in F:
            if key not in W:
                continue
            n = len(F[key]["FETCH_SIZE"])
            fm, wm = 2 * sum(F[key]["FETCH_SIZE"]) / n / 1024, sum(W[key]["WRITE_SIZE"]) / len(W[key]["WRITE_SIZE"]) / 1024
            rows.append((key[0], key[1], n, fm, wm, (fm + wm) * n / 1024))
        for r in sorted(rows, key=lambda r: -r[5]):
            w.writerow([r[0], r[1], r[2], f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.3f}"])
print(os.listdir(OUT))
