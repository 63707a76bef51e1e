"""Build libdsvc.so (HIP kernels + host coder) for gfx950, in-tree, with hipcc.

    python -m deepsvc_amd.build [--force]

The shared library lands in deepsvc_amd/lib/ (git-ignored; it travels to the GPU box with the
repo snapshot).  hipcc cross-compiles without a GPU.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libdsvc.so")
SOURCES = ["conv_mfma.hip", "conv3x3_ws.hip", "conv3x3_pp.hip", "conv3x3_pps.hip", "conv_head.hip", "conv_headtap.hip", "conv_cin8.hip", "conv7x7_rr.hip", "conv_head_f32.hip", "conv1x1.hip", "conv1x1_f32.hip", "pointwise.hip", "msssim.hip", "entropy.hip", "rans_ms.hip", "rans_host.cpp", "host_misc.cpp", "pframe_program.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(HERE, "..", "include", "dsvc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("DSVC_BUILD_DEFINES", "").split()


# kernel -> scratch bytes per lane tolerated by the build check below
SCRATCH_ALLOWED = {
    # opt-in variant (option "ws_waves4": one wave per SIMD with a 4 x 2 register tile, 256 VGPRs): three dwords spilled around its epilogue
    "_ZN12_GLOBAL__N_117conv3x3_ws_kernelILi2ELb1ELb0ELi4EEEvN9dsvc_conv10ConvParamsEi": 16,
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _resource_usage(remarks):
    """hipcc -Rpass-analysis=kernel-resource-usage output -> {kernel: {vgprs, agprs, sgprs, scratch, lds, occupancy}}."""
    out = {}
    for blk in remarks.split("Function Name: ")[1:]:
        name = blk.split()[0]
        num = lambda pat: int(m.group(1)) if (m := re.search(pat, blk)) else -1
        out[name] = {"vgprs": num(r" VGPRs: (\d+)"), "agprs": num(r"AGPRs: (\d+)"), "sgprs": num(r"SGPRs: (\d+)"),
                     "scratch": num(r"ScratchSize \[bytes/lane\]: (\d+)"), "lds": num(r"LDS Size \[bytes/block\]: (\d+)"),
                     "occupancy": num(r"Occupancy \[waves/SIMD\]: (\d+)")}
    return out


_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_WIDE_STORE_SOFF = re.compile(r"\b(buffer|tbuffer)_store_(dwordx[34]|format_xyzw?)\s+v\[\d+:\d+\],\s*(?:v\d+|v\[\d+:\d+\]|off),\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b")


def wide_stores_with_register_soffset(obj):
    """Disassembles the gfx950 code object inside a compiled .hip object and returns the wide buffer stores (more than 64 bits of data) whose
    scalar offset operand is a REGISTER.  On gfx950 with ROCm 7.2 such a store may be followed directly by a vector instruction that overwrites
    its data registers (the compiler inserts the wait state only for stores WITHOUT a register offset), and under memory load the store then
    writes the new register contents: the two-group conv's epilogue produced run-to-run different frames that way in round 5 (isolated launches
    were always right; DESIGN.md section 4.2 [r5]).  The library therefore contains no such instruction: the build refuses one."""
    if not os.path.exists(_OBJDUMP):
        return None
    import glob
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.run([_OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, cwd=tmp)
        dev = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
        if not dev:
            return None
        dis = subprocess.run([_OBJDUMP, "-d", dev[0]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False).stdout.decode(errors="replace")
    if "s_endpgm" not in dis:
        return None
    return [m.group(0) for m in _WIDE_STORE_SOFF.finditer(dis)]


def resource_usage(variant=None):
    """Per-kernel register / scratch / LDS use of the last build (lib/obj*/ *.usage.json, written while compiling)."""
    obj_dir = os.path.join(LIB_DIR, "obj" if variant is None else f"obj_{variant}")
    out = {}
    for src in SOURCES:
        up = os.path.join(obj_dir, os.path.splitext(src)[0] + ".usage.json")
        if os.path.exists(up):
            with open(up) as f:
                out.update(json.load(f))
    return out


def build(force=False, verbose=False, variant=None, defines=()):
    """Compile each source to an object (only when stale) and link libdsvc.so.  Returns the path.
    variant="clock", defines=["-DDSVC_PP_CLOCK"]: a diagnostic build next to the product one (lib/libdsvc_clock.so, own object
    directory), loaded with DSVC_LIB=... by the tools that read its probes."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj" if variant is None else f"obj_{variant}")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    LIB = os.path.join(LIB_DIR, "libdsvc.so" if variant is None else f"libdsvc_{variant}.so")
    FLAGS = globals()["FLAGS"] + list(defines)
    objs, relink = [], force or not os.path.exists(LIB)
    procs = []
    # objects / reports of sources that are no longer part of the library (a removed kernel file) must not be linked, gated or counted
    keep = {os.path.splitext(s)[0] for s in SOURCES}
    for f in os.listdir(obj_dir):
        stem = f.split(".")[0]
        if stem not in keep and f.endswith((".o", ".usage.json", ".stores.json")):
            os.remove(os.path.join(obj_dir, f))
            relink = True
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        up = os.path.splitext(op)[0] + ".usage.json"
        # (an object without its resource report -- or with an older one -- is recompiled: the scratch gate below must describe what ships)
        if force or _stale(op, [sp] + HEADERS) or (src.endswith(".hip") and (not os.path.exists(up) or os.path.getmtime(up) < os.path.getmtime(op))):
            cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, op, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            relink = True
    for src, op, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if src.endswith(".hip"):
            usage = _resource_usage(out.decode(errors="replace"))
            # the gate must not go inert: a compile whose remarks could not be parsed (wording change, remarks missing) is an error
            broken = [k for k, v in usage.items() if min(v["scratch"], v["vgprs"], v["lds"]) < 0]
            if not usage or broken:
                raise RuntimeError(f"{src}: no usable -Rpass-analysis=kernel-resource-usage report "
                                   f"({'no kernels parsed' if not usage else 'unparsed fields for ' + ', '.join(broken[:3])}): the scratch gate cannot vouch for this object")
            with open(os.path.splitext(op)[0] + ".usage.json", "w") as f:
                json.dump(usage, f, indent=0, sort_keys=True)

    # No wide buffer store with a register scalar offset (wide_stores_with_register_soffset): every .hip object is disassembled once
    # (report next to the object, redone when the object is newer)
    for src, op in zip(SOURCES, objs):
        if not src.endswith(".hip"):
            continue
        rp = os.path.splitext(op)[0] + ".stores.json"
        if not os.path.exists(rp) or os.path.getmtime(rp) < os.path.getmtime(op):
            hits = wide_stores_with_register_soffset(op)
            if hits is None:
                raise RuntimeError(f"{src}: could not disassemble the gfx950 code object (llvm-objdump): the store-hazard gate cannot vouch for this object")
            with open(rp, "w") as f:
                json.dump({"wide_stores_with_register_soffset": len(hits), "first": hits[:3]}, f)
        with open(rp) as f:
            rep = json.load(f)
        if rep["wide_stores_with_register_soffset"] and variant is None and not os.environ.get("DSVC_ALLOW_SOFFSET_STORES"):
            raise RuntimeError(f"{src}: {rep['wide_stores_with_register_soffset']} wide buffer store(s) with a register scalar offset, e.g. `{rep['first'][0]}` "
                               "-- see build.wide_stores_with_register_soffset")
    # No kernel of this library may touch scratch memory: a kernel that does has lost an array or its accumulators to the stack (once: the
    # generic conv's slab-epilogue variants, 3x slower, unnoticed by every parity test).  DSVC_ALLOW_SCRATCH=1 turns the check off.
    usage_all = resource_usage(variant)
    missing = [src for src in SOURCES if src.endswith(".hip") and not os.path.exists(os.path.join(obj_dir, os.path.splitext(src)[0] + ".usage.json"))]
    if missing or not usage_all:
        raise RuntimeError(f"no kernel resource report for {missing or SOURCES}: rebuild with --force")
    spilled = {k: v["scratch"] for k, v in usage_all.items() if v["scratch"] > SCRATCH_ALLOWED.get(k, 0)}
    if spilled and variant is not None:          # diagnostic builds (probe registers) may spill a little: report only
        print("note: kernels using scratch memory in this diagnostic build: " + ", ".join(f"{k[28:70]}: {v}" for k, v in sorted(spilled.items())), flush=True)
    elif spilled and not os.environ.get("DSVC_ALLOW_SCRATCH"):
        raise RuntimeError("kernels using scratch memory (bytes per lane): " + ", ".join(f"{k}: {v}" for k, v in sorted(spilled.items())))
    if relink or _stale(LIB, objs):      # (objects newer than the library: an earlier run compiled them and then stopped)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # python -m deepsvc_amd.build [--force] [--variant NAME -DFOO -DBAR ...]
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose=True, variant=var, defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
