"""CPU tests of the harness: GOP sharding, metric gather across 2 processes (gloo), GOP loop state chaining.
The codec itself is replaced by a stub here (the product has no CPU compute path); the chaining contract checked is
the one of test_video.py:332-369: key frame resets (ref, feature), every P-frame hands (recon, feature) to the next."""
import os
import socket
import sys

import pytest
import torch

from deepsvc_amd import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_units_partition():
    for n, w in ((64, 8), (7, 2), (3, 4), (0, 2)):
        shards = [harness.shard_units(n, r, w) for r in range(w)]
        flat = sorted(u for s in shards for u in s)
        assert flat == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


class _StubCodec:
    """Records the call sequence; 'recon' = cur * 0.5 so PSNR is finite and state chaining is observable."""

    def __init__(self):
        self.calls = []

    def compress(self, ref, cur, sm, feature):
        self.calls.append(("c", float(ref.mean()), None if feature is None else float(feature.mean())))
        return ({"strings": [[b"\x00" * 10], [b"\x00" * 6]], "shape": (1, 1)}, {"strings": [[b"\x00" * 20], [b"\x00" * 4]], "shape": (1, 1)})

    def decompress(self, ref, mv, res, sm, feature):
        self.calls.append(("d", float(ref.mean()), None if feature is None else float(feature.mean())))
        recon = ref * 0.5
        return torch.full((1, 64, 1, 1), float(len(self.calls))), recon, recon, recon


def test_code_gop_chains_state_and_measures():
    frames = [torch.full((1, 3, 60, 100), 0.8) for _ in range(4)]     # un-padded 60x100 -> padded 64x128
    sms = [None] + [torch.zeros(1, 256, 16, 32) for _ in range(3)]
    codec = _StubCodec()
    rows = harness.code_gop(codec, frames, sms, unit=5)
    assert len(rows) == 3 and all(len(r) == len(harness.METRIC_COLUMNS) for r in rows)
    assert [r[0] for r in rows] == [5.0] * 3 and [r[1] for r in rows] == [1.0, 2.0, 3.0]
    # bpp uses the UN-padded pixel count (test_video.py:295,364-366): (10+6)*8/(60*100), (20+4)*8/(60*100)
    assert abs(rows[0][2] - 16 * 8 / 6000) < 1e-12 and abs(rows[0][3] - 24 * 8 / 6000) < 1e-12
    # first P-frame sees feature=None, later ones the feature returned by the previous decompress; ref halves each time
    assert codec.calls[0][2] is None and codec.calls[1][2] is None
    assert codec.calls[2][2] == 2.0 and codec.calls[4][2] == 4.0
    assert codec.calls[2][1] < codec.calls[0][1]


class _StubICodec:
    def compress(self, x):
        return {"strings": [[b"\x01" * 50], [b"\x02" * 5]], "shape": (1, 2)}

    def decompress(self, strings, shape):
        assert len(strings[0][0]) == 50 and tuple(shape) == (1, 2)
        return {"x_hat": torch.full((1, 3, 64, 128), 0.4)}


def test_code_gop_with_intra_coded_key_frame():
    """BASELINE.json config 3 plumbing (test_video.py:296-312): the key frame goes through the I-frame codec, its clamped
    reconstruction is the first reference, and it gets a row (frame 0) with its own bpp."""
    frames = [torch.full((1, 3, 60, 100), 0.8) for _ in range(3)]
    sms = [None] + [torch.zeros(1, 256, 16, 32) for _ in range(2)]
    codec = _StubCodec()
    rows = harness.code_gop(codec, frames, sms, unit=2, i_model=_StubICodec())
    assert [r[1] for r in rows] == [0.0, 1.0, 2.0]
    assert rows[0][2] == 0.0 and abs(rows[0][3] - 55 * 8 / 6000) < 1e-12
    assert abs(codec.calls[0][1] - 0.4) < 1e-6          # the P-frame codec's first reference is the I-frame reconstruction


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from deepsvc_amd import harness
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)
units = harness.shard_units(5, rank, world)          # 5 GOPs over 2 ranks: ranks get 3 and 2 -> ragged gather
rows = [[float(u), float(t), 0.1 * u, 0.2, 30.0 + u, 0, 0, 0.01, 0.02, 0.9, 0.8, 0.7] for u in units for t in (1, 2)]
table = harness.gather_rows(rows, torch.device("cpu"), world)
assert table.shape == (10, len(harness.METRIC_COLUMNS)), table.shape
assert sorted(set(table[:, 0].tolist())) == [0.0, 1.0, 2.0, 3.0, 4.0]
s = harness.summarise(table)
assert s["frames"] == 10 and abs(s["ppsnr"] - 32.0) < 1e-9
r = harness.reference_results(table)
assert len(r["ppsnr"]) == 5 and abs(r["ppsnr"][3] - 33.0) < 1e-9 and abs(r["mvbpp"][2] - 0.2) < 1e-12
print("rank", rank, "ok", flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_gather_gloo(tmp_path, world):
    """2 ranks (3 + 2 GOPs: a ragged gather) and the 8 ranks of the driver's scaling run (5 GOPs: three ranks have nothing to code)."""
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_bench_launcher_refuses_inconsistent_rank_counts():
    """`bench.py --gpus N` launches its own ranks; it must fail loudly -- not silently measure one GPU -- when N GPUs are
    not visible (no rehearsal flag) and when it is started under a launcher whose WORLD_SIZE differs from --gpus."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSVC_BENCH_REHEARSAL")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "visible" in r.stderr, r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr, r.stderr


def test_reference_json_schema_and_averaging(tmp_path):
    """harness.reference_results / write_reference_json reproduce the reference's per-run JSON (test_video.py:442-460): all 26
    keys in its order, one list entry per sequence, averaged the way its loops append (:312-391): I-frame and P-frame rows,
    semantic-layer rate / time added where the reference adds them."""
    import json
    import numpy as np
    c = harness.METRIC_COLUMNS
    row = lambda **kw: [float(kw.get(n, 0.0)) for n in c]
    table = torch.tensor([
        row(unit=0, frame=0, bpp_res=2.0, psnr=30, msssim=0.9, t_enc=0.5, t_dec=0.25),
        row(unit=0, frame=1, bpp_mv=0.1, bpp_res=0.3, psnr=28, psnr_mc=25, psnr_warp=24, msssim=0.8, msssim_mc=0.7, msssim_warp=0.6, t_enc=0.2, t_dec=0.1),
        row(unit=0, frame=2, bpp_mv=0.3, bpp_res=0.5, psnr=26, psnr_mc=23, psnr_warp=22, msssim=0.6, msssim_mc=0.5, msssim_warp=0.4, t_enc=0.4, t_dec=0.3),
        row(unit=1, frame=1, bpp_mv=0.2, bpp_res=0.2, psnr=20, psnr_mc=19, psnr_warp=18, msssim=0.5, msssim_mc=0.4, msssim_warp=0.3, t_enc=1.0, t_dec=2.0),
    ], dtype=torch.float64)
    sm = {0.0: [(0.05, 0.01, 0.02), (0.07, 0.03, 0.04)]}
    out = harness.write_reference_json(str(tmp_path / "r.json"), table, "UVG_MSE_80", sm_rows=sm)
    on_disk = json.load(open(tmp_path / "r.json"))
    assert on_disk == json.loads(json.dumps(out)) and on_disk["name"] == "UVG_MSE_80" and on_disk["description"] == "Inference (ans)"
    r = on_disk["results"]
    assert tuple(r.keys()) == harness.REFERENCE_JSON_KEYS and len(harness.REFERENCE_JSON_KEYS) == 26
    assert all(len(v) == 2 for v in r.values())
    ap = lambda a, b: abs(a - b) < 1e-12
    assert ap(r["ipsnr"][0], 30) and ap(r["ibpp"][0], 2.0) and ap(r["imsssim"][0], 0.9) and ap(r["ienc"][0], 0.5) and ap(r["idec"][0], 0.25)
    assert ap(r["ppsnr"][0], 27) and ap(r["mcpsnr"][0], 24) and ap(r["warppsnr"][0], 23)
    assert ap(r["mvbpp"][0], 0.2) and ap(r["resbpp"][0], 0.4) and ap(r["smbpp"][0], 0.06)
    assert ap(r["pbpp"][0], (0.1 + 0.3 + 0.05 + 0.3 + 0.5 + 0.07) / 2)                       # mv + res + sm (test_video.py:366)
    assert ap(r["bpp"][0], (2.0 + 0.45 + 0.87) / 3) and ap(r["porposed_pbpp_wosm"][0], (2.0 + 0.4 + 0.8) / 3)
    assert ap(r["bpp2l"][0], (2.0 + 0.15 + 0.37) / 3)                                       # I bpp | mv + sm (:318,384)
    assert ap(r["psnr"][0], 28) and ap(r["msssim"][0], (0.9 + 0.8 + 0.6) / 3) and ap(r["pmsssim"][0], 0.7)
    assert ap(r["mcmsssim"][0], 0.6) and ap(r["warmsssim"][0], 0.5)
    assert ap(r["pent"][0], (0.21 + 0.43) / 2) and ap(r["pdec"][0], (0.12 + 0.34) / 2)      # + semantic-layer times (:388-391)
    assert ap(r["ent"][0], (0.5 + 0.21 + 0.43) / 3) and ap(r["dec"][0], (0.25 + 0.12 + 0.34) / 3)
    assert ap(r["sment"][0], 0.02) and ap(r["smdec"][0], 0.03)
    # a sequence without a coded key frame (configs 1-2) and without a semantic layer
    assert r["ipsnr"][1] == 0.0 and r["ibpp"][1] == 0.0 and ap(r["pbpp"][1], 0.4) and ap(r["bpp"][1], 0.4) and ap(r["smbpp"][1], 0.0) and ap(r["pent"][1], 1.0)
