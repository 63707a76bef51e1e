"""Host range coder throughput on realistic tables (64 Gaussian CDFs of get_scale_table) -- CPU only.
    python tools/bench_rans.py [n_symbols]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsvc_amd import ops
from deepsvc_amd.entropy_models import GaussianConditional
from deepsvc_amd.modules import get_scale_table

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_400_000
gc = GaussianConditional(None)
gc.update_scale_table(get_scale_table())
tables = gc.host_tables()
rng = np.random.default_rng(0)
table = get_scale_table().numpy()
# sigma indexes concentrated on the low rows like a real latent (most sigmas small), symbols ~ N(0, sigma)
idx = np.clip(np.rint(rng.normal(26.0, 9.0, n)), 0, 63).astype(np.int32)     # ~4 bit/symbol like the calibrated synthetic weights
sym = np.rint(rng.normal(0.0, 1.0, n) * table[idx]).astype(np.int32)
for rep in range(3):
    enc = ops.RansEncoder()
    t0 = time.perf_counter()
    enc.encode_with_indexes(sym, idx, tables)
    t1 = time.perf_counter()
    data = enc.flush()
    t2 = time.perf_counter()
    dec = ops.RansDecoder(data)
    t3 = time.perf_counter()
    out = dec.decode_stream(idx, tables)
    t4 = time.perf_counter()
    assert np.array_equal(out, sym)
    print(f"n={n} {len(data)*8/n:.2f} bit/sym: encode pass1 {n/(t1-t0)/1e6:6.1f} + flush {n/(t2-t1)/1e6:6.1f} -> {n/(t2-t0)/1e6:6.1f} Msym/s | "
          f"decode {n/(t4-t3)/1e6:6.1f} Msym/s", flush=True)
