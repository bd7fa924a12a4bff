"""Masked (where=) scan at 10M x 768 bf16: time by eligible fraction."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import FlatIndex, SynthSpec
n = 10_000_000
ix = FlatIndex(768, "bf16", True)
ix.add_synth(0, n, SynthSpec(seed=1234, kind=1, n_clusters=3162, noise=0.3))
rng = np.random.default_rng(0)
q = rng.standard_normal((4, 768)).astype(np.float32)
for frac in (1.0, 0.5, 0.1, 0.01, 0.0001):
    mask = rng.random(n) < frac
    for b in (1, 4):
        ix.search_masked(q[:b], 10, mask, 40)
        t = time.perf_counter()
        for _ in range(10): ix.search_masked(q[:b], 10, mask, 40)
        dt = (time.perf_counter() - t) / 10
        scan = ix.last_timing()["scan_ms"]
        print(f"eligible {frac:7.4f} b={b}: host-to-host {dt*1e3:.3f} ms (bitmap pack+upload included), scan kernel {scan:.3f} ms", flush=True)
t = time.perf_counter()
for _ in range(10): ix.search(q[:1], 10, 40)
print("unmasked b=1 host-to-host", (time.perf_counter() - t) / 10 * 1e3, "ms")
