"""Allocation order of a store's two arrays (option alloc_f32_first) against the batch-1 step: ONE store per process (a
fresh heap), the order given on the command line.  Usage: alloc_order_ab.py <rows> <0|1>"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]); f32_first = int(sys.argv[2])
dim = 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=int(np.sqrt(n)), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
ix = FlatIndex(dim, "bf16", True)
ix.set_option("alloc_f32_first", f32_first)
ix.add_synth(0, n, spec)
st = torch.cuda.current_stream().cuda_stream
q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, st)
oi = torch.empty((1024, 10), dtype=torch.int64, device="cuda"); od = torch.empty((1024, 10), dtype=torch.float32, device="cuda")
qb = torch.empty((1024, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(qb.data_ptr(), 0, 1024, qspec, st)
torch.cuda.synchronize()
ix.set_option("timing_every", 65536)
out = {"rows": n, "alloc_f32_first": f32_first, "b1_ms": [], "b1024_ms": []}
reps = max(50, int(0.6 / (n * 1536 / 7e12)))
for rnd in range(3):
    for i in range(reps):
        ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    out["b1_ms"].append(round(1e3 * (time.perf_counter() - t0) / reps, 4))
for rnd in range(2):
    for i in range(6):
        ix.search_device(qb.data_ptr(), 1024, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        ix.search_device(qb.data_ptr(), 1024, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    out["b1024_ms"].append(round(1e3 * (time.perf_counter() - t0) / 10, 4))
out["probe_GBps"] = round(ix.get_stat("probe_hbm_read_mbps") / 1e3, 1)
print(json.dumps(out))
