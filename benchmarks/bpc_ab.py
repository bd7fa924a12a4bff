"""Scan blocks per CU (option scan_blocks_per_cu) on the super-round partition: values in turn on one store, three rounds.
Usage: bpc_ab.py <rows>"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=int(np.sqrt(n)), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
st = torch.cuda.current_stream().cuda_stream
ix = FlatIndex(dim, "bf16", True)
ix.add_synth(0, n, spec)
ix.set_option("timing_every", 65536)
q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, st)
oi = torch.empty((1, 10), dtype=torch.int64, device="cuda"); od = torch.empty((1, 10), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
reps = max(100, int(0.4 / (n * 1536 / 7e12)))
out = {"rows": n, "ms_per_step_by_blocks_per_cu": {}}
for rnd in range(3):
    for bpc in (2, 3, 4, 1):
        ix.set_option("scan_blocks_per_cu", bpc)
        for i in range(3 * reps):      # the shares of the new grid's groups tune themselves first
            ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        out["ms_per_step_by_blocks_per_cu"].setdefault(str(bpc), []).append(round(1e3 * (time.perf_counter() - t0) / reps, 4))
print(json.dumps(out))
