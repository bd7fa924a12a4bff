"""Does the stream a search is issued on change the scan's rate?  200 back-to-back batch-1 searches (wall clock) on torch's
default stream, a torch-created stream, and the handle's OWN stream (hipStreamCreateWithFlags(hipStreamNonBlocking) at vdb_create)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=int(np.sqrt(n)), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
ix = FlatIndex(dim, "bf16", True)
ix.add_synth(0, n, spec)
q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, torch.cuda.current_stream().cuda_stream)
oi = torch.empty((1, 10), dtype=torch.int64, device="cuda"); od = torch.empty((1, 10), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ix.set_option("timing_every", 65536)
created = torch.cuda.Stream()
streams = {"torch default": torch.cuda.current_stream().cuda_stream, "torch created": created.cuda_stream, "handle's own": ix.get_stat("own_stream_handle")}
out = {}
for rnd in range(3):
    for name, st in streams.items():
        for i in range(150):
            ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        out.setdefault(name, []).append(round(1e3 * (time.perf_counter() - t0) / 200, 4))
print(json.dumps({"rows": n, "ms_per_step": out}))
