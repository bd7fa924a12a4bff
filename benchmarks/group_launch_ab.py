"""A GEMV-path batch as ONE 2-D launch of its groups of four queries (option scan_group_launch = 1, the default) against
one launch per group (0): interleaved, wall clock over back-to-back device-pointer searches.
Usage: group_launch_ab.py <rows> <dtype> <kprime>   (f32 stores and kprime 0 keep every batch on the scan)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14_777
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
kp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dim = 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=max(2, int(np.sqrt(n))), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
ix = FlatIndex(dim, dtype, True)
ix.add_synth(0, n, spec)
ix.set_option("batch_mfma_min_b", 1 << 20)
st = torch.cuda.current_stream().cuda_stream
q = torch.empty((256, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 256, qspec, st)
torch.cuda.synchronize()
ix.set_option("timing_every", 65536)
reps = max(4, min(300, int(2e8 / n)))
out = {}
for b in (8, 16, 32, 64, 128, 256):
    if n > 2_000_000 and b > 32:
        break
    oi = torch.empty((b, 10), dtype=torch.int64, device="cuda"); od = torch.empty((b, 10), dtype=torch.float32, device="cuda")
    for rnd in range(3):
        for mode in (1, 0):
            ix.set_option("scan_group_launch", mode)
            for i in range(max(2, reps // 4)):
                ix.search_device(q.data_ptr(), b, 10, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                ix.search_device(q.data_ptr(), b, 10, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            out.setdefault(f"b{b}", {}).setdefault("grouped" if mode else "per_group", []).append(round(1e3 * (time.perf_counter() - t0) / reps, 4))
print(json.dumps({"rows": n, "dtype": dtype, "kprime": kp, "ms_per_batch": out}))
