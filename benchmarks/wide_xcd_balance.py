"""Do the XCDs finish a pass of the tiled MFMA kernel together?  In-kernel cycle stamps (option mfma_ablate = 8: the
kernel's diagnostic build; its answers are invalid) of the LAST pass of a 1024-query search: mean cycles per wave, by XCD
(blocks b with b & 7 == x).  Usage: wide_xcd_balance.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, b = 768, 1024
spec = SynthSpec(seed=1234, kind=1, n_clusters=int(np.sqrt(n)), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
st = torch.cuda.current_stream().cuda_stream
ix = FlatIndex(dim, "bf16", True)
ix.add_synth(0, n, spec)
q = torch.empty((b, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, b, qspec, st)
oi = torch.empty((b, 10), dtype=torch.int64, device="cuda"); od = torch.empty((b, 10), dtype=torch.float32, device="cuda")
for i in range(10):
    ix.search_device(q.data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), st)
torch.cuda.synchronize()
ix.set_option("mfma_ablate", 8)
res = []
for rep in range(5):
    for i in range(2):
        ix.search_device(q.data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    res.append([ix.get_stat(f"mfma_prof_xcd_{x}") for x in range(8)])
res = np.array(res, dtype=np.float64)
rel = res / res.mean(axis=1, keepdims=True) - 1.0
print(json.dumps({"rows": n, "batch": b, "mean_cycles_per_wave_by_xcd_last_rep": res[-1].tolist(),
                  "relative_to_mean_by_xcd_per_rep_percent": np.round(100 * rel, 2).tolist(),
                  "max_over_mean_percent_per_rep": np.round(100 * rel.max(axis=1), 2).tolist()}))
