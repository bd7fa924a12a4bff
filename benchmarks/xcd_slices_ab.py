"""Blocks of one XCD on adjacent tiles (options stream_xcd_slices / scan_xcd_slices = 1, default) against block-by-block
round-robin (0): the streaming kernel (batches 3..64, bf16 + rerank) and the unfused scan (fp32 store, 4 and 8 queries),
forms alternating on one store.  Usage: xcd_slices_ab.py <rows>"""
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
out = {"rows": n}
def run(ix, opt, b, kp, reps):
    q = torch.empty((b, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, b, qspec, st)
    oi = torch.empty((b, 10), dtype=torch.int64, device="cuda"); od = torch.empty((b, 10), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    res = {"slices": [], "round_robin": []}
    for rnd in range(3):
        for mode in (1, 0):
            ix.set_option(opt, mode)
            for i in range(reps):
                ix.search_device(q.data_ptr(), b, 10, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                ix.search_device(q.data_ptr(), b, 10, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            res["slices" if mode else "round_robin"].append(round(1e3 * (time.perf_counter() - t0) / reps, 4))
    ix.set_option(opt, 1)
    return res
reps = max(30, int(0.4 / (n * 1536 / 7e12)))
ix = FlatIndex(dim, "bf16", True)
ix.add_synth(0, n, spec)
ix.set_option("timing_every", 65536)
for b in (8, 32, 64):
    out[f"stream_b{b}"] = run(ix, "stream_xcd_slices", b, 40, reps)
ix.set_option("batch_mfma_min_b", 1 << 20); ix.set_option("fuse_small_batch", 0)
for b in (1, 4):
    out[f"unfused_scan_bf16_b{b}"] = run(ix, "scan_xcd_slices", b, 40, reps)
ix.close()
if n <= 4_000_000:
    ix = FlatIndex(dim, "f32", True)
    ix.add_synth(0, n, spec)
    ix.set_option("timing_every", 65536)
    for b in (8, 32):
        out[f"scan_f32_b{b}"] = run(ix, "scan_xcd_slices", b, 0, max(10, reps // 4))
    ix.close()
print(json.dumps(out))
