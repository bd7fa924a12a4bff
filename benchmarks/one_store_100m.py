"""All of C4's 100M x 768 bf16 rows as ONE store on one GPU (153.6 GB of its 288; no fp32 copy: candidates are re-scored
against the bf16 rows): batch-1 and 64-query steps, the planted neighbour of every query first.  Usage: one_store_100m.py [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
from vectordb_amd.synth import query_source_row
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dim = 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=10000, noise=0.3)
qspec = SynthSpec(seed=7, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=10000, base_noise=0.3)
st = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
ix = FlatIndex(dim, "bf16", False)
ix.add_synth(0, n, spec)
torch.cuda.synchronize()
out = {"rows": n, "generated_s": round(time.perf_counter() - t0, 2)}
ix.set_option("timing_every", 65536)
q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, st)
src = np.array([query_source_row(qspec, j) for j in range(1024)])
qb = torch.empty((1024, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(qb.data_ptr(), 0, 1024, qspec, st)
for b, reps in ((1, 40), (64, 10), (1024, 3)):
    oi = torch.empty((b, 10), dtype=torch.int64, device="cuda"); od = torch.empty((b, 10), dtype=torch.float32, device="cuda")
    for i in range(reps):
        ix.search_device((qb if b > 64 else q[(i % 64) if b == 1 else 0]).data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        ix.search_device((qb if b > 64 else q[(i % 64) if b == 1 else 0]).data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    out[f"b{b}"] = {"ms_per_step": round(ms, 3), "GBps": round(n * dim * 2 / ms / 1e6, 1), "TFLOPs": round(2.0 * b * n * dim / ms / 1e9, 1), "qps": round(1e3 * b / ms, 1), "last_path": ix.get_stat("last_path")}
oi = torch.empty((64, 10), dtype=torch.int64, device="cuda"); od = torch.empty((64, 10), dtype=torch.float32, device="cuda")
ix.search_device(q.data_ptr(), 64, 10, 40, oi.data_ptr(), od.data_ptr(), st)
torch.cuda.synchronize()
got = oi.cpu().numpy()
out["planted_neighbour_first_64_of_64"] = bool(np.array_equal(got[:, 0], src[:64]))
ob = torch.empty((1024, 10), dtype=torch.int64, device="cuda"); obd = torch.empty((1024, 10), dtype=torch.float32, device="cuda")
ix.search_device(qb.data_ptr(), 1024, 10, 40, ob.data_ptr(), obd.data_ptr(), st)
torch.cuda.synchronize()
gb = ob.cpu().numpy()
out["planted_neighbour_first_1024_of_1024"] = bool(np.array_equal(gb[:, 0], src))
out["batch1024_equals_batch64_on_the_shared_queries"] = bool(np.array_equal(gb[:64], got))
out["mfma_overflow_fallback"] = ix.get_stat("mfma_last_overflow")
one = torch.empty((1, 10), dtype=torch.int64, device="cuda"); oned = torch.empty((1, 10), dtype=torch.float32, device="cuda")
ok1 = True
for j in range(8):
    ix.search_device(q[j].data_ptr(), 1, 10, 40, one.data_ptr(), oned.data_ptr(), st)
    torch.cuda.synchronize()
    ok1 = ok1 and int(one[0, 0]) == int(src[j]) and bool(np.array_equal(one.cpu().numpy()[0], got[j]))
out["batch1_equals_batch64_and_planted_first"] = ok1
print(json.dumps(out))
