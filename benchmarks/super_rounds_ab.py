"""The fused scan's tile partition: block groups on slices of super-rounds — one window over the corpus — (option
scan_super_rounds = 1, default) against one contiguous range per group (0, round 3's first form).  Store after store in
ONE process, so that the stores land in different places of the heap; per store the two forms alternate.
Usage: super_rounds_ab.py <rows> <stores> [alloc_f32_first]   (1: the fp32 copy in front of the scanned rows — at 20M rows that
puts them in a slow place, profiles/r03_placement.txt)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
stores = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f32_first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dim = 768
reps = max(100, int(0.5 / (n * 1536 / 7e12)))
out = []
for sidx in range(stores):
    spec = SynthSpec(seed=1234 + sidx, kind=1, n_clusters=int(np.sqrt(n)), noise=0.3)
    qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234 + sidx, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("alloc_f32_first", f32_first)
    ix.add_synth(0, n, spec)
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, st)
    oi = torch.empty((1, 10), dtype=torch.int64, device="cuda"); od = torch.empty((1, 10), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ix.set_option("timing_every", 65536)
    res = {"store": sidx, "super_rounds": [], "ranges": []}
    for rnd in range(2):
        for mode in (1, 0):
            ix.set_option("scan_super_rounds", mode)
            for i in range(2 * reps):
                ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            res["super_rounds" if mode else "ranges"].append(round(1e3 * (time.perf_counter() - t0) / reps, 4))
    ix.set_option("scan_super_rounds", 1)
    res["probe_GBps"] = round(ix.get_stat("probe_hbm_read_mbps") / 1e3, 1)
    out.append(res)
    ix.close()
    del ix
print(json.dumps({"rows": n, "alloc_f32_first": f32_first, "ms_per_step": out}))
