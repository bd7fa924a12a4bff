"""Why does benchmarks/configs.py time a 12.5M-row batch-1 step at 2.90 ms when bench.py --rows 12500000 shows 2.68-2.79?
Same store, same box: per-call timing events on / off, the two query/corpus distributions."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dim = 768
out = {}
for name, ncl, qseed in (("bench_1st", int(np.sqrt(n)), 1235), ("configs_2nd", 10000, 7), ("bench_3rd", int(np.sqrt(n)), 1235), ("configs_4th", 10000, 7)):
    spec = SynthSpec(seed=1234, kind=1, n_clusters=ncl, noise=0.3)
    qspec = SynthSpec(seed=qseed, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=ncl, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, st)
    oi = torch.empty((1, 10), dtype=torch.int64, device="cuda"); od = torch.empty((1, 10), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for every in (65536, 65536):
        ix.set_option("timing_every", every)
        for nq in (64,):
            for i in range(300):
                ix.search_device(q[i % nq].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                ix.search_device(q[i % nq].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            out.setdefault(name, {}).setdefault(f"events_every_{every}_queries_{nq}", []).append(round(1e3 * (time.perf_counter() - t0) / 200, 4))
    ix.close()
print(json.dumps({"rows": n, "ms_per_step": out}))
