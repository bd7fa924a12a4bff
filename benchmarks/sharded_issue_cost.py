"""Host-side issue cost of vdb_search_dev on a sharded handle, measured on ONE GPU: device_ids =
[0] * g with tiny shards, so that the device work is negligible and the host's runtime calls are what is timed.
  python benchmarks/sharded_issue_cost.py [--shards 8] [--rows-per-shard 4096] [--calls 1000] [--threads 0|1]
Prints one JSON line: host time inside the library per call (its own steady clock around the whole call), wall time per
back-to-back call without a sync, and per call with a sync."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import FlatIndex, SynthSpec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--rows-per-shard", type=int, default=4096)
ap.add_argument("--calls", type=int, default=1000)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--burst", type=int, default=8, help="back-to-back calls between device syncs")
ap.add_argument("--threads", type=int, default=-1, help="option shard_threads (-1 = library default)")
ap.add_argument("--timing-every", type=int, default=4)
args = ap.parse_args()

dim, k, kp = 768, 10, 40
g, n = args.shards, args.shards * args.rows_per_shard
spec = SynthSpec(seed=3, kind=1, n_clusters=64, noise=0.3)
out = {"shards": g, "rows_per_shard": args.rows_per_shard, "batch": args.batch, "calls": args.calls, "burst": args.burst}
for name, devices in (("sharded", [0] * g), ("single_store", None)):
    ix = FlatIndex(dim, "bf16", True, devices=devices)
    ix.reserve(n)
    ix.add_synth(0, n, spec)
    if devices is not None and args.threads >= 0:
        ix.set_option("shard_threads", args.threads)
    ix.set_option("timing_every", args.timing_every)
    q = torch.randn(args.batch, dim, device="cuda")
    oi = torch.empty((args.batch, k), dtype=torch.int64, device="cuda")
    od = torch.empty((args.batch, k), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(50):
        ix.search_device(q.data_ptr(), args.batch, k, kp, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    if devices is not None:
        ix.get_stat("sharded_issue_ns_avg")          # reset
    # bursts of a few calls with a device sync between them: a long unsynchronised run fills the hardware queues and
    # the host is then throttled to the DEVICE's rate (eight chained tiny kernels per call), which is not its own cost
    t_issue, done = 0.0, 0
    t00 = time.perf_counter()
    while done < args.calls:
        t0 = time.perf_counter()
        for _ in range(args.burst):
            ix.search_device(q.data_ptr(), args.batch, k, kp, oi.data_ptr(), od.data_ptr(), st)
        t_issue += time.perf_counter() - t0
        done += args.burst
        torch.cuda.synchronize()
    t_all = time.perf_counter() - t00
    args_calls = done
    lib_stats = {}
    if devices is not None:
        names = ["ev_q record", "set-device + ev_q wait", "query copy + prep", "search_core (launches, timing events)", "shard_done record (+ result copies)",
                 "main-stream waits", "merge launch"]
        lib_stats["library_host_us_per_call_by_phase"] = {nm: ix.get_stat(f"sharded_issue_phase_{i}") / 1e3 for i, nm in enumerate(names)}
        lib_stats["library_host_us_per_call"] = ix.get_stat("sharded_issue_ns_avg") / 1e3
        lib_stats["library_host_us_per_shard"] = lib_stats["library_host_us_per_call"] / g
    lat = []
    for _ in range(min(args.calls, 300)):
        t1 = time.perf_counter()
        ix.search_device(q.data_ptr(), args.batch, k, kp, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    r = {"host_us_per_call_no_sync": 1e6 * t_issue / args_calls, "us_per_call_incl_drain": 1e6 * t_all / args_calls,
         "sync_latency_us_median": 1e6 * float(np.median(lat))}
    r.update(lib_stats)
    out[name] = r
    ix.close()
print(json.dumps(out))
