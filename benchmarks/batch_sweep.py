#!/usr/bin/env python3
"""Cost of one search step against the batch size (10M x 768 bf16, k=10, k'=40) on one GPU: which kernel family the
engine picks, the scan phase from HIP events, and the same batch forced onto the other families.
  python3 benchmarks/batch_sweep.py [--rows N] [--batches 1,2,4,...] [--force]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd import FlatIndex, SynthSpec  # noqa: E402


def timed(ix, q, b, k, kp, iters):
    oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    od = torch.empty((b, k), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        ix.search_device(q.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    ix.set_option("timing_ring", iters)
    t0 = time.perf_counter()
    for _ in range(iters):
        ix.search_device(q.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    ts = ix.timing_stats()
    return {"path": ix.get_stat("last_path"), "ms_per_step": 1e3 * dt, "scan_ms": ts["scan_ms_sum"] / ts["n"],
            "finish_ms": ts["merge_ms_sum"] / ts["n"], "overflow": ix.get_stat("mfma_last_overflow")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--batches", default="1,2,3,4,8,16,17,32,33,48,64,65,128,256,1024")
    ap.add_argument("--force", action="store_true", help="also time each batch on the families the cost model did not pick")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--option", action="append", default=[])
    args = ap.parse_args()
    n, dim, k, kp = args.rows, 768, 10, 40
    spec = SynthSpec(seed=1234, kind=1, n_clusters=max(1, int(np.sqrt(n))), noise=0.3)
    qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    for ov in args.option:
        name, _, val = ov.partition("=")
        ix.set_option(name, int(val))
    out = {"rows": n, "floor_ms_at_8TBps": n * dim * 2 / 8e12 * 1e3, "steps": {}}
    for b in [int(x) for x in args.batches.split(",")]:
        q = torch.empty((b, dim), dtype=torch.float32, device="cuda")
        ix.synth_rows_device(q.data_ptr(), 0, b, qspec, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        iters = args.iters if b <= 256 else max(5, args.iters // 2)
        r = {"picked": timed(ix, q, b, k, kp, iters)}
        if args.force:
            ix.set_option("mfma_cost_model", 0)
            if b <= 64:
                ix.set_option("batch_mfma_min_b", 1 << 20); r["gemv"] = timed(ix, q, b, k, kp, max(3, iters // 4)); ix.set_option("batch_mfma_min_b", 2)
                ix.set_option("mfma_variant", 6); ix.set_option("stream_nt_loads", 0); r["stream"] = timed(ix, q, b, k, kp, iters)
                ix.set_option("stream_nt_loads", 1); r["stream_nt"] = timed(ix, q, b, k, kp, iters)   # (1 is the engine's default: it stays)
            if b >= 2:
                ix.set_option("mfma_variant", 5); r["tiled"] = timed(ix, q, b, k, kp, iters)
            ix.set_option("mfma_variant", 6)
            ix.set_option("mfma_cost_model", 1)
        out["steps"][str(b)] = r
        print(b, json.dumps(r), flush=True)
    print(json.dumps(out))
    ix.close()


if __name__ == "__main__":
    main()
