#!/usr/bin/env python3
"""A/B of one engine option inside ONE process, arms interleaved (boxes differ by +-2 %, runs on one box by ~1 %):
  python3 benchmarks/ab_option.py --option mfma_skip_sample --values 0,1 --batch 1024 [--rows N] [--rounds 6] [--iters 10]
  python3 benchmarks/ab_option.py --arms "mfma_split_x1000=500,mfma_append_cap=16384;mfma_split_x1000=650,mfma_append_cap=32768" --batch 32"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", default=None)
    ap.add_argument("--arms", default=None, help="instead of --option/--values: arms separated by ';', each a comma list of NAME=VALUE")
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    args = ap.parse_args()
    n, dim, k, kp, b = args.rows, 768, 10, 40, args.batch
    spec = SynthSpec(seed=1234, kind=1, n_clusters=max(1, int(np.sqrt(n))), noise=0.3)
    qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    for ov in args.set:
        name, _, val = ov.partition("=")
        ix.set_option(name, int(val))
    q = torch.empty((b, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, b, qspec, torch.cuda.current_stream().cuda_stream)
    oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    od = torch.empty((b, k), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if args.arms:
        vals = [a.strip() for a in args.arms.split(";") if a.strip()]
    else:
        vals = [int(v) for v in args.values.split(",")]
    res = {v: [] for v in vals}
    ref = None
    for r in range(args.rounds + 1):
        for v in vals:
            if args.arms:
                for kv in v.split(","):
                    name, _, val = kv.partition("=")
                    ix.set_option(name.strip(), int(val))
            else:
                ix.set_option(args.option, v)
            ix.search_device(q.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            if ref is None:
                ref = (oi.clone(), od.clone())
            assert torch.equal(oi, ref[0]) and torch.equal(od, ref[1]), f"{args.option}={v} changes the answer"
            t0 = time.perf_counter()
            for _ in range(args.iters):
                ix.search_device(q.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            if r:                      # round 0 warms up
                res[v].append(1e3 * (time.perf_counter() - t0) / args.iters)
    print(json.dumps({"option": args.option, "batch": b, "rows": n,
                      "ms_per_step": {str(v): {"median": float(np.median(x)), "min": float(np.min(x)), "max": float(np.max(x))} for v, x in res.items()}}))
    ix.close()


if __name__ == "__main__":
    main()
