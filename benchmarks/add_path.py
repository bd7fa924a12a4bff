#!/usr/bin/env python3
"""The add path: rows/s and HBM GB/s of vdb_add_dev (device-resident embeddings -> normalize_rows_kernel ->
store) and the PCIe-inclusive rate of vdb_add (host rows).  Algorithmic bytes per row: read dim*4 (raw fp32) + write dim*4
(fp32 rows) [+ dim*2 (bf16 rows)].
  python3 benchmarks/add_path.py [--rows 2000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd import FlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=768)
    args = ap.parse_args()
    n, dim = args.rows, args.dim
    x = torch.randn((n, dim), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    out = {"rows": n, "dim": dim}
    for name, dtype, keep, bytes_per_row in (("f32", "f32", True, dim * 8), ("bf16+f32", "bf16", True, dim * 10), ("bf16", "bf16", False, dim * 6)):
        best = 1e9
        for rep in range(3):
            ix = FlatIndex(dim, dtype, keep)
            ix.reserve(n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.add_device(x.data_ptr(), n)          # synchronises its stream before returning
            dt = time.perf_counter() - t0
            best = min(best, dt)
            ix.close()
        out[name] = {"add_dev_s": best, "rows_per_s": n / best, "GBps": n * bytes_per_row / best / 1e9, "frac_of_8TBps": n * bytes_per_row / best / 8e12}
        print(name, json.dumps(out[name]), flush=True)
    nh = min(n, 500_000)
    xh = x[:nh].cpu().numpy()
    ix = FlatIndex(dim, "bf16", True)
    ix.reserve(nh)
    t0 = time.perf_counter()
    ix.add(xh)
    dt = time.perf_counter() - t0
    out["host_add_bf16+f32"] = {"rows": nh, "s": dt, "rows_per_s": nh / dt, "host_GBps": nh * dim * 4 / dt / 1e9}
    print("host", json.dumps(out["host_add_bf16+f32"]), flush=True)
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
