#!/usr/bin/env python3
"""Two box-side reference points for DESIGN §4.2 / §8 (one GPU):
  (1) what the vendor GEMM library (hipBLASLt/rocBLAS behind torch.matmul) reaches in bf16 on RANDOM operands on this
      box — the search's own shape (1024 x 768 against chunks of the corpus, scores written out) and a square 8192^3 —
      next to the same GEMMs on all-zero operands (the chip is power-limited: zeros flatter the ceiling);
  (2) whether replaying a batch-1 search step from a captured HIP graph beats launching its kernels one by one.
  python3 benchmarks/ceiling_probe.py [--rows N]"""
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


def gemm_rate(m, n, k, zeros, iters=10):
    mk = (lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")) if zeros else \
         (lambda *s: (torch.randn(*s, device="cuda") / np.sqrt(k)).to(torch.bfloat16))
    a, b = mk(m, k), mk(n, k)
    c = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"ms": 1e3 * dt, "tflops": 2.0 * m * n * k / dt / 1e12}


def graph_vs_launches(rows, iters):
    dim, k, kp = 768, 10, 40
    spec = SynthSpec(seed=1234, kind=1, n_clusters=max(1, int(np.sqrt(rows))), noise=0.3)
    qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=rows, base_n_clusters=spec.n_clusters, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, rows, spec)
    q = torch.empty((1, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, 1, qspec, torch.cuda.current_stream().cuda_stream)
    oi = torch.empty((1, k), dtype=torch.int64, device="cuda")
    od = torch.empty((1, k), dtype=torch.float32, device="cuda")
    oi2, od2 = torch.empty_like(oi), torch.empty_like(od)
    ix.set_option("timing_every", 65536)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = side.cuda_stream
        for _ in range(5):                      # workspaces exist, the events of call 0 are behind us
            ix.search_device(q.data_ptr(), 1, k, kp, oi.data_ptr(), od.data_ptr(), st)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ix.search_device(q.data_ptr(), 1, k, kp, oi.data_ptr(), od.data_ptr(), st)
        side.synchronize()
        plain = (time.perf_counter() - t0) / iters
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ix.search_device(q.data_ptr(), 1, k, kp, oi2.data_ptr(), od2.data_ptr(), side.cuda_stream)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / iters
    same = bool(torch.equal(oi, oi2) and torch.equal(od, od2))
    ix.close()
    return {"rows": rows, "launches_ms": 1e3 * plain, "graph_replay_ms": 1e3 * graph, "same_result": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_250_000)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--skip-graph", action="store_true")
    args = ap.parse_args()
    out = {"gemm": {}}
    for name, (m, n, k) in {"search_shape_1024x1Mx768": (1024, 1 << 20, 768), "search_shape_4096x262144x768": (4096, 1 << 18, 768),
                            "square_8192": (8192, 8192, 8192)}.items():
        out["gemm"][name] = {"random": gemm_rate(m, n, k, False), "zeros": gemm_rate(m, n, k, True)}
        print(name, json.dumps(out["gemm"][name]), flush=True)
    if not args.skip_graph:
        out["graph"] = [graph_vs_launches(args.rows, args.iters)]
        print(json.dumps(out["graph"][-1]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
