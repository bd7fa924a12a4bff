#!/usr/bin/env python3
"""Single-GPU measurements of BASELINE.json's other configurations (C1, C2, one C4
shard, C5) — supporting numbers for BASELINE.md's table; bench.py remains the
contract line (C3).  Writes one JSON document to stdout."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd import FlatIndex, SynthSpec, EphemeralClient  # noqa: E402


def timed_search_dev(ix, q_dev, b, k, kp, iters, warm=3, settle_s=0.8):
    oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    od = torch.empty((b, k), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # settle first, as bench.py does: a loop's first ~0.4 s runs 2-3 % slow on an idle device, and the fused scan's
    # per-XCD tile shares tune themselves over the first launches
    t_settle = time.perf_counter()
    n_warm = 0
    while n_warm < warm or time.perf_counter() - t_settle < settle_s:
        ix.search_device(q_dev.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
        n_warm += 1
        if n_warm % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ix.set_option("timing_ring", iters)
    t0 = time.perf_counter()
    for _ in range(iters):
        ix.search_device(q_dev.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = ix.timing_stats()
    return dt / iters, ts["scan_ms_sum"] / ts["n"], ts["merge_ms_sum"] / ts["n"]


def synth_queries(ix, b, qspec):
    q = torch.empty((b, ix.dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(q.data_ptr(), 0, b, qspec, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return q


def main():
    out = {}
    which = set(sys.argv[1:]) or {"c1", "c2", "c3host", "c4", "c5"}
    dim = 768
    if "c1" in which:   # real-catalog size through the Chroma-shaped API (embeddings synthetic: SURVEY 8c)
        n = 14777
        spec = SynthSpec(seed=1234, kind=1, n_clusters=121, noise=0.3)
        gen = FlatIndex(dim, "f32", False)
        rows_dev = torch.empty((n, dim), dtype=torch.float32, device="cuda")
        gen.synth_rows_device(rows_dev.data_ptr(), 0, n, spec, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rows = rows_dev.cpu().numpy()
        gen.close()
        col = EphemeralClient().get_or_create_collection("actual_products", metadata={"hnsw:space": "cosine"})
        for lo in range(0, n, 5000):
            col.add(ids=[f"item_{i}" for i in range(lo, min(n, lo + 5000))], embeddings=rows[lo:lo + 5000],
                    metadatas=[{"product_description": f"p{i}", "product_code": str(i), "usda_code": "NOT_FOUND"} for i in range(lo, min(n, lo + 5000))])
        qs = rows[:200] + 0.01
        for q in qs[:10]:
            col.query(query_embeddings=[q.tolist()], n_results=30, include=["metadatas", "distances", "embeddings"])
        t0 = time.perf_counter()
        for q in qs:
            col.query(query_embeddings=[q.tolist()], n_results=30, include=["metadatas", "distances", "embeddings"])
        api_us = (time.perf_counter() - t0) / len(qs) * 1e6
        ix = col._index
        ix.set_option("timing_every", 1)      # the Collection switched the per-search timing events off; the engine-level rows below read them
        qd = torch.from_numpy(qs[:1].copy()).cuda()
        step, scan_ms, fin_ms = timed_search_dev(ix, qd, 1, 10, 0, 500, 20)
        q1 = qs[:1].copy()
        for _ in range(20):
            ix.search(q1, 10)
        t0 = time.perf_counter()
        for _ in range(500):
            ix.search(q1, 10)                 # vdb_search: host pointers in and out, synchronous (what Collection.query calls)
        host_us = (time.perf_counter() - t0) / 500 * 1e6
        out["C1_14777x768_f32"] = {"collection_query_k30_us": api_us, "engine_host_call_k10_us": host_us,
                                   "engine_k10_us_per_query": step * 1e6,
                                   "scan_kernel_us": scan_ms * 1e3, "finish_us": fin_ms * 1e3,
                                   "reference_published_s_per_query": 0.011}
    if "c2" in which:   # 1M x 768 fp32, batch 1
        n = 1_000_000
        spec = SynthSpec(seed=1234, kind=1, n_clusters=1000, noise=0.3)
        qspec = SynthSpec(seed=7, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=1000, base_noise=0.3)
        ix = FlatIndex(dim, "f32")
        ix.add_synth(0, n, spec)
        step, scan_ms, fin_ms = timed_search_dev(ix, synth_queries(ix, 1, qspec), 1, 10, 0, 300, 20)
        out["C2_1Mx768_f32_b1"] = {"qps": 1 / step, "scan_ms": scan_ms, "finish_ms": fin_ms,
                                   "scan_GBps": n * dim * 4 / scan_ms / 1e6, "frac_hbm_peak": n * dim * 4 / scan_ms / 1e6 / 8000,
                                   "roofline_qps": 2604}
        ix.close()
    if "c3host" in which:   # the bench workload through the HOST-pointer entry point (PCIe in the loop)
        n = 10_000_000
        spec = SynthSpec(seed=1234, kind=1, n_clusters=3162, noise=0.3)
        qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=3162, base_noise=0.3)
        ix = FlatIndex(dim, "bf16", True)
        ix.add_synth(0, n, spec)
        res = {}
        for b, iters in ((1, 100), (1024, 8)):
            q = synth_queries(ix, b, qspec).cpu().numpy()
            ix.search(q, 10, 40)
            t0 = time.perf_counter()
            for _ in range(iters):
                ix.search(q, 10, 40)          # vdb_search: H2D queries, search, D2H results, synchronous
            dt = (time.perf_counter() - t0) / iters
            res[f"b{b}"] = {"qps_host_pointers": b / dt, "ms_per_call": dt * 1e3}
        out["C3_10Mx768_bf16_host_pointer_calls"] = res
        ix.close()
    if "c4" in which:   # one shard of the 100M x 768 bf16 corpus over 8 GPUs: 12.5M rows
        n = 12_500_000
        spec = SynthSpec(seed=1234, kind=1, n_clusters=10000, noise=0.3)
        qspec = SynthSpec(seed=7, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=10000, base_noise=0.3)
        ix = FlatIndex(dim, "bf16", True)
        ix.add_synth(0, n, spec)
        step, scan_ms, fin_ms = timed_search_dev(ix, synth_queries(ix, 1, qspec), 1, 10, 40, 100, 10)
        out["C4_shard_12.5Mx768_bf16_b1"] = {"local_qps": 1 / step, "scan_ms": scan_ms, "finish_ms": fin_ms,
                                             "scan_GBps": n * dim * 2 / scan_ms / 1e6, "roofline_ms": 2.4}
        step, scan_ms, fin_ms = timed_search_dev(ix, synth_queries(ix, 1024, qspec), 1024, 10, 40, 4, 1)
        out["C4_shard_12.5Mx768_bf16_b1024"] = {"local_qps": 1024 / step, "scan_ms": scan_ms, "finish_ms": fin_ms,
                                                "overflow_fallback": ix.get_stat("mfma_last_overflow"),
                                                "TFLOPs": 2 * 1024 * n * dim / scan_ms / 1e9}
        ix.close()
    if "c5" in which:   # 2M x 2M self-join top-10 (exclude self), one GPU, a 1/16 slice of the queries timed
        n = 2_000_000
        spec = SynthSpec(seed=1234, kind=1, n_clusters=1414, noise=0.3)
        ix = FlatIndex(dim, "bf16", True)
        ix.add_synth(0, n, spec)
        qn = n // 16
        ix.selfjoin(10, True, 40, 0, 8192)   # warm
        t0 = time.perf_counter()
        idx, dist = ix.selfjoin(10, True, 40, 0, qn)
        dt = time.perf_counter() - t0
        ok = bool(not np.any(idx == np.arange(qn)[:, None])) and bool(np.all(np.diff(dist, axis=1) >= 0))
        out["C5_selfjoin_2Mx2M_bf16_k10"] = {"queries_timed": qn, "seconds": dt, "pairs_per_s": qn * n / dt,
                                             "TFLOPs_incl_host_copies": 2 * qn * n * dim / dt / 1e12,
                                             "full_join_one_gpu_s_extrapolated": dt * 16, "no_self_and_ascending": ok}
        ix.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
