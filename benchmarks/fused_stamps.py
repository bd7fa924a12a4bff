#!/usr/bin/env python3
"""Where the one-launch batch-1 search (scan_fused.h) spends its time: 100 MHz clock stamps written by the kernel
itself (option scan_stamps) — per block: start, end of scan, lists published, ticket drawn; the last block: each
step of its epilogue.  python benchmarks/fused_stamps.py [--rows N] [--reps R]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vectordb_amd import FlatIndex, SynthSpec

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_250_000)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--blocks-per-cu", type=int, default=2)
ap.add_argument("--balance", type=int, default=1)
args = ap.parse_args()
n, dim = args.rows, 768
spec = SynthSpec(seed=1234, kind=1, n_clusters=max(1, int(np.sqrt(n))), noise=0.3)
qspec = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=spec.n_clusters, base_noise=0.3)
ix = FlatIndex(dim, "bf16", True)
ix.add_synth(0, n, spec)
ix.set_option("scan_blocks_per_cu", args.blocks_per_cu)
ix.set_option("scan_balance", args.balance)
q = torch.empty((64, dim), dtype=torch.float32, device="cuda")
ix.synth_rows_device(q.data_ptr(), 0, 64, qspec, torch.cuda.current_stream().cuda_stream)
oi = torch.empty((1, 10), dtype=torch.int64, device="cuda")
od = torch.empty((1, 10), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(200):
    ix.search_device(q[i % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
torch.cuda.synchronize()
ix.set_option("scan_stamps", 1)
grid = min(1024, 256 * args.blocks_per_cu)
acc = []
raw = []
for r in range(args.reps):
    for i in range(3):
        ix.search_device(q[(r + i) % 64].data_ptr(), 1, 10, 40, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    v = np.array([ix.get_stat(f"scan_stamp_{i}") for i in range(16 + 6 * grid)], dtype=np.int64)
    tail, blk = v[:16], v[16:].reshape(grid, 6)
    t0 = blk[:, 0].min()
    raw.append({"dur_us": ((blk[:, 1] - blk[:, 0]) / 100.0).tolist(), "xcc": blk[:, 4].tolist(), "hw_id": blk[:, 5].tolist()})
    us = lambda x: (x - t0) / 100.0
    acc.append({
        "first_block_start_to_last_block_start": float(us(blk[:, 0].max())),
        "scan_end_first": float(us(blk[:, 1].min())), "scan_end_median": float(us(np.median(blk[:, 1]))), "scan_end_last": float(us(blk[:, 1].max())),
        "published_last": float(us(blk[:, 2].max())), "ticket_last": float(us(blk[:, 3].max())),
        "publish_cost_median": float(np.median(blk[:, 2] - blk[:, 1]) / 100.0), "ticket_cost_median": float(np.median(blk[:, 3] - blk[:, 2]) / 100.0),
        "tail_begin": float(us(tail[0])), "bound": float((tail[1] - tail[0]) / 100.0), "sweep": float((tail[2] - tail[1]) / 100.0),
        "sort": float((tail[3] - tail[2]) / 100.0), "tree_merge": float((tail[4] - tail[3]) / 100.0), "rerank": float((tail[5] - tail[4]) / 100.0),
        "rank_write": float((tail[6] - tail[5]) / 100.0), "kernel_total": float(us(tail[6])), "survivors": int(tail[8]),
    })
# who is slow?  scan duration by XCD and by the block's position on its CU (last rep)
dur = (blk[:, 1] - blk[:, 0]) / 100.0
xcc = blk[:, 4]
hw = blk[:, 5]
by_xcd = {int(x): [float(np.min(dur[xcc == x])), float(np.median(dur[xcc == x])), float(np.max(dur[xcc == x])), int(np.sum(xcc == x))] for x in np.unique(xcc)}
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print(json.dumps({"scan_us_by_xcd[min,median,max,blocks]": by_xcd,
                  "scan_us_by_se[min,median,max]": {int(x): [float(np.min(dur[se == x])), float(np.median(dur[se == x])), float(np.max(dur[se == x]))] for x in np.unique(se)},
                  "scan_us_by_blockidx_mod8": {int(x): float(np.median(dur[np.arange(grid) % 8 == x])) for x in range(8)},
                  "slowest_10_blocks[idx,xcc,se,sh,cu,us]": [[int(i), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), float(dur[i])] for i in np.argsort(-dur)[:10]],
                  "fastest_10_blocks[idx,xcc,se,sh,cu,us]": [[int(i), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), float(dur[i])] for i in np.argsort(dur)[:10]]}))
if os.environ.get("VDB_STAMPS_RAW"):
    json.dump(raw, open(os.environ["VDB_STAMPS_RAW"], "w"))
keys = acc[0].keys()
print(json.dumps({"rows": n, "grid": grid, "class_shares": [ix.get_stat(f"scan_share_{c}") for c in range(16)], "unit": "us (100 MHz clock)", "median_over_reps": {k: float(np.median([a[k] for a in acc])) for k in keys},
                  "max_over_reps": {k: float(np.max([a[k] for a in acc])) for k in keys}}, indent=1))
