#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: QPS @k=10 over a synthetic
10M x 768 bf16 corpus (fp32 rerank, k'=40), batch 1 AND batch 1024.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus 8                 # launches its own ranks (torch.distributed.run) before any GPU call
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

For N > 1 the line also carries `inproc_sharded`: the same workload through ONE vdb_create_sharded handle over the same
GPUs (one process, P2P result stores, no collective), run in a child process once the ranks have released their shards
(--no-inproc skips it; --inproc adds it at N = 1).  --backend gloo --device-map 0,0 runs two ranks on one GPU
(diagnostic: it exercises the multi-rank code path where no second GPU exists).

A "step" is one pass of the hot path over one query batch: normalise the queries,
scan the (row-sharded) corpus with the fused top-k' kernel, merge, fp32 rerank,
and for N>1 one RCCL all-gather + on-device merge.  The corpus (10M rows in
total, row-sharded over the ranks: strong scaling) and the queries are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line: the headline fields
are the batch-1 workload (the north-star target), `batch1024` holds the other half
of the metric (MFMA path) timed in the same run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows (BASELINE: 10M)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1, help="queries per step of the headline workload (BASELINE: 1 and 1024)")
    ap.add_argument("--batch2", type=int, default=1024, help="second workload timed in the same run (0 = none)")
    ap.add_argument("--steps2", type=int, default=20, help="timed steps of the second workload (at most --steps)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--kprime", type=int, default=40)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="skip the rocprofv3 --pmc FETCH_SIZE pass (a child process, run before this one touches the GPU) that fills roofline.traffic")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic, 1 GPU: run the all-gather + merge exchange with a single-rank RCCL group")
    ap.add_argument("--sync-exchange", action="store_true", help="diagnostic: wait for each batch's exchange (no overlap)")
    ap.add_argument("--mfma-min-b", type=int, default=0, help="diagnostic: smallest batch sent to the MFMA path")
    ap.add_argument("--clusters", type=int, default=0, help="synthetic clusters (default sqrt(rows))")
    ap.add_argument("--time-every", type=int, default=4, help="HIP-event timing of the kernel on every Nth step (an event record idles the queue ~6 us)")
    ap.add_argument("--two-phase", action="store_true", help="diagnostic: the merge/rerank of a batch moves to the exchange's side stream")
    ap.add_argument("--no-skip-sample", action="store_true", help="diagnostic: the last MFMA pass re-multiplies the sampled tiles")
    ap.add_argument("--mfma-split", type=int, default=0, help="diagnostic: mfma_split_x1000 option")
    ap.add_argument("--ablate", type=int, default=0, help="diagnostic: MFMA kernel ablation bits (results invalid)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="diagnostic: vdb_set_option on the local shard")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 exchange: nccl (= RCCL over xGMI, the measured path) or gloo (diagnostic: "
                         "the all-gather goes through host memory; lets the multi-rank code path run where RCCL cannot, e.g. two ranks on one GPU)")
    ap.add_argument("--device-map", default="", metavar="D0,D1,...", help="diagnostic: rank r uses GPU D[r] (default: its LOCAL_RANK); '0,0' runs two ranks on one GPU")
    ap.add_argument("--inproc", action="store_true", help="also time the ONE-process form (vdb_create_sharded over the same GPUs, P2P result stores, no collective) "
                                                          "in a child process and report it as `inproc_sharded` (default when --gpus > 1)")
    ap.add_argument("--no-inproc", action="store_true")
    ap.add_argument("--inproc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stream", default="default", choices=["default", "created", "created-high"],
                    help="diagnostic: the stream the searches are issued on: the process' default (null) stream, a created one, or a created high-priority one")
    return ap.parse_args()


def device_map(args):
    return [int(x) for x in args.device_map.split(",") if x.strip() != ""]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process and
    relay its exit code.  Nothing in this process has touched the GPU yet (device_count() does not initialise it)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not device_map(args):
        raise SystemExit(f"--gpus {args.gpus} but this node shows {have} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(dim, k, target_rows, seed, budget_s=12.0):
    """The CPU oracle (a port of the reference's algorithm: exact cosine top-k, fp32 storage,
    fp64 accumulate, OpenMP over rows) and numpy's BLAS brute force (SURVEY 8d baseline (a)), both timed on the first
    1M rows of the same corpus with the same queries; the 10M figure is that rate scaled linearly (a linear scan)."""
    import numpy as np
    from oracle import oracle
    from vectordb_amd import SynthSpec
    # the whole corpus when the host has the memory for it (fp32 rows + working room), else its first 1M rows
    # with the rate scaled linearly — a brute-force scan's cost per query is proportional to the rows it reads
    mem_gb = avail_gb = 0.0
    try:
        with open("/proc/meminfo") as fh:
            mi = {ln.split(":")[0]: float(ln.split()[1]) / 1048576.0 for ln in fh if ":" in ln}
        mem_gb, avail_gb = mi.get("MemTotal", 0.0), mi.get("MemAvailable", 0.0)
    except OSError:
        pass
    need_gb = target_rows * dim * 4 / 2**30 * 1.25 + 4.0
    full = avail_gb >= need_gb
    sample_rows = target_rows if full else min(1_000_000, target_rows)
    spec = SynthSpec(seed=seed, kind=1, n_clusters=max(1, int(np.sqrt(target_rows))), noise=0.3)
    rows = np.empty((sample_rows, dim), np.float32)
    for lo in range(0, sample_rows, 500_000):            # in pieces: one fp32 copy of the corpus, never two
        m = min(500_000, sample_rows - lo)
        rows[lo:lo + m] = oracle.normalize_rows(oracle.synth_rows(lo, m, dim, spec))
    qspec = SynthSpec(seed=seed + 1, kind=2, noise=0.5, base_kind=1, base_seed=seed, base_n=sample_rows,
                      base_n_clusters=spec.n_clusters, base_noise=0.3)
    queries = oracle.synth_rows(0, 64, dim, qspec)
    oracle.exact_topk(rows, queries[:1], k)  # warm
    t0 = time.perf_counter()
    done = 0
    while done < 4096 and (time.perf_counter() - t0) < budget_s:     # ~12 s of CPU work, cycling over 64 distinct queries
        q = done % 64
        oracle.exact_topk(rows, queries[q:q + 1], k)
        done += 1
    dt = time.perf_counter() - t0
    qps_sample = done / dt
    t1 = time.perf_counter()
    reps = 0
    qn = oracle.normalize_rows(queries)
    while reps < 256 and (time.perf_counter() - t1) < 5.0:
        s = rows @ qn[reps % 64]
        top = np.argpartition(-s, k)[:k]
        top = top[np.argsort(-s[top], kind="stable")]
        reps += 1
    np_qps = reps / (time.perf_counter() - t1)
    scale = sample_rows / target_rows
    return {"value": qps_sample * scale, "unit": "queries/s", "cores": int(oracle.lib().ora_num_threads()), "kind": "port",
            "sample": (f"{done} batch-1 queries over ALL {sample_rows} rows (x{dim} fp32) of the corpus, exact top-{k}: measured at full size, nothing scaled"
                       if full else
                       f"{done} batch-1 queries over the first {sample_rows} rows (x{dim} fp32) of the corpus, exact top-{k}; "
                       f"QPS scaled by {sample_rows}/{target_rows} rows (linear scan) because the host showed {avail_gb:.0f} GiB available "
                       f"and the full fp32 corpus needs {need_gb:.0f}"),
            "host_ram_gib": round(mem_gb, 1), "host_ram_available_gib": round(avail_gb, 1), "measured_at_full_size": full,
            "qps_at_sample_rows": qps_sample, "numpy_sgemv_qps_at_sample_rows": np_qps,
            "numpy_sgemv_qps_scaled": np_qps * scale, "host_cpus": os.cpu_count(),
            "chroma_baseline": _chroma_note()}


def _chroma_note():
    """the reference's own engine, if it can be imported on this box."""
    import importlib.util
    if importlib.util.find_spec("chromadb") is None:
        return "Chroma baseline unavailable (chromadb is not installed here); the reference's only published figure is 0.011 s/query incl. embedding on its toy set (README.md:120)"
    return "chromadb importable: not timed by this script"


def traffic_pass(args):
    """HBM bytes per launch of the dominant kernels, measured: a short run of this same script as a CHILD process under
    `rocprofv3 --pmc FETCH_SIZE` (a counter pass of its own, no tracing), started before this process touches the GPU.
    FETCH_SIZE is in KiB and on gfx950 tallies a wide coalesced 16 B/lane stream at half its bytes
    (MI355X_MICROARCH.md, HBM): bytes = 2 * 1024 * FETCH_SIZE.  Returns {kernel family: bytes per launch} or a reason."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="vdb_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
           "--gpus", "1", "--steps", "4", "--warmup", "1", "--steps2", "2", "--no-cpu-baseline", "--no-traffic-pass",
           "--rows", str(args.rows), "--dim", str(args.dim), "--batch", str(args.batch), "--batch2", str(args.batch2),
           "--k", str(args.k), "--kprime", str(args.kprime), "--dtype", args.dtype, "--seed", str(args.seed)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    try:
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            return None, "the counter pass did not finish in 240 s"
        if rc != 0:
            return None, f"the counter pass exited with {rc}"
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == "FETCH_SIZE":
                        per.setdefault(r["Kernel_Name"], []).append(2.0 * 1024.0 * float(r["Counter_Value"]))
        out = {}
        for name, vals in per.items():
            fam = next((f for f in ("scan_topk_fused_kernel", "scan_topk_kernel", "scan_mfma_wide16_kernel<1", "scan_stream_lds_kernel<1", "probe_read_kernel") if f in name), None)
            if fam is None:
                continue
            big = [v for v in vals if v >= 0.5 * max(vals)]      # the batched kernels launch twice per batch: the last pass is the large one
            out[fam.split("<")[0]] = {"bytes_per_launch": sum(big) / len(big), "launches": len(big)}
        return (out, "rocprofv3 --pmc FETCH_SIZE child pass of this command (4 + 2 steps), bytes = 2*1024*FETCH_SIZE") if out else (None, "no FETCH_SIZE rows in the counter pass's output")
    except OSError as e:
        return None, f"the counter pass could not start: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_inproc_child(args, world):
    """`python bench.py --inproc-child ...` as a child process with a time limit -> its JSON block, or {"error": ...}."""
    import signal
    cmd = [sys.executable, os.path.abspath(__file__), "--inproc-child", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--rows", str(args.rows), "--dim", str(args.dim), "--batch", str(args.batch), "--batch2", str(args.batch2), "--steps2", str(args.steps2),
           "--k", str(args.k), "--kprime", str(args.kprime), "--dtype", args.dtype, "--seed", str(args.seed), "--time-every", str(args.time_every)]
    if args.device_map:
        cmd += ["--device-map", args.device_map]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                                                             "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")}
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            so, se = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            return {"error": "the one-process sharded run did not finish in 180 s"}
        lines = [ln for ln in so.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": f"the one-process sharded run exited with {p.returncode}: " + se.decode("utf-8", "replace")[-400:]}
        return json.loads(lines[-1])
    except (OSError, ValueError) as e:
        return {"error": f"the one-process sharded run could not be started or read: {e}"}


def inproc_child(args):
    """The one-process form of the same workload: ONE vdb_create_sharded handle over the N GPUs, queries
    and results on shard 0's device, every shard's last block storing its k results into shard 0's gather buffer (P2P over
    xGMI), one merge kernel — no torch.distributed, no collective.  Prints one JSON block."""
    import numpy as np
    import torch
    from vectordb_amd import FlatIndex, SynthSpec
    from vectordb_amd.synth import query_source_row
    g = args.gpus
    dmap = device_map(args)
    devices = [dmap[r % len(dmap)] for r in range(g)] if dmap else list(range(g))
    n, dim, k = args.rows, args.dim, args.k
    kp = args.kprime if args.dtype == "bf16" else 0
    spec = SynthSpec(seed=args.seed, kind=1, n_clusters=args.clusters or max(1, int(np.sqrt(n))), noise=0.3)
    qspec = SynthSpec(seed=args.seed + 1, kind=2, noise=0.5, base_kind=1, base_seed=args.seed, base_n=n,
                      base_n_clusters=spec.n_clusters, base_noise=0.3)
    torch.cuda.set_device(devices[0])
    dev = torch.device("cuda", devices[0])
    ix = FlatIndex(dim, args.dtype, True, devices=devices)
    ix.reserve(n)
    ix.add_synth(0, n, spec)
    out = {"n_shards": g, "devices": devices, "p2p_direct_shards": ix.get_stat("p2p_direct_shards"), "rows_per_shard": ix.get_stat("rows_per_shard"),
           "exchange": "each shard's finish stores its k results into shard 0's gather buffer (P2P over xGMI; a peer copy without peer access), one merge kernel on shard 0"}
    st = torch.cuda.current_stream(dev).cuda_stream
    for b, steps, label in ((args.batch, args.steps, "batch%d" % args.batch), (args.batch2, max(1, min(args.steps, args.steps2)), "batch%d" % args.batch2)):
        if not b or (label in out):
            continue
        n_sets = 16 if b > 1 else 64
        qpool = torch.empty((n_sets, b, dim), dtype=torch.float32, device=dev)
        for s_ in range(n_sets):
            ix.synth_rows_device(qpool[s_].data_ptr(), s_ * b, b, qspec, st)
        oi = [torch.empty((b, k), dtype=torch.int64, device=dev) for _ in range(4)]
        od = [torch.empty((b, k), dtype=torch.float32, device=dev) for _ in range(4)]
        ix.warm(b, k, kp)
        ix.set_option("timing_every", max(1, args.time_every))

        def step(i):
            ix.search_device(qpool[i % n_sets].data_ptr(), b, k, kp, oi[i & 3].data_ptr(), od[i & 3].data_ptr(), st)

        for i in range((150 if b <= 64 else 8) + args.warmup):
            step(i)
        torch.cuda.synchronize(dev)
        for d in set(devices):
            torch.cuda.synchronize(d)
        ix.get_stat("sharded_issue_ns_avg")
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        issue_us = ix.get_stat("sharded_issue_ns_avg") / 1e3
        lat = []
        for i in range(min(steps, 100)):
            t1 = time.perf_counter()
            step(i)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t1) * 1e3)
        last = (min(steps, 100) - 1) % n_sets
        idx_h, dist_h = oi[(min(steps, 100) - 1) & 3].cpu().numpy(), od[(min(steps, 100) - 1) & 3].cpu().numpy()
        src = np.array([query_source_row(qspec, last * b + j) for j in range(min(b, 8))])
        rccl_blk = None
        if len(set(devices)) == len(devices):
            # the same steps with the exchange north_star names — ONE RCCL all-gather + merge — behind the same handle
            try:
                ix.set_option("shard_exchange_rccl", 1)
                for i in range(20 + args.warmup):
                    step(i)
                torch.cuda.synchronize(dev)
                t2 = time.perf_counter()
                for i in range(steps):
                    step(i)
                torch.cuda.synchronize(dev)
                e2 = time.perf_counter() - t2
                lat2 = []
                for i in range(min(steps, 100)):
                    t3 = time.perf_counter()
                    step(i)
                    torch.cuda.synchronize(dev)
                    lat2.append((time.perf_counter() - t3) * 1e3)
                rccl_blk = {"ms_per_step": 1e3 * e2 / steps, "value": steps * b / e2, "sync_latency_ms_median": float(np.median(lat2))}
            except Exception as exc:          # noqa: BLE001 — a diagnostic block must not take the line down
                rccl_blk = {"error": str(exc)[:300]}
            finally:
                try:
                    ix.set_option("shard_exchange_rccl", 0)
                except Exception:             # noqa: BLE001
                    pass
        out[label] = {"value": steps * b / elapsed, "unit": "queries/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
                      "with_rccl_all_gather_exchange": rccl_blk,
                      "host_issue_ms_per_step": 1e3 * t_host / steps, "library_host_us_per_call": issue_us,
                      "sync_latency_ms": {"median": float(np.median(lat)), "p99": float(np.sort(lat)[min(len(lat) - 1, int(0.99 * len(lat)))])},
                      "planted_neighbour_first_and_ascending": bool(np.array_equal(idx_h[: len(src), 0], src)) and bool(np.all(np.diff(dist_h, axis=1) >= 0))}
        del qpool
    ix.close()
    print(json.dumps(out), flush=True)


def dist_rank0():
    return int(os.environ.get("RANK", "0")) == 0


def _library_gemm(m, n, k, iters=5):
    import numpy as np
    import torch
    a = (torch.randn(m, k, device="cuda") / np.sqrt(k)).to(torch.bfloat16)
    bt = (torch.randn(n, k, device="cuda") / np.sqrt(k)).to(torch.bfloat16)
    c = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        torch.matmul(a, bt.t(), out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.matmul(a, bt.t(), out=c)
    torch.cuda.synchronize()
    return 2.0 * m * n * k * iters / (time.perf_counter() - t0) / 1e12


def measure(sh, args, b, steps, warmup, qspec, device, world, n_local):
    """Time `steps` steps of batch b (after `warmup` untimed ones) -> result block with its roofline."""
    import numpy as np
    import torch
    import torch.distributed as dist
    n, dim, k = args.rows, args.dim, args.k
    kp = args.kprime if args.dtype == "bf16" else 0
    # a pool of distinct query batches, resident in HBM
    n_sets = 16 if b > 1 else 64
    qpool = torch.empty((n_sets, b, dim), dtype=torch.float32, device=device)
    for s_ in range(n_sets):
        sh.local.synth_rows_device(qpool[s_].data_ptr(), s_ * b, b, qspec, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

    def step(i):
        # throughput mode: the exchange of batch i (side stream) overlaps the scan of batch i+1;
        # every batch's exchange and merge still complete inside the timed region (device sync)
        return sh.search(qpool[i % n_sets], k, kp, wait=args.sync_exchange)

    # Settle, then the W warm-up steps.  A fresh process sometimes runs its first few hundred milliseconds of kernels ~2 %
    # slow (clocks still ramping: two of four otherwise identical runs with 20 warm-up steps on one box); the same fixed
    # number of untimed steps on every rank (the steps are collective) takes the timed region past that.
    # (round 3: at least ~0.8 s of them — benchmarks/stream_ab.py: a 1.25M-row step runs 0.306 ms for the first ~0.4 s of a
    # loop and 0.295 ms from then on, whichever stream it is issued on; 150 steps of 0.3 ms ended before that)
    settle = 150 if b <= 64 else 8
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t_est = time.perf_counter()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    per_step = max((time.perf_counter() - t_est) / 20, 1e-6)
    settle = max(settle, min(20000, int(0.8 / per_step)))
    if world > 1:   # the steps are collective: every rank must run the SAME number of them
        t = torch.tensor([settle], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        settle = int(t.item())
    for i in range(settle + warmup):
        step(i)
    torch.cuda.synchronize()
    sh.local.set_option("timing_ring", max(1, min(steps, 65536)))
    sh.local.set_option("timing_every", max(1, args.time_every))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tstats = sh.local.timing_stats()
    sh.local.set_option("timing_every", 1)
    # outside the timed region: per-call latency with a sync after every call (median and p99)
    lat = []
    for i in range(min(steps, 100)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(i)
        torch.cuda.synchronize()   # waits for the side stream too: full per-call latency incl. the exchange
        lat.append((time.perf_counter() - t1) * 1e3)
    lat = np.sort(np.asarray(lat))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity inside the bench: planted neighbour of the last batch is rank 1 and results ascend
    # (a fresh call: the tensors the timed loop returned are ring buffers the latency loop has reused since)
    last = (steps - 1) % n_sets
    idx, dd = sh.search(qpool[last], k, kp)
    torch.cuda.synchronize()
    idx_h, dist_h = idx.cpu().numpy(), dd.cpu().numpy()
    from vectordb_amd.synth import query_source_row
    src = np.array([query_source_row(qspec, last * b + j) for j in range(min(b, 8))])
    planted_ok = bool(np.array_equal(idx_h[: len(src), 0], src)) and bool(np.all(np.diff(dist_h, axis=1) >= 0))

    esz = 2 if args.dtype == "bf16" else 4
    path = sh.local.get_stat("last_path")            # 0 = GEMV scan, 4/5 = MFMA pass kernels, 6 = streaming MFMA kernel
    scan_ms = tstats["scan_ms_sum"] / max(tstats["n"], 1)
    kernel = {0: "scan_topk_fused_kernel (normalise + scan + fold + fp32 rerank in one launch)" if sh.local.get_stat("last_fused") else "scan_topk_kernel", 4: "scan_mfma_wide16_kernel<1, 0, 2> (128-wide query tile) + sample passes", 5: "scan_mfma_wide16_kernel<1> + sample passes",
              6: "scan_stream_lds_kernel<1> + sample passes"}.get(path, f"path {path}")
    # one query tile covers the batch (the corpus comes from HBM once) and b flop per byte is under the ridge of
    # 2500 TFLOP/s / 8 TB/s = 312: such a batch is HBM-bound like a single query, whichever kernel serves it
    one_tile_hbm_bound = path in (4, 5) and b <= (256 if path == 5 else 128) and 2.0 * b / esz < MFMA_BF16_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBPS
    if path in (4, 5) and not one_tile_hbm_bound:
        alg = 2.0 * b * n_local * dim
        roof = {"bound": "mfma", "achieved": alg / (scan_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s"}
        # operand bytes staged through LDS per batch: the path that actually binds the kernel
        bn = 256 if path == 5 else 128
        b_pad = ((b + bn - 1) // bn) * bn
        roof["lds_staged_bytes"] = float(n_local) * dim * esz * b_pad * (1.0 / bn + 1.0 / 256)
        roof["lds_staging_TBps"] = roof["lds_staged_bytes"] / (scan_ms * 1e-3) / 1e12
    else:
        passes = 1 if one_tile_hbm_bound else sh.local.get_stat("last_corpus_passes")
        alg = float(n_local) * dim * esz * passes   # bytes per corpus pass x passes of the batch
        roof = {"bound": "hbm", "achieved": alg / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "corpus_passes_per_batch": passes}
    roof["frac"] = roof["achieved"] / roof["peak"]
    # the same ceilings as measured on this device (streaming read of the store in the scan's own launch shape;
    # bare bf16 MFMA loop)
    try:
        if roof["bound"] == "mfma":
            # bare MFMA loops on RANDOM operands, every CU busy: the shape the pass kernel is built on, and the other one
            roof["measured_peak"] = sh.local.get_stat("probe_mfma16_bf16_gflops") / 1e3
            roof["measured_peak_other_shape_32x32x16"] = sh.local.get_stat("probe_mfma_bf16_gflops") / 1e3
        else:
            roof["measured_peak"] = sh.local.get_stat("probe_hbm_read_mbps") / 1e3
        roof["frac_of_measured_peak"] = roof["achieved"] / roof["measured_peak"] if roof["measured_peak"] else None
    except RuntimeError:
        roof["measured_peak"] = None
    if roof["bound"] == "mfma" and dist_rank0():
        # the vendor library's bf16 GEMM (torch.matmul) on random operands on this box, outside the timed region: the
        # search's own shape with the scores written out, and a square one (the chip is power-limited)
        roof["library_gemm_tflops_random_operands"] = {"%dx262144x%d" % (b, dim): _library_gemm(b, 1 << 18, dim),
                                                       "8192x8192x8192": _library_gemm(8192, 8192, 8192)}
    # HBM bytes per launch: only a rocprofv3 --pmc FETCH_SIZE pass can tell; main() fills this in from traffic_pass()
    roof["traffic"] = None
    roof["kernel"] = kernel
    roof["kernel_avg_ms"] = scan_ms
    roof["kernel_timing"] = f"HIP events around the scan phase on every {max(1, args.time_every)}th step of the timed region ({tstats['n']} samples)"
    roof["merge_rerank_avg_ms"] = tstats["merge_ms_sum"] / max(tstats["n"], 1)
    del qpool
    return {
        "value": steps * b / elapsed, "unit": "queries/s", "steps": steps, "warmup": warmup, "settle_steps_before_warmup": settle, "ms_per_step": 1e3 * elapsed / steps,
        "batch": b, "roofline": roof,
        "checks": {"planted_neighbour_first_and_ascending": planted_ok,
                   # measured_peak (HBM): the best of bare reads in the scan's launch shape and partition (one or two tiles in
                   # flight per wave, with and without pauses) and of the scan's own loop with the selection taken out, each as
                   # bursts of 24 back-to-back launches after 0.7 s of reads
                   "measured_peak_is_a_ceiling": (roof.get("frac_of_measured_peak") or 0) <= 1.02,

                   "mfma_overflow_fallback": sh.local.get_stat("mfma_last_overflow"),
                   "mfma_hits_per_query": sh.local.get_stat("mfma_last_hits") / b,
                   **({"prof": [sh.local.get_stat("mfma_prof_" + c) for c in "abcd"]} if args.ablate & 8 else {})},
        "sync_latency_ms": {"median": float(np.median(lat)), "p99": float(lat[min(len(lat) - 1, int(0.99 * len(lat)))]), "calls": len(lat)},
    }


def main():
    args = parse_args()
    if args.inproc_child:
        return inproc_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                      # does not return
    # stdout carries the ONE JSON line and nothing else: whatever libraries write to file descriptor 1 (RCCL prints a
    # version banner there when a communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    traffic, traffic_note = None, "not collected (--no-traffic-pass, or more than one rank)"
    profiled = any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "HSA_TOOLS_LIB" in os.environ or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled:
        # under a profiler this process may already hold the GPU (its preloaded tool initialises it): no child processes then
        traffic_note = "not collected: this run is itself under a profiler"
    elif world == 1 and not args.no_traffic_pass:
        traffic, traffic_note = traffic_pass(args)          # a child process; nothing here has touched the GPU yet
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from vectordb_amd import SynthSpec
    from vectordb_amd.distributed import ShardedIndex, shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    dmap = device_map(args)
    dev_index = dmap[local_rank % len(dmap)] if dmap else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 and args.backend == "gloo":
        dist.init_process_group("gloo")                          # diagnostic: the exchange goes through host memory
    elif world > 1 or (args.force_collective and "TORCHELASTIC_RUN_ID" in os.environ):
        dist.init_process_group("nccl", device_id=device)        # the launcher's store (env://)
    elif args.force_collective:
        # (under the launcher a tcp:// init would look for the AGENT's store on that port and wait for ever)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1, device_id=device)

    kp = args.kprime if args.dtype == "bf16" else 0
    n, dim, k = args.rows, args.dim, args.k
    spec = SynthSpec(seed=args.seed, kind=1, n_clusters=args.clusters or max(1, int(np.sqrt(n))), noise=0.3)   # clustered (SURVEY 8d ii)
    qspec = SynthSpec(seed=args.seed + 1, kind=2, noise=0.5, base_kind=1, base_seed=args.seed, base_n=n,
                      base_n_clusters=spec.n_clusters, base_noise=0.3)

    t_build = time.perf_counter()
    sh = ShardedIndex(dim, args.dtype, True, device=device, force_collective=args.force_collective, two_phase=args.two_phase)
    sh.add_synth_partition(n, spec)
    lo, hi = shard_range(n, rank, world)
    n_local = hi - lo
    if args.blocks_per_cu:
        sh.local.set_option("scan_blocks_per_cu", args.blocks_per_cu)
    if args.ablate:
        sh.local.set_option("mfma_ablate", args.ablate)
    if args.no_skip_sample:
        sh.local.set_option("mfma_skip_sample", 0)
    if args.mfma_split:
        sh.local.set_option("mfma_split_x1000", args.mfma_split)
    if args.mfma_min_b:
        sh.local.set_option("batch_mfma_min_b", args.mfma_min_b)
    if args.nt >= 0:
        sh.local.set_option("scan_nt_loads", args.nt)
    for ov in args.option:
        name, _, val = ov.partition("=")
        sh.local.set_option(name, int(val))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    if args.stream != "default":
        lo_hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        torch.cuda.set_stream(torch.cuda.Stream(device, priority=min(lo_hi) if args.stream == "created-high" else 0))
    head = measure(sh, args, args.batch, args.steps, args.warmup, qspec, device, world, n_local)
    second = None
    if args.batch2 and args.batch2 != args.batch and args.dtype == "bf16":
        second = measure(sh, args, args.batch2, max(1, min(args.steps, args.steps2)), max(1, min(args.warmup, 3)), qspec, device, world, n_local)

    if rank == 0:
        def workload(b):
            return (f"{n}x{dim} {args.dtype} clustered synthetic corpus row-sharded over {world} GPU(s), "
                    f"batch {b}, exact cosine top-{k}" + (f", k'={kp} fp32 rerank" if kp else ""))
        out = {
            "metric": f"QPS @k={k}, {n // 1_000_000}Mx{dim} corpus, batch {args.batch}",
            "value": head["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload(args.batch),
                       "rows": n, "dim": dim, "batch": args.batch, "k": k, "kprime": kp, "rows_per_gpu": n_local,
                       "parallelism": (f"row-shard x{world} + all-gather merge ({'RCCL' if args.backend == 'nccl' else 'gloo through host memory: diagnostic'})"
                                       if world > 1 else "single GPU"),
                       **({"device_map": dmap} if dmap else {})},
            "roofline": head["roofline"],
            "checks": head["checks"],
            "sync_latency_ms": head["sync_latency_ms"],
            "build_s": t_build,
        }
        if second is not None:
            second["metric"] = f"QPS @k={k}, {n // 1_000_000}Mx{dim} corpus, batch {args.batch2}"
            second["workload"] = workload(args.batch2)
            out[f"batch{args.batch2}"] = second
        # HBM bytes per launch of each block's dominant kernel, from the counter pass run before the timed work (or null)
        for blk in [out] + ([second] if second is not None else []):
            fam = blk["roofline"]["kernel"].split("<")[0].split(" ")[0]
            got = (traffic or {}).get(fam)
            blk["roofline"]["traffic"] = got["bytes_per_launch"] if got else None
            blk["roofline"]["traffic_unit"] = "HBM bytes per launch of " + fam
            blk["roofline"]["traffic_source"] = traffic_note if got or traffic is None else "the counter pass saw no launch of " + fam
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dim, k, n, args.seed)
    sh.close()
    del sh
    torch.cuda.empty_cache()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the P2P form next to the collective one.  The other ranks have released their shards and are
        # leaving; the one-process handle over the same GPUs runs in a CHILD (a hang or fault there cannot take this
        # line down: it is timed out and reported as an error string)
        if (world > 1 or args.inproc) and not args.no_inproc:
            out["inproc_sharded"] = run_inproc_child(args, world)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
