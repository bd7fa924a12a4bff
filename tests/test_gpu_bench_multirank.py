"""bench.py's multi-rank code path, exercised before an 8-GPU node ever runs it: self-launch through
torch.distributed.run, barriers, max-over-ranks, rank 0's one JSON line, the exchange + merge with world size 2 — and the
one-process vdb_create_sharded block beside it.  Both ranks share the one GPU of the test box (--device-map 0,0) and the
collective goes through host memory (--backend gloo): a diagnostic configuration, nothing here is a measured number.
bench.py runs as a fresh CHILD process (never a re-exec of the pytest process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu_and_the_inproc_block(hip_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--rows", "200000",
           "--steps", "20", "--warmup", "3", "--steps2", "3", "--no-traffic-pass", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, timeout=420)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 3 and out["scaling"] == "strong"
    assert out["config"]["rows"] == 200000 and out["config"]["rows_per_gpu"] == 100000 and out["config"]["device_map"] == [0, 0]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["checks"]["planted_neighbour_first_and_ascending"] is True          # the merged answer of two shards
    assert out["batch1024"]["checks"]["planted_neighbour_first_and_ascending"] is True
    assert out["roofline"]["bound"] == "hbm" and out["batch1024"]["batch"] == 1024
    ip = out["inproc_sharded"]
    assert "error" not in ip, ip
    assert ip["n_shards"] == 2 and ip["devices"] == [0, 0] and ip["p2p_direct_shards"] == 2
    for blk in ("batch1", "batch1024"):
        assert ip[blk]["planted_neighbour_first_and_ascending"] is True and ip[blk]["value"] > 0
