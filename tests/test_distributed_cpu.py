"""CPU suite: the N>1 path (row sharding + one packed all-gather + merge) with
world_size 2 and 3 on gloo, using the CPU oracle library as the engine (tests only).
The sharded answer must equal the single-shard answer bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, dim, b, k, dtype, kp, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from vectordb_amd import SynthSpec
        from vectordb_amd.distributed import ShardedIndex, shard_range
        spec = SynthSpec(seed=1234, kind=1, n_clusters=31, noise=0.3)
        qspec = SynthSpec(seed=9, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=31, base_noise=0.3)
        sh = ShardedIndex(dim, dtype, True, device=torch.device("cpu"), lib=oracle.lib())
        sh.add_synth_partition(n, spec)
        lo, hi = shard_range(n, rank, world)
        assert sh.local.count() == hi - lo
        q = torch.from_numpy(oracle.synth_rows(0, b, dim, qspec))
        for _ in range(2):  # buffers are reused across calls
            idx, dd = sh.search(q, k, kp)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx.numpy(), dist=dd.numpy())
        sh.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype,kp", [(2, "f32", 0), (3, "bf16", 24)])
def test_sharded_equals_single(tmp_path, oracle_mod, world, dtype, kp):
    from vectordb_amd import FlatIndex, SynthSpec
    n, dim, b, k = 1001, 96, 5, 6
    mp.spawn(_worker, args=(world, _free_port(), n, dim, b, k, dtype, kp, str(tmp_path)), nprocs=world, join=True)
    spec = SynthSpec(seed=1234, kind=1, n_clusters=31, noise=0.3)
    qspec = SynthSpec(seed=9, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=31, base_noise=0.3)
    single = FlatIndex(dim, dtype, True, lib=oracle_mod.lib())
    single.add_synth(0, n, spec)
    ref_idx, ref_dist = single.search(oracle_mod.synth_rows(0, b, dim, qspec), k, kp)
    single.close()
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["idx"], ref_idx), r      # every rank holds the merged global answer
        assert np.array_equal(got["dist"], ref_dist), r


def test_shard_range_covers_everything():
    from vectordb_amd.distributed import shard_range
    for n in (0, 1, 7, 1000, 1001):
        for g in (1, 2, 3, 8):
            spans = [shard_range(n, r, g) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(g - 1))


def _sj_worker(rank, world, port, n, dim, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from vectordb_amd import SynthSpec
        from vectordb_amd.distributed import ReplicatedSelfJoin
        sj = ReplicatedSelfJoin(dim, "bf16", True, device=torch.device("cpu"), lib=oracle.lib())
        sj.add_synth(n, SynthSpec(seed=77, kind=1, n_clusters=17, noise=0.3))
        lo, idx, dd = sj.selfjoin_partition(k, exclude_self=True, rerank_kprime=16)
        np.savez(os.path.join(out_dir, f"sj{rank}.npz"), lo=lo, idx=idx, dist=dd)
        sj.close()
    finally:
        dist.destroy_process_group()


def test_replicated_selfjoin_shards_tile_the_whole_join(tmp_path, oracle_mod):
    """C5's multi-GPU form: corpus replicated, queries sharded, no collective — the ranks'
    slices laid end to end are the single-process self-join."""
    from vectordb_amd import FlatIndex, SynthSpec
    n, dim, k, world = 503, 64, 4, 2
    mp.spawn(_sj_worker, args=(world, _free_port(), n, dim, k, str(tmp_path)), nprocs=world, join=True)
    whole = FlatIndex(dim, "bf16", True, lib=oracle_mod.lib())
    whole.add_synth(0, n, SynthSpec(seed=77, kind=1, n_clusters=17, noise=0.3))
    ref_idx, ref_dist = whole.selfjoin(k, exclude_self=True, rerank_kprime=16)
    whole.close()
    parts = [np.load(tmp_path / f"sj{r}.npz") for r in range(world)]
    assert [int(p["lo"]) for p in parts] == [0, 252]
    assert np.array_equal(np.concatenate([p["idx"] for p in parts]), ref_idx)
    assert np.array_equal(np.concatenate([p["dist"] for p in parts]), ref_dist)
    assert not np.any(ref_idx == np.arange(n)[:, None])
