"""The cross-device half of vdb_create_sharded on PHYSICAL GPUs: peer copies of the queries, P2P result stores
into shard 0's gather buffer, events across devices, save + vdb_load_sharded — against a single store, bit for bit.
Skips on a one-GPU box (where tests/test_gpu_round3.py runs the same code through the shard_force_remote hook).  In a file
of its own, collected near the end: it is the one test whose code path no box of this pool has ever executed."""
import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]), f"{what}: ids differ"
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), f"{what}: distances differ in their bits"


def test_two_physical_gpus_when_the_box_has_them(hip_lib, oracle_mod, tmp_path):
    """The cross-device half of vdb_create_sharded (peer copies of the queries, P2P result stores, events across
    devices) needs a second GPU; the one-GPU box runs the same code through shard_force_remote above."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    dim, n = 768, 50_000
    spec = SynthSpec(seed=71, kind=1, n_clusters=223, noise=0.3)
    qspec = SynthSpec(seed=72, kind=2, noise=0.5, base_kind=1, base_seed=71, base_n=n, base_n_clusters=223, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 70, dim, qspec)
    one = FlatIndex(dim, "bf16", True)
    one.add_synth(0, n, spec)
    devs = list(range(torch.cuda.device_count()))
    sh = FlatIndex(dim, "bf16", True, devices=devs)
    sh.reserve(n)
    sh.add_synth(0, n, spec)
    for b in (1, 4, 33, 70):
        _same(sh.search(queries[:b], 10, 40), one.search(queries[:b], 10, 40), f"{len(devs)} GPUs b={b}")
    i1, d1, r1 = sh.search_fetch(queries[:3], 10, 40)
    i0, d0, r0 = one.search_fetch(queries[:3], 10, 40)
    assert np.array_equal(i1, i0) and np.array_equal(d1, d0) and np.array_equal(r1, r0)
    allow = np.arange(n) % 7 == 0
    _same(sh.search_masked(queries[:4], 10, allow, 40), one.search_masked(queries[:4], 10, allow, 40), "masked, physical GPUs")
    sh.set_option("shard_exchange_rccl", 1)            # the same searches with ONE RCCL all-gather instead of the P2P stores
    for b in (1, 4, 33):
        _same(sh.search(queries[:b], 10, 40), one.search(queries[:b], 10, 40), f"{len(devs)} GPUs, RCCL exchange, b={b}")
    sh.set_option("shard_exchange_rccl", 0)
    sh.save(str(tmp_path / "s"))
    back = FlatIndex.load(str(tmp_path / "s"), "bf16", True, devices=devs[::-1])
    _same(back.search(queries[:5], 10, 40), one.search(queries[:5], 10, 40), "reload, physical GPUs")
    for ix in (one, sh, back):
        ix.close()
