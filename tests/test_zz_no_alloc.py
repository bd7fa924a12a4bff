"""GPU test of the entry point that allocates nothing (vdb_search_dev_noalloc): through the C ABI, against
vdb_search_dev on the same inputs — bit-exact ids, distances within the north-star tolerance (1e-5 relative; they are
in fact bit-identical whenever both calls run the GEMV scan) — on two calls in a row, the second with a SMALLER
input, so that whatever the first call left in the per-GPU scratch array must not leak into the second."""
import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec
from vectordb_amd._capi import VdbError

pytestmark = pytest.mark.gpu


def _both(ix, q_t, b, k, kp):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    oi = torch.full((b, k), -7, dtype=torch.int64, device="cuda")
    od = torch.full((b, k), -7.0, dtype=torch.float32, device="cuda")
    ri = torch.empty((b, k), dtype=torch.int64, device="cuda")
    rd = torch.empty((b, k), dtype=torch.float32, device="cuda")
    ix.search_device(q_t.data_ptr(), b, k, kp, ri.data_ptr(), rd.data_ptr(), st)
    torch.cuda.synchronize()
    ix.search_device_noalloc(q_t.data_ptr(), b, k, kp, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    return oi.cpu().numpy(), od.cpu().numpy(), ri.cpu().numpy(), rd.cpu().numpy()


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 40)])
def test_noalloc_entry_point_matches_search_dev(hip_lib, oracle_mod, dtype, kp):
    import torch
    n, dim = 50_021, 768
    spec = SynthSpec(seed=606, kind=1, n_clusters=223, noise=0.3)
    qspec = SynthSpec(seed=607, kind=2, noise=0.5, base_kind=1, base_seed=606, base_n=n, base_n_clusters=223, base_noise=0.3)
    ix = FlatIndex(dim, dtype, True)
    ix.add_synth(0, n, spec)
    assert 0 < ix.get_stat("scratch_bytes") <= 4 << 20
    queries = torch.from_numpy(oracle_mod.synth_rows(0, 8, dim, qspec)).cuda()

    # call 1: the largest input the entry point takes (8 queries, k' or k up to 64) ...
    k1 = 10 if kp else 64
    oi, od, ri, rd = _both(ix, queries, 8, k1, kp)
    assert np.array_equal(oi, ri), "ids differ from vdb_search_dev"
    assert np.allclose(od, rd, rtol=1e-5, atol=2e-6)
    # ... call 2, right behind it: a smaller input (1 query, smaller k) over the scratch the first call dirtied
    oi, od, ri, rd = _both(ix, queries[5:6].contiguous(), 1, 3, kp and 16)
    assert np.array_equal(oi, ri) and np.allclose(od, rd, rtol=1e-5, atol=2e-6)
    assert np.array_equal(od, rd), "both calls run the GEMV scan for one query: bit-identical"
    # and against the oracle
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    r_idx, _, _ = oracle_mod.exact_topk(rows, queries[5:6].cpu().numpy(), 3)
    assert np.array_equal(oi, r_idx)

    # limits are refused with a clear error, before anything is launched
    q9 = torch.zeros((9, dim), dtype=torch.float32, device="cuda")
    o = torch.empty((9, 65), dtype=torch.int64, device="cuda")
    d = torch.empty((9, 65), dtype=torch.float32, device="cuda")
    with pytest.raises(VdbError, match="at most 8 queries"):
        ix.search_device_noalloc(q9.data_ptr(), 9, 10, kp, o.data_ptr(), d.data_ptr())
    with pytest.raises(VdbError, match="<= 64"):
        ix.search_device_noalloc(q9.data_ptr(), 2, 65, 0, o.data_ptr(), d.data_ptr())
    ix.close()


def test_noalloc_other_shapes_and_a_second_handle(hip_lib, oracle_mod):
    """Generic row length (dim 100 -> the LDS-staged scan), an empty store, and a second handle on the same GPU
    sharing the scratch array right after the first."""
    import torch
    rng = np.random.default_rng(9)
    a = FlatIndex(100, "f32")
    rows = rng.standard_normal((3000, 100)).astype(np.float32)
    a.add(rows)
    b = FlatIndex(768, "f32")
    b.add_synth(0, 4000, SynthSpec(seed=1, kind=0))
    qa = torch.from_numpy(rows[[7, 2999]] + 0.01).cuda()
    qb = torch.from_numpy(oracle_mod.synth_rows(100, 3, 768, SynthSpec(seed=1, kind=0))).cuda()
    oi, od, ri, rd = _both(a, qa, 2, 20, 0)
    assert np.array_equal(oi, ri) and np.array_equal(od, rd)
    oi, od, ri, rd = _both(b, qb, 3, 5, 0)
    assert np.array_equal(oi, ri) and np.array_equal(od, rd)
    assert np.array_equal(oi[:, 0], [100, 101, 102])
    oi, od, ri, rd = _both(a, qa[:1].contiguous(), 1, 1, 0)
    assert np.array_equal(oi, ri) and np.array_equal(od, rd) and oi[0, 0] == 7
    e = FlatIndex(768, "f32")
    oi, od, ri, rd = _both(e, qb, 3, 4, 0)
    assert np.all(oi == -1) and np.all(np.isinf(od))
    for ix in (a, b, e):
        ix.close()


def test_noalloc_against_the_batched_kernels(hip_lib, oracle_mod):
    """The header's contract where vdb_search_dev does NOT run the GEMV scan: with the cost model off, 8 queries on a
    bf16 + rerank store take the batched MFMA path (bf16-rounded queries pick the k' candidates); the entry point that
    allocates nothing stays on the scan (fp32 queries).  On data without near-duplicate clusters larger than k' both
    return the same rows, with fp32 distances of those rows."""
    import torch
    n, dim = 50_021, 768
    spec = SynthSpec(seed=606, kind=1, n_clusters=223, noise=0.3)
    qspec = SynthSpec(seed=607, kind=2, noise=0.5, base_kind=1, base_seed=606, base_n=n, base_n_clusters=223, base_noise=0.3)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    ix.set_option("mfma_cost_model", 0)
    queries = torch.from_numpy(oracle_mod.synth_rows(0, 8, dim, qspec)).cuda()
    oi, od, ri, rd = _both(ix, queries, 8, 10, 40)
    assert ix.get_stat("last_path") == 0                      # the last call (noalloc): the scan
    st = torch.cuda.current_stream().cuda_stream
    ix.search_device(queries.data_ptr(), 8, 10, 40, torch.empty((8, 10), dtype=torch.int64, device="cuda").data_ptr(),
                     torch.empty((8, 10), dtype=torch.float32, device="cuda").data_ptr(), st)
    torch.cuda.synchronize()
    assert ix.get_stat("last_path") == 6                      # vdb_search_dev: the streaming MFMA kernel
    assert np.array_equal(oi, ri) and np.allclose(od, rd, rtol=1e-5, atol=2e-6)
    ix.close()
