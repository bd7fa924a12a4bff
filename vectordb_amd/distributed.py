"""Row-sharded search across the GPUs of one node (new in the
build — the reference is single-process).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Rank r owns the contiguous row range
[r*ceil(N/G), min(N,(r+1)*ceil(N/G))) and reports *global* row indices.  A
query runs as: local scan (+ local fp32 rerank) -> ONE all-gather of the packed
per-rank top-k (k*(4+8) bytes per query per rank) -> on-device merge.  The
exchange runs on a side stream with two buffer sets, so a throughput-oriented
caller (wait=False) overlaps it with the next batch's scan: at 8 GPUs the
latency-bound all-gather is otherwise ~15 % of a 0.3 ms shard scan.  PyTorch is
used for device buffers, streams and the collective only.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .index import FlatIndex, SynthSpec


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


class ShardedIndex:
    def __init__(self, dim: int, dtype: str = "bf16", keep_f32_for_rerank: bool = True, *,
                 device: Optional[torch.device] = None, group=None, lib=None, force_collective: bool = False, two_phase: bool = False):
        self.group = group
        self.force_collective = bool(force_collective) and dist.is_initialized()   # world 1: still run the exchange (tests)
        self._side = None
        self._last_done = None
        self.two_phase = two_phase
        self._ws_slot = 0                     # next workspace set of the engine's two-phase search
        self._ws_done = [None, None]          # the event that ends each set's pending phase 2
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        dev_index = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self.local = FlatIndex(dim, dtype, keep_f32_for_rerank, dev_index, lib=lib)
        self.dim = dim
        self.n_total = 0
        self._bufs = {}

    # -- building the shard --------------------------------------------------
    def add_synth_partition(self, n_total: int, spec: SynthSpec) -> None:
        lo, hi = shard_range(n_total, self.rank, self.world)
        self.local.set_row_offset(lo)
        self.local.reserve(max(hi - lo, 1))
        self.local.add_synth(lo, hi - lo, spec)
        self.n_total = n_total

    def add_partition(self, rows_global, n_total: Optional[int] = None) -> None:
        """Every rank passes the same full [N, dim] host array; each keeps its slice."""
        n_total = len(rows_global) if n_total is None else n_total
        lo, hi = shard_range(n_total, self.rank, self.world)
        self.local.set_row_offset(lo)
        self.local.add(rows_global[lo:hi])
        self.n_total = n_total

    # -- search ----------------------------------------------------------------
    _RING = 4   # buffer sets per (b, k): lets the exchange of one batch overlap the scan of the next ones

    def _buffers(self, b: int, k: int):
        key = (b, k)
        if key not in self._bufs:
            # packed per-rank result: [ idx int64 b*k | dist fp32 b*k (padded to 8 B) ]
            words = b * k + (b * k + 1) // 2
            sets = []
            for _ in range(self._RING):
                local = torch.empty(words, dtype=torch.int64, device=self.device)
                gathered = torch.empty(self.world * words, dtype=torch.int64, device=self.device)
                out_idx = torch.empty((b, k), dtype=torch.int64, device=self.device)
                out_dist = torch.empty((b, k), dtype=torch.float32, device=self.device)
                cuda = self.device.type == "cuda"
                sets.append({"local": local, "gathered": gathered, "out_idx": out_idx, "out_dist": out_dist,
                             "ready": torch.cuda.Event() if cuda else None, "done": torch.cuda.Event() if cuda else None,
                             "used": False})
            self._bufs[key] = {"words": words, "sets": sets, "next": 0}
        return self._bufs[key]

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def search(self, queries: torch.Tensor, k: int, rerank_kprime: int = 0, wait: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries: [b, dim] fp32 tensor on self.device, identical on every rank.
        Returns (idx [b,k] int64 global rows, dist [b,k] fp32) on every rank.  The
        returned tensors are reused by the fourth-next call with the same (b, k).

        On a GPU with more than one rank the exchange (all-gather + merge) runs on a
        side stream.  wait=True (default) makes the current stream wait for it, so the
        results are ordered like any other work on that stream.  wait=False returns at
        once — the next batch's scan then overlaps this batch's exchange; call
        `wait_results()` (or synchronise the device) before using the tensors."""
        assert queries.dtype == torch.float32 and queries.is_contiguous() and queries.shape[1] == self.dim
        b = queries.shape[0]
        ring = self._buffers(b, k)
        words = ring["words"]
        cur = ring["sets"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self._RING
        cuda = self.device.type == "cuda"
        collective = self.world > 1 or self.force_collective
        if cuda and collective and cur["used"]:
            # this set's previous exchange (four calls ago) must have finished.  The HOST waits for it — it is
            # normally long done — rather than the stream: a stream-side wait is a barrier packet that idles
            # the queue ~5 us per step whether or not there is anything to wait for
            cur["done"].synchronize()
        st = self._stream()
        local = cur["local"]
        idx_ptr = local.data_ptr()
        dist_ptr = idx_ptr + 8 * b * k
        # two_phase (off by default): only the normalisation and the corpus scan stay on the caller's stream; the
        # merge/rerank/ranking of this batch goes to the side stream with the exchange (vdb_search_scan_dev /
        # vdb_search_finish_dev, one workspace set per slot).  Measured at the 8-GPU shard size on one MI355X it
        # LOSES: the next scan runs 25 % longer beside the side stream's small kernels (0.371 vs 0.322 ms/step).
        split = False
        slot = self._ws_slot
        if cuda and collective and not wait and self.two_phase:
            if self._ws_done[slot] is not None:
                torch.cuda.current_stream(self.device).wait_event(self._ws_done[slot])   # that set's last phase 2 has finished
            split = self.local.search_scan_device(queries.data_ptr(), b, k, rerank_kprime, slot, st)
        if not split:
            if cuda:                                   # a phase 2 still pending on the side stream reads the workspace sets
                for ev in self._ws_done:
                    if ev is not None:
                        torch.cuda.current_stream(self.device).wait_event(ev)
                self._ws_done = [None, None]
            self.local.search_device(queries.data_ptr(), b, k, rerank_kprime, idx_ptr, dist_ptr, st)
        if not collective:
            # the local result is already final (same (distance, index) ordering rule as the merge)
            return local[: b * k].view(b, k), local[b * k:].view(torch.float32)[: b * k].view(b, k)
        gathered, out_idx, out_dist = cur["gathered"], cur["out_idx"], cur["out_dist"]
        g_idx = gathered.data_ptr()
        if not cuda:
            dist.all_gather_into_tensor(gathered, local, group=self.group)
            self.local.merge_topk_device(g_idx + 8 * b * k, g_idx, self.world, b, k, out_idx.data_ptr(),
                                         out_dist.data_ptr(), 0, dist_shard_stride=2 * words, idx_shard_stride=words)
            return out_idx, out_dist
        if dist.get_backend(self.group) == "gloo":
            # diagnostic form (bench.py --backend gloo): the collective runs on host copies of the packed results;
            # the merge is the same kernel.  Synchronous — nothing here is a measured path.
            torch.cuda.current_stream(self.device).synchronize()
            host_all = torch.empty(self.world * words, dtype=torch.int64)
            dist.all_gather_into_tensor(host_all, local.cpu(), group=self.group)
            gathered.copy_(host_all)
            self.local.merge_topk_device(g_idx + 8 * b * k, g_idx, self.world, b, k, out_idx.data_ptr(), out_dist.data_ptr(), st,
                                         dist_shard_stride=2 * words, idx_shard_stride=words)
            return out_idx, out_dist
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        cur["ready"].record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(cur["ready"])
            if split:
                self.local.search_finish_device(slot, idx_ptr, dist_ptr, self._side.cuda_stream)
            dist.all_gather_into_tensor(gathered, local, group=self.group)
            self.local.merge_topk_device(g_idx + 8 * b * k, g_idx, self.world, b, k, out_idx.data_ptr(),
                                         out_dist.data_ptr(), self._side.cuda_stream,
                                         dist_shard_stride=2 * words, idx_shard_stride=words)
            cur["done"].record(self._side)
        cur["used"] = True
        self._last_done = cur["done"]
        if split:
            self._ws_done[slot] = cur["done"]
            self._ws_slot ^= 1
        if wait:
            main.wait_event(cur["done"])
        return out_idx, out_dist

    def wait_results(self) -> None:
        """Order the current stream after the last exchange started with wait=False."""
        if self._last_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._last_done)

    def close(self) -> None:
        self.local.close()


class ReplicatedSelfJoin:
    """All-pairs neighbours of a corpus small enough to replicate (SURVEY.md 8e, C5:
    2M x 768 bf16 + fp32 = 9.2 GB per GPU): every rank holds the WHOLE corpus and
    joins its contiguous slice of the rows-as-queries against it.  The shards are
    independent — no collective on the data path; a caller that wants the whole
    result on one rank gathers the [n, k] outputs itself."""

    def __init__(self, dim: int, dtype: str = "bf16", keep_f32_for_rerank: bool = True, *,
                 device: Optional[torch.device] = None, group=None, lib=None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        device = torch.device(device) if device is not None else torch.device("cpu")
        dev_index = device.index if device.type == "cuda" and device.index is not None else 0
        self.local = FlatIndex(dim, dtype, keep_f32_for_rerank, dev_index, lib=lib)

    def add(self, rows) -> None:
        self.local.add(rows)

    def add_synth(self, n: int, spec: SynthSpec) -> None:
        self.local.add_synth(0, n, spec)

    def selfjoin_partition(self, k: int, exclude_self: bool = True, rerank_kprime: int = 0):
        """-> (lo, idx [hi-lo, k] int64, dist [hi-lo, k] fp32): neighbours of rows [lo, hi), this rank's share."""
        lo, hi = shard_range(self.local.count(), self.rank, self.world)
        idx, dd = self.local.selfjoin(k, exclude_self=exclude_self, rerank_kprime=rerank_kprime, q_row0=lo, q_n=hi - lo)
        return lo, idx, dd

    def close(self) -> None:
        self.local.close()
