"""Eight 4096-query blocks of the 2M x 2M self-join (C5) for `rocprofv3 --kernel-trace --stats`: where a block's 11.7 ms go
(pass kernels 95 %, the two selections 3 %, finish 1.6 %)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import FlatIndex, SynthSpec
n = 2_000_000
ix = FlatIndex(768, "bf16", True)
ix.add_synth(0, n, SynthSpec(seed=515, kind=1, n_clusters=1414, noise=0.3))
ix.selfjoin(10, exclude_self=True, rerank_kprime=40, q_row0=0, q_n=4096)
t0 = time.perf_counter()
ix.selfjoin(10, exclude_self=True, rerank_kprime=40, q_row0=0, q_n=8 * 4096)
print("8 blocks", time.perf_counter() - t0)
ix.close()
