"""Generates tests/golden/*.npz with the numpy statement of the exact search
(oracle.exact_topk_numpy — independent of the C oracle and of the HIP code).

The reference holds no fixtures for this path (parity unpinned), so
these are produced here from seeded synthetic data; inputs are regenerated from
the spec by the bit-exact counter generator, adversarial inputs are stored whole.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from vectordb_amd import SynthSpec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def spec_dict(s):
    return {f: getattr(s, f) for f, _ in s._fields_}


def main():
    # 1. seeded synthetic: N=4096, d=768, B=8, k in {1,10,30,100}, seed 1234
    n, dim, b = 4096, 768, 8
    for name, spec in (("iso", SynthSpec(seed=1234, kind=0)),
                       ("clustered", SynthSpec(seed=1234, kind=1, n_clusters=64, noise=0.3))):
        qspec = SynthSpec(seed=4321, kind=2, noise=0.5, base_kind=spec.kind, base_seed=spec.seed, base_n=n,
                          base_n_clusters=spec.n_clusters, base_noise=spec.noise)
        rows = oracle.normalize_rows(oracle.synth_rows(0, n, dim, spec))
        queries = oracle.synth_rows(0, b, dim, qspec)
        out = {"n": n, "dim": dim, "b": b, "rows_checksum": np.float64(rows.astype(np.float64).sum()),
               "row0_head": rows[0, :8], "query0_head": queries[0, :8]}
        for k in (1, 10, 30, 100):
            idx, dist, sc = oracle.exact_topk_numpy(rows, queries, k)
            out[f"idx_k{k}"] = idx
            out[f"dist_k{k}"] = dist
            out[f"score_k{k}"] = sc
        for key, val in spec_dict(spec).items():
            out[f"spec_{key}"] = val
        for key, val in spec_dict(qspec).items():
            out[f"qspec_{key}"] = val
        np.savez_compressed(os.path.join(OUT, f"synth_{name}_n4096.npz"), **out)

    # 2. adversarial small set, stored whole: duplicates, exact ties, a zero row, k > N
    rng = np.random.default_rng(7)
    dim = 64
    base = rng.standard_normal((20, dim)).astype(np.float32)
    rows = np.concatenate([base, base[:5], 2.5 * base[5:8], np.zeros((1, dim), np.float32), -base[:2]])
    queries = np.concatenate([base[:3] + 0.01 * rng.standard_normal((3, dim)).astype(np.float32),
                              np.zeros((1, dim), np.float32), base[6:7]])
    rows_n = oracle.normalize_rows(rows)
    out = {"rows": rows, "queries": queries}
    for k in (1, 5, 31, 40):
        kk = min(k, len(rows))
        idx, dist, sc = oracle.exact_topk_numpy(rows_n, queries, kk)
        pad = k - kk
        out[f"idx_k{k}"] = np.pad(idx, ((0, 0), (0, pad)), constant_values=-1)
        out[f"dist_k{k}"] = np.pad(dist, ((0, 0), (0, pad)), constant_values=np.inf)
        out[f"score_k{k}"] = np.pad(sc, ((0, 0), (0, pad)), constant_values=-np.inf)
    np.savez_compressed(os.path.join(OUT, "adversarial_d64.npz"), **out)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
