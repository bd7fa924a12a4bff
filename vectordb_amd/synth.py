"""Host-side facts about the synthetic generator (include/vdb.h: vdb_synth_spec;
csrc/synth.h): which corpus row a kind-2 query row is a noisy copy of.  Pure
integer arithmetic, so benchmark sanity checks need no device round trip."""
_M = (1 << 64) - 1


def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def _rowkey(seed: int, row: int) -> int:
    return _mix64((seed ^ ((row * 0xD6E8FEB86659FD93) & _M)) & _M)


def query_source_row(spec, row: int) -> int:
    """Corpus row planted as the nearest neighbour of query row `row` of a kind-2 spec."""
    base_n = int(spec.base_n) if int(spec.base_n) > 0 else 1
    return _mix64(_rowkey(int(spec.seed), int(row)) ^ 0x51C0FFEE) % base_n
