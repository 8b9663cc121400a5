"""CPU: the oracle's synthetic-matrix generator (oracle/csr_oracle.c, the twin of lcg_hip_csr_generate_ex) -- the three
column patterns give symmetric, strictly diagonally dominant matrices whose shards concatenate to the whole."""
import numpy as np
import pytest
import scipy.sparse as sp


@pytest.mark.parametrize("pattern,band", [(0, 0), (1, 300), (2, 300), (2, 2), (2, 40000)])
def test_patterns_are_symmetric_dominant_and_shard_consistent(port, pattern, band):
    n = 12345
    g = port.gen_init(n, 16, band, True, 11, 0.01, pattern=pattern)
    rp, ci, v = port.gen_rows(g)
    M = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert abs(M - M.T).max() == 0.0
    off = abs(M).sum(axis=1).A1 - M.diagonal()
    assert np.all(M.diagonal() - off > 0.0099)
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert np.all(np.diff(ci)[np.diff(rows) == 0] > 0)              # ascending, no duplicates
    if pattern:
        assert np.abs(ci - rows).max() <= min(band, n - 1)
    parts = [port.gen_rows(g, a, b) for a, b in ((0, 4000), (4000, 4001), (4001, n))]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ci)
    assert np.array_equal(np.concatenate([p[2] for p in parts]), v)


def test_row_random_rows_differ_and_old_entry_point_is_unchanged(port):
    n = 50000
    g = port.gen_init(n, 16, 4096, True, 3, 0.01, pattern=2)
    rp, ci, v = port.gen_rows(g, 20000, 20002)
    o0 = set((ci[rp[0]:rp[1]] - 20000).tolist()); o1 = set((ci[rp[1]:rp[2]] - 20001).tolist())
    assert len(o0 & o1) <= 3                                       # pattern 1 would give identical offset sets
    g1 = port.gen_init(n, 16, 4096, True, 3, 0.01)                 # round-1 call: band > 0 => constant diagonals
    g2 = port.gen_init(n, 16, 4096, True, 3, 0.01, pattern=1)
    a = port.gen_rows(g1, 100, 200); b = port.gen_rows(g2, 100, 200)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
