"""Host side of grafimo_amd/top_hits.py (no GPU): the per-region best row and the locus maximum over hit tables, the
reference's --top-graphs selection rule (res_writer.py:153-157), and the key encoding of gfm_region_best."""
import numpy as np
import pandas as pd

from grafimo_amd import top_hits as th


def _brute_best(region, scaled, rows, keep):
    out = {}
    for i in range(len(region)):
        if not keep[i]:
            continue
        cur = out.get(region[i])
        if cur is None or (scaled[i], -rows[i]) > (scaled[cur], -rows[cur]):
            out[region[i]] = i
    return sorted(out.values())


def test_best_rows_per_region_against_a_python_loop():
    rng = np.random.default_rng(5)
    for n in (0, 1, 17, 1000):
        region = rng.integers(0, 9, n)
        scaled = rng.integers(10_000, 10_020, n)      # many ties
        rows = rng.permutation(n).astype(np.int64)
        keep = rng.random(n) < 0.7
        got = th.best_rows_per_region(region, scaled, rows, keep)
        assert sorted(got.tolist()) == _brute_best(region, scaled, rows, keep)
        assert sorted(th.best_rows_per_region(region, scaled, rows).tolist()) == _brute_best(region, scaled, rows, np.ones(n, bool))


def test_locus_max_of_hits_groups_both_strands_of_a_span():
    region = np.array([0, 0, 0, 1, 1, 0])
    start = np.array([100, 119, 100, 100, 119, 101])
    stop = np.array([119, 100, 119, 119, 100, 120])     # rows 0, 1, 2: one locus of region 0 (+, -, + other allele)
    scaled = np.array([5, 9, 7, 3, 2, 8])
    assert th.locus_max_of_hits(region, start, stop, scaled).tolist() == [9, 9, 9, 3, 3, 8]
    assert len(th.locus_max_of_hits(region[:0], start[:0], stop[:0], scaled[:0])) == 0


def test_top_regions_is_the_first_n_distinct_names_of_the_report():
    df = pd.DataFrame({"sequence_name": ["b", "a", "b", "c", "a", "d"], "p-value": [1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4]})
    assert th.top_regions(df, 2) == ["b", "a"]
    assert th.top_regions(df, 10) == ["b", "a", "c", "d"]
    t = th.top_regions_table(df, 3)
    assert list(t["sequence_name"]) == ["b", "a", "c"] and list(t["p-value"]) == [1e-9, 1e-8, 1e-6]
    assert th.top_regions(df.iloc[:0], 3) == []


def test_decode_best_round_trips_the_key():
    score = np.array([0, 52, 18_111, 64_000], dtype=np.int64)
    row = np.array([0, 7, (1 << 44) - 2, 123_456_789_012], dtype=np.int64)
    key = (score << th.BEST_ROW_BITS) | (((1 << th.BEST_ROW_BITS) - 1) - row)
    key = np.concatenate([key, [0]])
    s, r, ok = th.decode_best(key)
    assert s[:4].tolist() == score.tolist() and r[:4].tolist() == row.tolist() and ok.tolist() == [True] * 4 + [False]
    assert s[4] == -1 and r[4] == -1
    # the maximum of two keys is the better score, the LOWER row among equals
    assert key[1] > ((52 << th.BEST_ROW_BITS) | (((1 << th.BEST_ROW_BITS) - 1) - 8))
