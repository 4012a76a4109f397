"""Comparators used by the parity tests.

`assert_identical`  — the product vs the oracle: bit-exact scores and ids (the bar for the HIP path).
`compare_with_reference` — the oracle vs outputs of the REFERENCE: the reference's backend rounds ~0.2 % of
scores 1 fp16 ulp away from the correctly rounded value and orders equal scores arbitrarily, so that
comparison is tie-aware and 1-ulp-aware, and says so.
"""
import numpy as np


def f16_ordinal(x: np.ndarray) -> np.ndarray:
    """fp16 -> int such that consecutive representable values differ by 1 (and -0 == +0)."""
    b = np.ascontiguousarray(x, dtype=np.float16).view(np.uint16).astype(np.int64)
    return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)


def assert_identical(got_scores, got_ids, exp_scores, exp_ids, what=""):
    gs = np.ascontiguousarray(got_scores, dtype=np.float16).view(np.uint16)
    es = np.ascontiguousarray(exp_scores, dtype=np.float16).view(np.uint16)
    gi = np.asarray(got_ids, dtype=np.int64)
    ei = np.asarray(exp_ids, dtype=np.int64)
    assert gs.shape == es.shape and gi.shape == ei.shape, (gs.shape, es.shape)
    bad_i = np.argwhere(gi != ei)
    bad_s = np.argwhere(gs != es)
    assert len(bad_i) == 0 and len(bad_s) == 0, (
        f"{what}: {len(bad_i)} id mismatches, {len(bad_s)} score-bit mismatches; first id mismatch at "
        f"{bad_i[:1].tolist()} got {gi[tuple(bad_i[0])] if len(bad_i) else None} "
        f"want {ei[tuple(bad_i[0])] if len(bad_i) else None}; first score mismatch at {bad_s[:1].tolist()}"
    )


def compare_with_reference(ref_scores, ref_ids, full_h, ora_scores, ora_ids):
    """Tie-aware / 1-ulp-aware comparison of a reference top-k with the canonical one.

    full_h: (B, N) canonical fp16 scores (oracle).  Returns a dict of statistics; raises on a violation of
      (a) every reference score is within 1 fp16 ulp of the canonical score of the id it names;
      (b) reference rows are sorted descending and have no duplicate ids;
      (c) band membership: every reference id scores (canonically) >= canonical k-th - 2 ulp, and every id whose
          canonical score is >= canonical k-th + 2 ulp is present in the reference row.
    """
    B, k = ref_ids.shape
    ord_full = f16_ordinal(full_h)
    ord_ref = f16_ordinal(ref_scores)
    stats = dict(rows=B, identical_lists=0, identical_sets=0, max_ulp=0, rows_with_ulp_diff=0, boundary_tie_rows=0)
    for b in range(B):
        ids = ref_ids[b]
        assert len(set(ids.tolist())) == k, f"row {b}: duplicate ids in reference"
        assert np.all(np.diff(ord_ref[b]) <= 0), f"row {b}: reference scores not descending"
        du = np.abs(ord_ref[b] - ord_full[b, ids])
        assert du.max() <= 1, f"row {b}: reference score {du.max()} ulp from canonical"
        stats["max_ulp"] = max(stats["max_ulp"], int(du.max()))
        stats["rows_with_ulp_diff"] += int(du.max() > 0)
        kth = f16_ordinal(ora_scores[b])[k - 1]
        assert ord_full[b, ids].min() >= kth - 2, f"row {b}: reference returned an id outside the k-th band"
        must = np.nonzero(ord_full[b] >= kth + 2)[0]
        assert np.isin(must, ids).all(), f"row {b}: reference misses an id clearly above the cut"
        stats["identical_lists"] += int(np.array_equal(ids, ora_ids[b]))
        stats["identical_sets"] += int(set(ids.tolist()) == set(ora_ids[b].tolist()))
        stats["boundary_tie_rows"] += int((ord_full[b] == kth).sum() > (f16_ordinal(ora_scores[b]) == kth).sum())
    return stats
