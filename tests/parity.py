"""Comparators used by the parity tests.

`assert_identical`  — the product vs the oracle: bit-exact scores and ids (the bar for the HIP path).
`compare_with_reference` — the oracle vs outputs of the REFERENCE: the reference's backend rounds ~0.2 % of
scores 1 fp16 ulp away from the correctly rounded value and orders equal scores arbitrarily (it is not even
run-to-run deterministic when threaded). The comparison therefore works row by row from the reference's own
scores around the cut: rows without a tie and without a 1-ulp difference near the cut must give IDENTICAL id
sets (lists, when the scores are distinct), and every difference in the other rows must be explained by a tie
or a 1-ulp score difference of the very id that differs. No agreement-rate thresholds.
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


def compare_with_reference(ref_scores, ref_ids, ext_scores, ext_ids, full_h, ora_scores, ora_ids):
    """Row-by-row comparison of a REFERENCE top-k with the canonical one, using the reference's own scores around the cut.

    ref_scores / ref_ids   (B, k)   what the reference returned
    ext_scores / ext_ids   (B, E)   the same reference call with E = k + 24 neighbours (tests/golden/make_golden.py): the
                                    reference's score s_ref of every id near the k-th place; ids outside it have s_ref <= ext[-1]
    full_h                 (B, N)   canonical fp16 score s* of every row (oracle);  ora_*  (B, k) the canonical top-k

    Asserted for every row (scores compared as fp16 ordinals, "ulp" = one representable step):
      (a) the reference row is sorted, has no duplicate ids, and its scores are the ext call's scores of those ids;
      (b) |s_ref - s*| <= 1 ulp for every id of the ext list (the reference rounds an fp32-accumulated sum, the canonical score
          is the correctly rounded exact product);
      (c) SHARP RULE: a row is CLEAN when the canonical scores have no tie across the k-th place, the ext list reaches at least
          2 ulp below the canonical k-th score, and s_ref == s* for every id whose s* or s_ref is within 1 ulp of it or above.
          In a clean row exactly the k canonical members have s_ref >= k-th and everything else is strictly below, so any correct
          top-k of the reference's scores IS the canonical set: the id SETS must be identical; if the k canonical scores are also
          pairwise different, the id LISTS must be identical;
      (d) in every other row each id in the symmetric difference of the two sets must be explained by its own evidence:
          s_ref != s* for it, or it sits in a canonical tie across the cut, or it sits in a tie of the reference's scores across
          the reference's cut. Anything else is a genuine disagreement.
    Returns the statistics (how many rows were clean, identical, tied ...)."""
    B, k = ref_ids.shape
    E = ext_ids.shape[1]
    ord_full = f16_ordinal(full_h)
    ord_ref, ord_ext, ord_ora = f16_ordinal(ref_scores), f16_ordinal(ext_scores), f16_ordinal(ora_scores)
    N = ord_full.shape[1]
    stats = dict(rows=B, clean_rows=0, identical_lists=0, identical_sets=0, max_ulp=0, rows_with_ulp_diff=0,
                 canonical_tie_rows=0, reference_tie_rows=0, explained_differences=0)
    for b in range(B):
        ids, xids = ref_ids[b], ext_ids[b]
        assert len(set(ids.tolist())) == k and len(set(xids.tolist())) == E, f"row {b}: duplicate ids in the reference"
        assert np.all(np.diff(ord_ref[b]) <= 0) and np.all(np.diff(ord_ext[b]) <= 0), f"row {b}: reference scores not descending"
        sref = dict(zip(xids.tolist(), ord_ext[b].tolist()))
        for j in range(k):                                                                                        # (a)
            assert sref.get(int(ids[j])) == int(ord_ref[b, j]), f"row {b}: the k and k+ext calls of the reference disagree"
        du = np.abs(ord_ext[b] - ord_full[b, xids])                                                               # (b)
        assert du.max() <= 1, f"row {b}: reference score {du.max()} ulp from canonical"
        stats["max_ulp"] = max(stats["max_ulp"], int(du.max()))
        stats["rows_with_ulp_diff"] += int(du.max() > 0)
        kth, kth_ref = int(ord_ora[b, k - 1]), int(ord_ref[b, k - 1])
        tail = int(ord_ext[b, -1]) if E < N else -(1 << 30)          # E == N: every id is in the ext list
        tie_canon = int((ord_full[b] == kth).sum()) > int((ord_ora[b] == kth).sum())
        tie_ref = (tail >= kth_ref) or int((ord_ext[b] == kth_ref).sum()) > int((ord_ref[b] == kth_ref).sum())
        stats["canonical_tie_rows"] += int(tie_canon)
        stats["reference_tie_rows"] += int(tie_ref)
        canon, refset = set(ora_ids[b].tolist()), set(ids.tolist())
        near = (ord_full[b, xids] >= kth - 1) | (ord_ext[b] >= kth - 1)
        clean = (not tie_canon) and tail <= kth - 2 and canon <= set(sref) and bool(np.all(du[near] == 0))
        if clean:                                                                                                 # (c)
            stats["clean_rows"] += 1
            assert refset == canon, f"row {b}: clean row (no tie, no 1-ulp difference near the cut) but the id sets differ: {sorted(refset ^ canon)}"
            if len(set(ord_ora[b].tolist())) == k:
                assert np.array_equal(ids, ora_ids[b]), f"row {b}: clean row with distinct scores but the id lists differ"
        else:                                                                                                     # (d)
            for g in refset ^ canon:
                explained = (g in sref and sref[g] != int(ord_full[b, g])) or (tie_canon and int(ord_full[b, g]) == kth) \
                    or (tie_ref and g in sref and sref[g] == kth_ref)
                assert explained, f"row {b}: id {g} differs between the reference and the canonical top-k with no tie / 1-ulp difference to explain it"
                stats["explained_differences"] += 1
        stats["identical_lists"] += int(np.array_equal(ids, ora_ids[b]))
        stats["identical_sets"] += int(refset == canon)
    return stats
