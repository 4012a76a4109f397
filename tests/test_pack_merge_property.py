"""Property test (hypothesis, CPU): the packed (score, id) exchange format orders like the canonical order -- fp16 score descending,
global passage id ascending -- for ANY scores (ties, +-0, +-inf, subnormals), shard counts and padding, and the W*k -> k merge of
packed words equals sorting the union of the shards' candidates (what src/index.py:151 computes with torch.topk, made deterministic)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from atlas_amd import index as index_mod

F16_BITS = st.integers(min_value=0, max_value=0xFFFF).filter(lambda b: (b & 0x7C00) != 0x7C00 or (b & 0x3FF) == 0)   # no NaN


def _f16(bits):
    return np.array(bits, dtype=np.uint16).view(np.float16)


@settings(max_examples=200, deadline=None)
@given(st.data())
def test_packed_order_is_canonical_order(data):
    W = data.draw(st.integers(1, 8))
    n = data.draw(st.integers(2, 40))
    bits = data.draw(st.lists(F16_BITS, min_size=n, max_size=n))
    rows = data.draw(st.lists(st.integers(0, 2 ** 30), min_size=n, max_size=n))
    ranks = data.draw(st.lists(st.integers(0, W - 1), min_size=n, max_size=n))
    s = _f16(bits)
    packed = np.array([int(index_mod.pack_candidates_host(s[i:i + 1], np.array([rows[i]], dtype=np.int64), W, ranks[i])[0]) for i in range(n)],
                      dtype=np.int64)
    gid = np.array(rows, dtype=np.int64) * W + np.array(ranks, dtype=np.int64)
    # canonical comparison: score desc (as numbers: -0 == +0), then id asc
    sv = s.astype(np.float64)
    for i in range(n):
        for j in range(n):
            better = (sv[i] > sv[j]) or (sv[i] == sv[j] and gid[i] < gid[j])
            same = (sv[i] == sv[j]) and gid[i] == gid[j]
            if better:
                assert packed[i] > packed[j], (bits[i], bits[j], gid[i], gid[j])
            if same:
                assert packed[i] == packed[j]
    us, ug = index_mod.unpack_candidates_host(packed[None, :])
    assert np.array_equal(ug[0], gid)
    assert np.array_equal(us[0].astype(np.float64), sv)            # scores survive (-0 comes back as a zero)


@settings(max_examples=100, deadline=None)
@given(st.data())
def test_merge_of_packed_words_is_the_sorted_union(data):
    W = data.draw(st.integers(1, 6))
    B = data.draw(st.integers(1, 4))
    k = data.draw(st.integers(1, 12))
    shard_rows = data.draw(st.integers(k, 64))                      # rows per shard: ids are unique inside a shard
    rng = np.random.default_rng(data.draw(st.integers(0, 2 ** 31)))
    levels = data.draw(st.integers(1, 6))                           # few distinct scores: lots of cross-shard ties
    vals = rng.standard_normal(levels).astype(np.float16)
    s = vals[rng.integers(0, levels, (W, B, k))]
    rows = np.stack([np.stack([rng.permutation(shard_rows)[:k] for _ in range(B)]) for _ in range(W)]).astype(np.int64)
    npad = data.draw(st.integers(0, k - 1))
    if npad:
        rows[W - 1, :, k - npad:] = -1                              # a short shard pads with (-inf, -1)
    packed = np.stack([index_mod.pack_candidates_host(s[w], rows[w], W, w) for w in range(W)])
    merged = index_mod.merge_packed_host(packed, k)
    ms, mg = index_mod.unpack_candidates_host(merged)
    for b in range(B):
        cand = [(float(s[w, b, j]), int(rows[w, b, j]) * W + w) for w in range(W) for j in range(k) if rows[w, b, j] >= 0]
        cand.sort(key=lambda t: (-t[0], t[1]))
        want = cand[:k]
        got = [(float(ms[b, j]), int(mg[b, j])) for j in range(k) if mg[b, j] >= 0]
        assert got == want[:len(got)] and len(got) == min(k, len(cand)), (b, got, want)
