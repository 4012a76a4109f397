"""atlas_amd.token_store.TokenStore on the host: one tokenisation with the reference's own call (src/atlas.py:66-75, incl. its
sic `max_length = min(text_maxlength, gpu_embedder_batch_size)`), nothing but real tokens stored, batches that reproduce what the
reference's per-batch tokenisation would have produced, and length bucketing that still covers every slab row exactly once."""
import numpy as np
import torch

from atlas_amd.token_store import TokenStore
from stub_tokenizer import HashTokenizer

WORDS = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
FMT = "{title} {text}"


def _passages(n, seed=3, longest=60):
    rng = np.random.default_rng(seed)
    return [{"id": str(i), "title": f"t{WORDS[i % 12]}", "text": " ".join(rng.choice(WORDS, size=int(rng.integers(1, longest))))} for i in range(n)]


def test_store_reproduces_the_reference_batches():
    passages, tok, bs = _passages(300), HashTokenizer(), 64
    store = TokenStore.from_passages(passages, tok, FMT, text_maxlength=40, gpu_embedder_batch_size=bs, chunk=97)
    assert len(store) == 300 and store.max_length == 40 and all(c["padding"] == "longest" and c["max_length"] == 40 for c in tok.calls)
    assert int(store.lengths.max()) <= 40 and int(store.lengths.min()) >= 3 and store.n_tokens == int(store.lengths.sum())
    ref_tok = HashTokenizer()
    for (rows, ids, mask), a in zip(store.batches(bs, bucket=False), range(0, 300, bs)):
        want = ref_tok([FMT.format(**p) for p in passages[a : a + bs]], padding="longest", return_tensors="pt", max_length=min(40, bs), truncation=True)
        assert rows.tolist() == list(range(a, min(300, a + bs)))
        assert torch.equal(ids, want["input_ids"]) and torch.equal(mask, want["attention_mask"])     # the batch atlas.py:68-75 builds


def test_the_sic_bound_is_kept():
    """atlas.py:74: the BATCH SIZE caps the token count (gpu_embedder_batch_size = 16 -> 16 tokens) -- reproduced, not fixed"""
    store = TokenStore.from_passages(_passages(50), HashTokenizer(), FMT, text_maxlength=512, gpu_embedder_batch_size=16)
    assert store.max_length == 16 and int(store.lengths.max()) == 16


def test_bucketed_plan_covers_every_row_once_and_pads_less():
    store = TokenStore.from_passages(_passages(1000, longest=120), HashTokenizer(), FMT, 128, 512)
    for bucket in (True, False):
        plan = store.plan(128, bucket)
        allrows = np.concatenate(plan)
        assert sorted(allrows.tolist()) == list(range(1000)) and all(len(g) <= 128 for g in plan)
    slots = lambda plan: sum(len(g) * int(store.lengths[g].max()) for g in plan)      # noqa: E731
    by_len, by_pos = slots(store.plan(128, True)), slots(store.plan(128, False))
    assert store.n_tokens <= by_len < 0.6 * by_pos
    lens_sorted = [int(store.lengths[g].max()) for g in store.plan(128, True)]
    assert lens_sorted == sorted(lens_sorted)


def test_token_budget_plan_covers_every_row_once_within_budget():
    rng = np.random.default_rng(3)
    lists = [[7] * int(n) for n in rng.integers(1, 200, size=5000)]
    store = TokenStore.from_token_lists(lists)
    for budget, bs in ((4096, 64), (65536, 512), (100, 8)):
        plan = store.plan(bs, True, budget)
        assert sorted(np.concatenate(plan).tolist()) == list(range(5000))
        for gi, g in enumerate(plan[:-1]):
            tok = int(store.lengths[g].sum())
            # a group holds as many passages as fit the budget (or one passage longer than it, or the cap of 2 x batch_size passages)
            assert tok <= budget or len(g) == 1
            assert len(g) == 2 * bs or tok + int(store.lengths[plan[gi + 1][0]]) > budget or len(g) == 1
        assert all(1 <= len(g) <= 2 * bs for g in plan)
    assert [g.tolist() for g in store.plan(64, True, 0)] == [g.tolist() for g in store.plan(64, True)]


def test_fill_matches_a_per_passage_loop():
    lists = [[5, 6, 7], [9], [1, 2, 3, 4, 5, 6], [], [8, 8]]
    store = TokenStore.from_token_lists(lists)
    rows = np.array([2, 0, 4, 1], dtype=np.int64)
    ids, mask = torch.full((4, 8), -1, dtype=torch.int64), torch.full((4, 8), -1, dtype=torch.int64)
    L = store.fill(rows, ids, mask)
    got_ids, got_mask = ids.view(-1)[: 4 * L].view(4, L), mask.view(-1)[: 4 * L].view(4, L)
    assert L == 6
    for j, r in enumerate(rows):
        assert got_ids[j, : len(lists[r])].tolist() == lists[r] and not got_ids[j, len(lists[r]):].any()
        assert got_mask[j].tolist() == [1] * len(lists[r]) + [0] * (L - len(lists[r]))


def test_refresher_plan_scales_the_token_budget_and_fits_the_staging_buffers():
    """round 6: a streamed refresh forms groups of BUDGET_SCALE x TOKEN_BUDGET tokens (refresh.IndexRefresher.plan, pure host arithmetic): every row
    once, every group within the budget and inside the staging buffers as [n, Lmax] -- also when a few long passages share a group with many
    short ones -- and a shard too small for token groups keeps the caller's passage batches"""
    import types

    from atlas_amd import refresh
    from atlas_amd.token_store import TokenStore

    rs = np.random.default_rng(3)
    for lens in (rs.integers(64, 201, size=40_000), np.concatenate([np.full(30_000, 9), np.full(50, 200)]), rs.integers(5, 30, size=700)):
        off = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        store = TokenStore(torch.zeros(int(off[-1]), dtype=torch.int32), off, 200)
        fake = types.SimpleNamespace(max_batch=512, max_len=200, stage_batch=512 * refresh.BUDGET_SCALE,
                                     index=types.SimpleNamespace(_slab=torch.empty((len(store), 1))))
        plan = refresh.IndexRefresher.plan(fake, store, 512)
        rows = np.concatenate(plan)
        assert np.array_equal(np.sort(rows), np.arange(len(store)))
        slots = fake.stage_batch * fake.max_len
        for g in plan:
            assert g.shape[0] <= 2 * fake.stage_batch and g.shape[0] * int(store.lengths[g].max()) <= slots
        if 512 * store.n_tokens / len(store) >= 0.75 * refresh.TOKEN_BUDGET:
            assert max(int(store.lengths[g].sum()) for g in plan) <= refresh.TOKEN_BUDGET * refresh.BUDGET_SCALE
            assert len(plan) <= -(-store.n_tokens // (refresh.TOKEN_BUDGET * refresh.BUDGET_SCALE)) * 2 + 2
        else:                                                        # short passages: batches of 512 PASSAGES, by length
            assert all(g.shape[0] <= 512 for g in plan)
