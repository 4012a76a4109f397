"""Parity tests proper (MI355X): the HIP path, called through the C-ABI, against the CPU oracle — bit-exact
scores and ids — plus the reference golden vectors, edge cases, and size-independent properties at 1M rows."""
import ctypes
import os

import numpy as np
import pytest
import torch

import parity
import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _index(cls, P):
    idx = cls()
    idx.init_embeddings([{"id": str(i)} for i in range(P.shape[0])], P.shape[1])
    idx.embeddings[:, :] = torch.from_numpy(P).cuda().T          # the write atlas.py:79 performs
    return idx


def _search(idx, Q, k):
    s, i = idx._compute_scores_and_indices(torch.from_numpy(Q).cuda(), k)
    return s.cpu().numpy(), i.cpu().numpy()


def test_single_hip_runtime_in_process(gpu_index_cls):
    """the extension must bind to the HIP runtime torch loaded (one libamdhip64 in the process)"""
    from atlas_amd import _lib

    _lib.lib()
    maps = open("/proc/self/maps").read()
    libs = {ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}
    assert len(libs) == 1, libs
    assert any("libatlas_hip.so" in ln for ln in maps.splitlines())


@pytest.mark.parametrize("N,B,k,ps", [(10000, 64, 40, 11), (3000, 7, 5, 21), (100000, 64, 40, 71), (4097, 1, 1, 72),
                                     (70000, 33, 128, 73), (512, 64, 256, 74), (130000, 130, 40, 75)])
def test_scan_bit_exact_vs_oracle(N, B, k, ps, gpu_index_cls, oracle_mod):
    P = synth.passages_f16(N, 768, ps)
    Q = synth.queries_f32(B, 768, ps + 1)
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, f"N={N} B={B} k={k}")
    st = idx.last_search_stats
    assert st["path"] == "scan" and st["fallback_queries"] == 0, st
    assert st["max_err_over_eps"] < 0.25, st        # measured MFMA error vs the certified bound


@pytest.mark.parametrize("N,B,k,ps", [(100000, 96, 40, 301), (100000, 97, 40, 302), (100000, 160, 40, 303), (120000, 192, 40, 304),
                                     (100000, 200, 100, 305), (70000, 512, 40, 306), (70000, 80, 256, 307), (40000, 300, 40, 308)])
def test_batches_above_64_queries_bit_exact_vs_oracle(N, B, k, ps, gpu_index_cls, oracle_mod):
    """a rank of an N-GPU search scores ALL gathered queries (index.py:127-131): one 96-query pass up to 96 queries, GEMM-shaped passes
    (csrc/gscan_kernel.h) above that on shards of >= 65 536 rows (97 -> one 256-query tile, 512 -> two column tiles, ...), round 3's 64- /
    96-query passes on the 40k-row shard"""
    P = synth.passages_f16(N, 768, ps)
    Q = synth.queries_f32(B, 768, ps + 1)
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, f"N={N} B={B} k={k}")
    st = idx.last_search_stats
    # (k = 256 on a 70k-row shard: the k-th of the 256 first-tile maxima is no threshold at all, every row is a candidate and some queries'
    #  bands outgrow the merge -- they take the exact path, with 64-query passes as well: correct, and not what this test is about)
    assert st["path"] == "scan" and (k > 128 or st["fallback_queries"] == 0), st
    # same again with duplicated passages in the shard: the flush / compaction path of the wide pass's small candidate buffer
    P2 = np.concatenate([P[: N // 2], P[: N // 2]])
    idx2 = _index(gpu_index_cls, P2)
    s2, i2 = _search(idx2, Q[:B], k)
    es2, ei2 = oracle_mod.search(oracle_mod.f32_to_f16(Q), P2, k)
    parity.assert_identical(s2, i2, es2, ei2, f"duplicated halves, N={N} B={B} k={k}")


@pytest.mark.parametrize("N,B,k,ps,certify", [(150000, 128, 40, 321, 0), (100000, 256, 100, 322, 0), (70000, 512, 256, 323, 0), (66000, 300, 40, 324, 0),
                                             (150000, 128, 100, 325, 1), (100000, 384, 40, 326, 1), (70000, 512, 256, 327, 1),
                                             (120000, 97, 40, 328, 0), (90000, 113, 100, 329, 1), (80000, 320, 40, 330, 0),
                                             (110000, 192, 40, 331, 0), (90000, 160, 100, 332, 1), (80000, 384, 40, 333, 0), (70000, 700, 40, 334, 1)])
def test_gemm_shaped_passes_bit_exact_vs_oracle(N, B, k, ps, certify, gpu_index_cls, oracle_mod):
    """batches above 96 queries on the GEMM-shaped passes (one / two column tiles of 256 queries, partial tiles, the tail range shorter than
    the others, k = 40 / 100 / 256), the twin that trusts pmax and the one that measures every row norm (certify_every = 1: the C-ABI's
    default mode) -- and the same shard with duplicated halves (every score twice: twice the candidates, ties across the cut)"""
    P = synth.passages_f16(N, 768, ps)
    Q = synth.queries_f32(B, 768, ps + 1)
    for dup in (False, True):
        Pd = np.concatenate([P[: N // 2], P[: N // 2]]) if dup else P
        idx = gpu_index_cls(certify_every=1) if certify else gpu_index_cls()
        idx.init_embeddings([{"id": str(i)} for i in range(N)], 768)
        idx.embeddings[:, :] = torch.from_numpy(Pd).cuda().T
        s, i = _search(idx, Q, k)
        es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), Pd, k)
        parity.assert_identical(s, i, es, ei, f"N={N} B={B} k={k} certify={certify} dup={dup}")
        st = idx.last_search_stats
        assert st["path"] == "scan" and st["plan"]["gemm_passes"] >= 1 and st["pmax_trusted"] is (not certify), st
        assert st["max_err_over_eps"] < 0.25 and (k > 128 or st["fallback_queries"] == 0), st


@pytest.mark.parametrize("N,B,k,ps", [(65536 + 17, 1100, 7, 341), (65536, 97, 1, 342), (90001, 1024, 40, 343), (131072 + 255, 600, 33, 344)])
def test_gemm_shaped_passes_at_their_edges(N, B, k, ps, gpu_index_cls, oracle_mod):
    """edges of the GEMM-shaped pass: the smallest shard that takes it (65 536 rows: 256 tiles, half of them the sample), a last tile of 17 / 145 /
    255 rows, more queries than one pass takes (1100 = 1024 + a second pass), four column tiles, k = 1"""
    P = synth.passages_f16(N, 768, ps)
    Q = synth.queries_f32(B, 768, ps + 1)
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, f"N={N} B={B} k={k}")
    st = idx.last_search_stats
    assert st["path"] == "scan" and st["fallback_queries"] == 0 and st["plan"]["gemm_passes"] >= 1 and sum(st["plan"].values()) >= (2 if B > 1024 else 1), st


def test_gemm_shaped_pass_with_mass_ties_falls_back_per_query(gpu_index_cls, oracle_mod):
    """3 000 copies of one passage straddle the cut of the queries that like it: their candidate lists overflow the merge's band -- those queries take
    the exact path (flagged per query), the others stay on the scan; same canonical result (ties: lowest row first)"""
    N, B, k = 100000, 160, 40
    P = synth.passages_f16(N, 768, 351)
    P[20000:23000] = P[777]
    Q = synth.queries_f32(B, 768, 352)
    Q[:5] = P[777].astype(np.float32) * 3.0 + 0.01 * Q[:5]              # five queries whose best passage is the duplicated one
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, "mass ties under the GEMM-shaped pass")
    st = idx.last_search_stats
    assert st["plan"]["gemm_passes"] >= 1 and 1 <= st["fallback_queries"] <= 16, st


def test_gemm_shaped_pass_finds_a_row_that_violates_the_hint(gpu_index_cls, oracle_mod):
    """the certifying twin of the GEMM-shaped pass measures every row: a 9 x row raises ATLAS_F_PMAX_VIOLATION, the search is repeated
    with the measured bound (the C-ABI contract of atlas_scan_topk), same canonical result"""
    N, B, k = 120000, 200, 40
    P = synth.passages_f16(N, 768, 331)
    Q = synth.queries_f32(B, 768, 332)
    P[100001] = (P[100001].astype(np.float32) * 9.0).astype(np.float16)
    idx = gpu_index_cls(certify_every=1)
    idx.init_embeddings([{"id": str(i)} for i in range(N)], 768)
    idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
    idx._pmax, idx._pmax_version = 1.002, idx._slab_version()        # a stale hint: what a caller of the C-ABI without a certificate passes
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, "a 9 x row, GEMM-shaped certifying pass")
    st = idx.last_search_stats
    assert st["reruns"] == 1 and st["plan"]["gemm_passes"] >= 1 and st["pmax"] > 8.0, st


def test_wide_passes_of_the_certifying_twin(gpu_index_cls, oracle_mod):
    """batches above 64 queries through the scan that measures every row norm itself (certify_every = 1: the C-ABI's default mode, the
    96-query instantiation of the certifying twin), incl. a row that violates the hint: one certifying rerun, same canonical result"""
    N, B, k = 150000, 160, 40
    P = synth.passages_f16(N, 768, 311)
    Q = synth.queries_f32(B, 768, 312)
    idx = gpu_index_cls(certify_every=1)
    idx.init_embeddings([{"id": str(i)} for i in range(N)], 768)
    idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    for _ in range(2):
        s, i = _search(idx, Q, k)
        parity.assert_identical(s, i, es, ei, "certifying, wide passes")
        assert idx.last_search_stats["pmax_trusted"] is False and idx.last_search_stats["fallback_queries"] == 0
    P2 = P.copy()
    P2[77777] = (P2[77777].astype(np.float32) * 9.0).astype(np.float16)
    idx._slab[77777] = torch.from_numpy(P2[77777]).cuda()            # (a write torch sees: the bound is re-measured before the scan)
    es2, ei2 = oracle_mod.search(oracle_mod.f32_to_f16(Q), P2, k)
    s2, i2 = _search(idx, Q, k)
    parity.assert_identical(s2, i2, es2, ei2, "certifying, wide passes, a 9 x row")


@pytest.mark.parametrize("case", ["a10k", "b3k", "c_dups", "d_k128"])
def test_scan_vs_reference_golden(case, gpu_index_cls, oracle_mod):
    """HIP path vs the outputs of the reference's own DistributedIndex (tie-/1-ulp-aware, see parity.py)."""
    g = np.load(os.path.join(G, f"{case}.npz"))
    N, B, dup, k = int(g["N"]), int(g["B"]), int(g["dup"]), int(g["k"])
    P = synth.passages_f16(N // dup, 768, int(g["ps"]))
    if dup > 1:
        P = np.tile(P, (dup, 1))
    Q = synth.queries_f32(B, 768, int(g["qs"]))
    assert synth.sha(P, Q) == str(g["sha"])
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei, full = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k, return_full=True)
    parity.assert_identical(s, i, es, ei, case)
    st = parity.compare_with_reference(g["ref_scores"], g["ref_ids"], g["ext_scores"], g["ext_ids"], full, s, i)
    assert st["max_ulp"] <= 1 and (case == "c_dups" or st["clean_rows"] >= 1)


def test_query_dtypes_match_half_cast(gpu_index_cls, oracle_mod):
    P = synth.passages_f16(5000, 768, 81)
    Q = synth.queries_f32(9, 768, 82) * 3.0
    idx = _index(gpu_index_cls, P)
    q = torch.from_numpy(Q).cuda()
    for qq in (q, q.half(), q.bfloat16()):
        s, i = idx._compute_scores_and_indices(qq, 10)
        ref_q = qq.half().cpu().numpy()                   # what `.half()` gives (src/index.py:117)
        es, ei = oracle_mod.search(ref_q, P, 10)
        parity.assert_identical(s.cpu().numpy(), i.cpu().numpy(), es, ei, str(qq.dtype))


def test_mass_ties(gpu_index_cls, oracle_mod):
    """Many identical rows straddling the cut. 1000 of them fit the candidate band (all are rescored, the lowest
    ids win); 3000 overflow it, the query is flagged and redone on the exact path. Same canonical answer."""
    for ndup, expect_fallback in ((1000, False), (3000, True)):
        P = synth.passages_f16(6000, 768, 83)
        P[500 : 500 + ndup] = P[7]
        Q = P[7:8].astype(np.float32) * 30.0                  # the duplicated row is the best match
        Q = np.concatenate([Q, synth.queries_f32(3, 768, 84)])
        idx = _index(gpu_index_cls, P)
        s, i = _search(idx, Q, 40)
        es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 40)
        parity.assert_identical(s, i, es, ei, f"mass ties {ndup}")
        assert i[0, 0] == 7 and i[0, 1:40].tolist() == list(range(500, 539))
        assert (idx.last_search_stats["fallback_queries"] >= 1) == expect_fallback, idx.last_search_stats


def test_degenerate_slabs(gpu_index_cls, oracle_mod):
    Q = synth.queries_f32(5, 768, 85)
    for P in (np.zeros((300, 768), np.float16), -np.abs(synth.passages_f16(300, 768, 86)) * np.sign(Q[0]).astype(np.float16)):
        idx = _index(gpu_index_cls, P)
        s, i = _search(idx, Q, 7)
        es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 7)
        parity.assert_identical(s, i, es, ei, "degenerate")
    P = synth.passages_f16(17, 768, 87)                    # shard smaller than one tile; k == N
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, 17)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 17)
    parity.assert_identical(s, i, es, ei, "tiny")
    with pytest.raises(RuntimeError, match="out of range"):   # torch.topk's contract (index.py:118)
        _search(idx, Q, 18)


def test_large_norm_row_triggers_recertification(gpu_index_cls, oracle_mod):
    """the scan's error margin is certified with the max row norm; a later, larger row must be noticed"""
    P = synth.passages_f16(20000, 768, 88)
    Q = synth.queries_f32(8, 768, 89)
    idx = _index(gpu_index_cls, P)
    _search(idx, Q, 10)
    p_before = idx._pmax
    big = (synth.passages_f16(1, 768, 90).astype(np.float32) * 50).astype(np.float16)
    idx.embeddings[:, 12345:12346] = torch.from_numpy(big).cuda().T
    P[12345] = big[0]
    s, i = _search(idx, Q, 10)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 10)
    parity.assert_identical(s, i, es, ei, "after big row")
    # the write went through torch (atlas.py:79's slice assignment): the version counter moved, the bound was measured again before
    # the scan (no violation / re-run protocol needed), and the scan trusted it
    st = idx.last_search_stats
    assert st["reruns"] == 0 and st["pmax_trusted"] and idx._pmax > 10 * p_before
    # a writer torch does not see (raw pointer): the stale bound is trusted until invalidate_pmax() -- the documented contract
    v = idx._slab_version()
    idx.invalidate_pmax()
    _search(idx, Q, 10)
    assert idx._pmax_version == v and idx.last_search_stats["pmax_trusted"]


def test_writes_behind_the_version_counter_are_caught(gpu_index_cls, oracle_mod):
    """`index.embeddings.data[...] = x` (or any alias torch does not track) does not move the slab's version counter, so the bound the
    scan trusts goes stale. Two nets: the merge holds every row it rescans to the trusted bound -- a long row
    that reaches the top-k candidates raises ATLAS_F_PMAX_VIOLATION and the search re-certifies from a fresh measurement -- and every
    certify_every-th search runs the certifying scan, which measures every row. A search never returns a result that differs from
    the oracle's without one of them having fired."""
    P = synth.passages_f16(30000, 768, 188)
    Q = synth.queries_f32(8, 768, 189)
    q16 = oracle_mod.f32_to_f16(Q)

    # (1) a 50x row ALIGNED with query 0, written through .data: it is query 0's best passage, the merge rescans it and notices
    idx = _index(gpu_index_cls, P)
    _search(idx, Q, 10)
    v0, p0 = idx._slab_version(), idx._pmax
    big = (q16[0].astype(np.float32) / np.linalg.norm(q16[0].astype(np.float32)) * 50).astype(np.float16)
    idx.embeddings.data[:, 777:778] = torch.from_numpy(big).cuda()[:, None]
    assert idx._slab_version() == v0                           # torch did not see the write
    P1 = P.copy(); P1[777] = big
    s, i = _search(idx, Q, 10)
    es, ei = oracle_mod.search(q16, P1, 10)
    parity.assert_identical(s, i, es, ei, "after an untracked aligned write")
    st = idx.last_search_stats
    assert i[0, 0] == 777 and st["reruns"] == 1 and not st["pmax_trusted"] and idx._pmax > 10 * p0, st

    # (2) a 50x row ANTI-aligned with every query (never a candidate): only a certifying scan can see it
    anti = (-(q16.astype(np.float32).sum(axis=0)) / np.linalg.norm(q16.astype(np.float32).sum(axis=0)) * 50).astype(np.float16)
    for every, expect_rerun_at in ((1, 1), (3, 3)):
        idx = gpu_index_cls(certify_every=every)
        idx.init_embeddings([{"id": str(j)} for j in range(P.shape[0])], 768)
        idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
        _search(idx, Q, 10)                                    # measures the bound; counts as search 1 of the period
        idx.embeddings.data[:, 4242:4243] = torch.from_numpy(anti).cuda()[:, None]
        P2 = P.copy(); P2[4242] = anti
        es, ei = oracle_mod.search(q16, P2, 10)
        fired = None
        for n in range(1, 5):
            s, i = _search(idx, Q, 10)
            parity.assert_identical(s, i, es, ei, f"certify_every={every}, search {n} after the write")
            if idx.last_search_stats["reruns"] and fired is None:
                fired = n
        assert fired is not None and fired <= expect_rerun_at, (every, fired)
        assert idx._pmax > 10 * p0


@pytest.mark.parametrize("N", [1, 2, 3, 7, 64, 1001, 4096, 65537])
def test_slab_pmax_is_a_tight_upper_bound_wherever_the_largest_row_is(N, gpu_index_cls):
    """the scan TRUSTS this number (ATLAS_SCAN_TRUST_PMAX): it must be >= every row's norm -- first / last row, first / second row of
    the pairs the d = 768 kernel walks, odd N, a row whose weight sits in the half of the middle load that belongs to it -- and tight
    (rounded up by 0.1 %), also for a slab that starts at an odd row of a larger tensor and for another d"""
    g = torch.Generator(device="cuda").manual_seed(N)
    base = torch.randn((N + 1, 768), generator=g, device="cuda") * 0.03
    for where in sorted({0, N - 1, N // 2, max(N // 2 - 1, 0), min(1, N - 1)}):
        for cols in (slice(0, 768), slice(500, 520), slice(760, 768), slice(0, 8)):      # where in the row the weight sits
            x = base.clone()
            x[1 + where, cols] = 3.0
            whole = x.half()
            slab = whole[1:]                                     # starts 1536 B into the allocation
            idx = gpu_index_cls()
            idx._set_slab(slab)
            true = float(slab.float().norm(dim=1).max())
            got = idx.slab_pmax()
            assert true * 0.99999 <= got <= true * 1.0015, (N, where, cols, true, got)
    other = (torch.randn((N, 96), generator=g, device="cuda")).half()
    idx = gpu_index_cls()
    idx._set_slab(other)
    true = float(other.float().norm(dim=1).max())
    assert true * 0.99999 <= idx.slab_pmax() <= true * 1.0015


def test_certifying_and_trusting_scans_agree_and_the_certifying_one_notices_a_large_row(gpu_index_cls):
    """the two modes of the C-ABI (atlas_scan_topk = certifying: measures every row's norm, reports a violation of pmax_hint;
    atlas_scan_topk_flags + ATLAS_SCAN_TRUST_PMAX: takes the bound as certified): same ids and score bits; a too-small hint raises
    ATLAS_F_PMAX_VIOLATION with the measured maximum in the certifying call only"""
    from atlas_amd import _lib

    L = _lib.lib()
    N, B, k = 300_000, 64, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    slab = torch.randn((N, 768), generator=g, device="cuda")
    slab = (slab / slab.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, 768), generator=g, device="cuda")
    outs = {}
    for name, fn, extra in (("certify", L.atlas_scan_topk, ()), ("trust", L.atlas_scan_topk_flags, (None, None, _lib.SCAN_TRUST_PMAX))):
        ws = torch.zeros(int(L.atlas_scan_topk_workspace_bytes(N, B, 768, k)), dtype=torch.uint8, device="cuda")
        o_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); o_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
        o_st = torch.zeros(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
        assert fn(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, 768, k, 1.002, o_s.data_ptr(), o_i.data_ptr(), o_st.data_ptr(),
                  ws.data_ptr(), ws.numel(), None, *extra) == 0
        torch.cuda.synchronize()
        outs[name] = (o_s.clone(), o_i.clone(), o_st.cpu().numpy().copy())
    assert torch.equal(outs["certify"][0], outs["trust"][0]) and torch.equal(outs["certify"][1], outs["trust"][1])
    assert int(outs["certify"][2][_lib.ST_FLAGS]) == 0 and int(outs["trust"][2][_lib.ST_FLAGS]) == 0
    measured = float(outs["certify"][2][_lib.ST_PMAX_BITS:_lib.ST_PMAX_BITS + 1].view(np.float32)[0])
    assert 0.99 < measured < 1.01 and int(outs["trust"][2][_lib.ST_PMAX_BITS]) == 0
    # a hint below the true maximum: only the certifying scan can tell
    ws = torch.zeros(int(L.atlas_scan_topk_workspace_bytes(N, B, 768, k)), dtype=torch.uint8, device="cuda")
    o_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); o_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    o_st = torch.zeros(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    assert L.atlas_scan_topk(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, 768, k, 0.5, o_s.data_ptr(), o_i.data_ptr(), o_st.data_ptr(),
                             ws.data_ptr(), ws.numel(), None) == 0
    st = o_st.cpu().numpy()
    assert int(st[_lib.ST_FLAGS]) & _lib.F_PMAX_VIOLATION and 0.99 < float(st[_lib.ST_PMAX_BITS:_lib.ST_PMAX_BITS + 1].view(np.float32)[0]) < 1.01
    assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, 768, k, 1.0, o_s.data_ptr(), o_i.data_ptr(), o_st.data_ptr(),
                                   ws.data_ptr(), ws.numel(), None, None, None, 2) == _lib.E_BADARG


@pytest.mark.parametrize("d,k", [(96, 300), (768, 1000), (1000, 10)])
def test_exact_path_any_shape(d, k, gpu_index_cls, oracle_mod):
    P = synth.passages_f16(4000, d, 91)
    Q = synth.queries_f32(4, d, 92)
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, f"exact d={d} k={k}")
    assert idx.last_search_stats["path"] == "exact"


def test_search_knn_returns_reference_types(gpu_index_cls, oracle_mod):
    P = synth.passages_f16(2000, 768, 93)
    Q = synth.queries_f32(3, 768, 94)
    idx = gpu_index_cls()
    passages = [{"id": str(i), "title": f"t{i}", "text": f"x{i}"} for i in range(2000)]
    idx.init_embeddings(passages)
    idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
    docs, scores = idx.search_knn(torch.from_numpy(Q).cuda(), 6)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 6)
    assert [[d["id"] for d in row] for row in docs] == [[str(x) for x in row] for row in ei.tolist()]
    assert docs[0][0] is passages[ei[0, 0]] and isinstance(scores[0][0], float)
    assert scores == es.astype(np.float32).tolist()


def test_device_f64_to_f16_conversions(gpu_index_cls, oracle_mod):
    """both device double->fp16 paths (bit-level, and hardware round-to-odd + v_cvt_f16_f32) are single RNE roundings"""
    from atlas_amd import _lib

    L = _lib.lib(tuning=True)           # (the conversion hook is a test hook: tuning build only; the conversions are common.h's, shared by both builds)
    rng = np.random.default_rng(7)
    allh = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = (allh[:-1] + allh[1:]) / 2
    x = np.concatenate([mids, np.nextafter(mids, 0), np.nextafter(mids, np.inf), allh,
                        mids * (1 + 2.0 ** -30), mids * (1 - 2.0 ** -30),            # inside one float ulp of a tie
                        rng.standard_normal(200000) * np.exp(rng.uniform(-25, 13, 200000)),
                        [0.0, 65504.0, 65519.999, 65520.0, 65536.0, 1e300, 2.0 ** -25, 2.0 ** -25 * (1 + 1e-12), 5e-324, np.inf]])
    x = np.concatenate([x, -x])
    xd = torch.from_numpy(x).cuda()
    out = torch.empty(2 * x.size, dtype=torch.int16, device="cuda")
    L.atlas_dbg_f64_to_f16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert L.atlas_dbg_f64_to_f16(xd.data_ptr(), out.data_ptr(), x.size, None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16).reshape(-1, 2)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got[:, 0], want), np.argwhere(got[:, 0] != want)[:5]
    assert np.array_equal(got[:, 1], want), (np.argwhere(got[:, 1] != want)[:5], x[got[:, 1] != want][:5])


def test_empty_query_batch(gpu_index_cls):
    P = synth.passages_f16(1000, 768, 96)
    idx = _index(gpu_index_cls, P)
    docs, scores = idx.search_knn(torch.empty((0, 768), device="cuda"), 5)     # atlas.py:106
    assert docs == [] and scores == []


def test_pack_merge_kernels_match_host(gpu_index_cls):
    from atlas_amd import _lib, index as im

    L = _lib.lib()
    rng = np.random.default_rng(3)
    W, B, k = 8, 64, 40
    s = rng.standard_normal((W, B, k)).astype(np.float16)
    s[:, 0, :] = s[0, 0, :]                                  # ties across shards
    rows = rng.integers(0, 4_000_000, (W, B, k)).astype(np.int64)
    rows[3, 5, 30:] = -1
    packed_h = np.stack([im.pack_candidates_host(s[w], rows[w], W, w) for w in range(W)])
    sd, rd = torch.from_numpy(s).cuda(), torch.from_numpy(rows).cuda()
    packed_d = torch.empty((W, B, k), dtype=torch.int64, device="cuda")
    for w in range(W):
        _lib.check(L.atlas_pack_candidates(sd[w].data_ptr(), rd[w].data_ptr(), B * k, W, w, packed_d[w].data_ptr(), None), "pack")
    torch.cuda.synchronize()
    assert np.array_equal(packed_d.cpu().numpy(), packed_h)
    out = torch.empty((B, k), dtype=torch.int64, device="cuda")
    _lib.check(L.atlas_merge_packed(packed_d.data_ptr(), W, B, k, out.data_ptr(), None), "merge")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), im.merge_packed_host(packed_h, k))


def test_pool_write_matches_oracle(gpu_index_cls, oracle_mod):
    from atlas_amd import _lib

    L = _lib.lib()
    n, Lq, d = 37, 50, 768
    H = (synth.normal_f32(n * Lq, d, 95, 0.7)).astype(np.float16).reshape(n, Lq, d)
    lens = np.random.default_rng(4).integers(1, Lq + 1, n)
    mask = (np.arange(Lq)[None, :] < lens[:, None]).astype(np.int64)
    H[0, lens[0]:, :] = np.float16(np.nan)                   # masked positions must not leak (masked_fill, retrievers.py:50)
    slab = torch.zeros((100, d), dtype=torch.float16, device="cuda")
    hd, md = torch.from_numpy(H).cuda(), torch.from_numpy(mask).cuda()
    _lib.check(L.atlas_pool_write(hd.data_ptr(), md.data_ptr(), slab.data_ptr(), 100, 20, n, Lq, d, None), "pool_write")
    torch.cuda.synchronize()
    exp = oracle_mod.pool(H, mask)
    got = slab.cpu().numpy()
    assert np.array_equal(got[20:57].view(np.uint16), exp.view(np.uint16))
    assert not got[:20].any() and not got[57:].any()
    # and the torch formulation the reference runs (fp16 copy on the GPU) agrees to <= 1 fp16 ulp
    lh = hd.masked_fill(~md[..., None].bool(), 0.0)
    ref = (lh.sum(dim=1) / md.sum(dim=1)[..., None]).cpu().numpy()
    assert np.abs(parity.f16_ordinal(ref) - parity.f16_ordinal(exp)).max() <= 1


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] size (1M x 768, 64 queries, top-40): size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(gpu_index_cls):
    g = torch.Generator(device="cuda").manual_seed(1234)
    N = 1_000_000
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 100_000):
        x = torch.randn((100_000, 768), generator=g, device="cuda")
        slab[r0 : r0 + 100_000] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((64, 768), generator=g, device="cuda")
    idx = gpu_index_cls()
    idx.init_embeddings([None] * 0)
    idx._set_slab(slab)
    idx.doc_map = {}
    return idx, slab, q


def test_1m_scan_equals_exact_path(big):
    """the MFMA fast path and the MFMA-free exact path are independent implementations of the same result: ALL 64 queries of
    BASELINE configs[1], ids and score bits"""
    idx, slab, q = big
    s, i = idx._compute_scores_and_indices(q, 40)
    assert idx.last_search_stats["path"] == "scan" and idx.last_search_stats["fallback_queries"] == 0
    assert idx.last_search_stats["max_err_over_eps"] < 0.25
    es, ei = idx._exact_topk(q, 40)
    assert torch.equal(s, es) and torch.equal(i, ei)
    # scores really are the rounded inner products of the rows returned (fp64 on device)
    sub = slab[i[:4].reshape(-1)].double().view(4, 40, 768)
    dots = torch.einsum("bkd,bd->bk", sub, q[:4].half().double())
    # fp64 -> fp16 with numpy (one rounding); torch's double->half rounds through float first
    assert np.array_equal(dots.cpu().numpy().astype(np.float16).view(np.uint16), s[:4].cpu().numpy().view(np.uint16))
    assert (s[:, :-1] >= s[:, 1:]).all()


@pytest.mark.parametrize("rows", [524_288, 524_527, 700_001, 1_000_000])      # 8 tiles per workgroup exactly (the smallest pooled shard), + 239 rows, ragged, 1M
def test_tail_pool_hand_out_does_not_change_the_result(rows, big):
    """The tail of the slab is handed out to the scan's workgroups at run time, a tile at a time (scan_kernel.h: fill_next_tile;
    who scans which pool tile differs from call to call). Through the tuning build of the same sources, with the pool switched off,
    at its product setting, and covering half / nearly all of the slab: every configuration returns the exact path's ids and score
    bits for all 64 queries, three calls in a row on one workspace (the ticket counter is put back by the merge kernel)."""
    import ctypes
    from atlas_amd import _lib

    idx, slab, q = big
    T = _lib.lib(tuning=True)
    T.atlas_tune_set_scan_pool.argtypes, T.atlas_tune_set_scan_pool.restype = [ctypes.c_int, ctypes.c_int], None
    N, B, k = rows, 64, 40
    sub = slab[:N]
    es, ei = _index_of(idx.__class__, sub)._exact_topk(q, k)
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda")
    out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.zeros(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    try:
        for permille, cap in ((0, 16), (60, 16), (500, 64), (950, 255)):
            T.atlas_tune_set_scan_pool(permille, cap)
            ws = torch.zeros(int(T.atlas_scan_topk_workspace_bytes(N, B, 768, k)), dtype=torch.uint8, device="cuda")
            for rep in range(3):
                out_s.zero_(); out_i.zero_()
                rc = T.atlas_scan_topk(q.data_ptr(), _lib.DT_F32, sub.data_ptr(), N, B, 768, k, 1.001, out_s.data_ptr(), out_i.data_ptr(),
                                       out_st.data_ptr(), ws.data_ptr(), ws.numel(), None)
                assert rc == 0, rc
                torch.cuda.synchronize()
                st = out_st.cpu().numpy()
                assert int(st[_lib.ST_FLAGS]) == 0 and int(st[_lib.ST_N_FALLBACK]) == 0, (permille, rep, st[:8])
                assert torch.equal(out_s, es) and torch.equal(out_i, ei), (permille, cap, rep)
    finally:
        T.atlas_tune_set_scan_pool(60, 32)


def _index_of(cls, slab):
    sh = cls()
    sh._set_slab(slab)
    return sh


def test_1m_sharding_invariance(big, gpu_index_cls):
    """top-k of 8 round-robin shards, packed + merged, is identical to the single-shard result"""
    from atlas_amd import index as im

    idx, slab, q = big
    s, i = idx._compute_scores_and_indices(q, 40)
    W = 8
    packed = []
    for r in range(W):
        sh = gpu_index_cls()
        sh._set_slab(slab[r::W].contiguous())
        ss, ii = sh._compute_scores_and_indices(q, 40)
        packed.append(im.pack_candidates_host(ss.cpu().numpy(), ii.cpu().numpy(), W, r))
    merged = idx._merge(torch.from_numpy(np.stack(packed)).cuda(), 40)
    ms, mg = im.unpack_candidates_host(merged)
    parity.assert_identical(ms, mg, s.cpu().numpy(), i.cpu().numpy(), "8 shards vs 1")


def test_1m_oracle_subset(big, oracle_mod):
    """8 queries against the CPU oracle at the full 1M size (the other 56 are tied to the same canonical result through the
    exact path above, which this test also holds to the oracle)"""
    idx, slab, q = big
    sel = [0, 9, 18, 27, 36, 45, 54, 63]
    s, i = idx._compute_scores_and_indices(q[sel], 40)
    es, ei = oracle_mod.search(q[sel].half().cpu().numpy(), slab.cpu().numpy(), 40)
    parity.assert_identical(s.cpu().numpy(), i.cpu().numpy(), es, ei, "1M oracle")
    xs, xi = idx._exact_topk(q[sel], 40)
    parity.assert_identical(xs.cpu().numpy(), xi.cpu().numpy(), es, ei, "1M exact path vs oracle")


def test_duplicated_passages_at_1m_stay_bounded(big, gpu_index_cls):
    """10 000 copies of one passage inside a 1M-row shard (boilerplate duplicates): the queries that hit them overflow the
    candidate band and are redone by the batched exact path. Same canonical answer (lowest ids first) in bounded time: one slab
    pass per 8 flagged queries, not eleven launches and nine passes per query."""
    import time

    idx, slab, q = big
    dup = slab.clone()
    rows = torch.arange(200_000, 210_000, device="cuda")
    dup[rows] = dup[77].clone()
    qq = q.clone()
    qq[:24] = dup[77].float()[None, :] * (1.0 + torch.arange(24, device="cuda")[:, None] * 0.01)    # 24 queries whose best match is the copy
    sh = gpu_index_cls()
    sh._set_slab(dup)
    sh._compute_scores_and_indices(qq, 40)                       # warm-up (workspaces, pmax)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, i = sh._compute_scores_and_indices(qq, 40)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = sh.last_search_stats
    assert st["fallback_queries"] >= 24, st
    assert i[0, 0] == 77 and i[0, 1:40].tolist() == list(range(200_000, 200_039))
    es, ei = sh._exact_topk(qq, 40)
    assert torch.equal(s, es) and torch.equal(i, ei)
    print(f"1M rows, {st['fallback_queries']} fallback queries: {dt * 1e3:.1f} ms")
    assert dt < 0.25, f"{dt:.3f} s for a search with {st['fallback_queries']} flagged queries"


def test_dma_staged_pass_random_cases_equal_the_exact_path_and_the_register_fed_kernel(gpu_index_cls):
    """dscan_kernel.h (round 6: the 64-query pass with the slab through LDS-DMA and the queries in registers) against scan_kernel.h (what rounds 1-5
    ran) and against the MFMA-free exact path: random shard sizes from the 65 536-row minimum of the coop exchange up (ragged last tiles, shards
    whose last workgroups have no rows, pooled and unpooled splits), 1..64 queries, k, score scales, duplicated rows, the three query dtypes, both
    twins, two calls per workspace, the DMA kernel with its static tiles dealt (the product) and in contiguous ranges -- ids and score bits equal, no flags."""
    import ctypes
    from atlas_amd import _lib

    T = _lib.lib(tuning=True)
    T.atlas_tune_set_scan_dma.argtypes, T.atlas_tune_set_scan_dma.restype = [ctypes.c_int], None
    T.atlas_tune_set_dma_deal.argtypes, T.atlas_tune_set_dma_deal.restype = [ctypes.c_int], None
    rng = np.random.default_rng(2026)
    g = torch.Generator(device="cuda").manual_seed(2027)
    dts = [(torch.float32, _lib.DT_F32), (torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)]
    try:
        for c in range(18):
            N = int([65536, 65536 + int(rng.integers(1, 256)), int(rng.integers(66000, 140000)), int(rng.integers(140000, 524288)), 524288 + int(rng.integers(0, 300)),
                     int(rng.integers(524288, 1500000))][c % 6])
            B = int(rng.choice([1, 7, 16, 17, 40, 63, 64, 64]))
            k = int(rng.choice([1, 5, 40, 40, 100, 256]))
            scale = float(rng.choice([1.0, 1.0, 0.05, 4.0]))
            slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
            for r0 in range(0, N, 200_000):
                n = min(200_000, N - r0)
                x = torch.randn((n, 768), generator=g, device="cuda")
                slab[r0 : r0 + n] = (x / x.norm(dim=1, keepdim=True) * scale).half()
            if c % 3 == 0:
                slab[N // 2 : N // 2 + 1000] = slab[:1000]
            tdt, code = dts[c % 3]
            q = (torch.randn((B, 768), generator=g, device="cuda") * float(rng.choice([1.0, 0.3, 3.0]))).to(tdt)
            flags = _lib.SCAN_TRUST_PMAX if c % 2 == 0 else 0
            pmax = float(slab.float().norm(dim=1).max()) * 1.001
            ref = _index_of(gpu_index_cls, slab)
            es, ei = ref._exact_topk(q, k)
            ws = torch.zeros(int(T.atlas_scan_topk_workspace_bytes(N, B, 768, k)), dtype=torch.uint8, device="cuda")
            got = {}
            for mode in (1, 0, 3):                  # 1 = dscan_kernel, its static tiles dealt (the product); 0 = scan_kernel; 3 = dscan_kernel with one contiguous range per workgroup
                T.atlas_tune_set_scan_dma(1 if mode == 3 else mode)
                T.atlas_tune_set_dma_deal(0 if mode == 3 else 1)
                for rep in range(2):
                    out_s = torch.zeros((B, k), dtype=torch.float16, device="cuda")
                    out_i = torch.zeros((B, k), dtype=torch.int64, device="cuda")
                    out_st = torch.zeros(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
                    rc = T.atlas_scan_topk_flags(q.data_ptr(), code, slab.data_ptr(), N, B, 768, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), None, None, None, flags)
                    assert rc == 0, (c, mode, rc)
                    torch.cuda.synchronize()
                    st = out_st.cpu().numpy()
                    assert int(st[_lib.ST_FLAGS]) == 0 and int(st[_lib.ST_N_FALLBACK]) == 0, (c, N, B, k, mode, rep, st[:8])
                    assert torch.equal(out_s, es) and torch.equal(out_i, ei), (c, N, B, k, scale, str(tdt), mode, rep)
                got[mode] = int(st[_lib.ST_N_CANDIDATES])
            # (the DMA kernel's first thresholds come from TWO scores per workgroup tile instead of one: never more candidates than the same order of
            #  magnitude, and for k = 256 several times FEWER; every query brings at least its k)
            assert B * min(k, N) <= got[1] <= 3.0 * got[0] + 64 * k, (c, N, B, k, got)
            del slab, ref, ws
    finally:
        T.atlas_tune_set_scan_dma(1)
        T.atlas_tune_set_dma_deal(1)



def test_gemm_shaped_passes_random_cases_equal_the_exact_path(gpu_index_cls):
    """random shard sizes (incl. the 65 536-row minimum and ragged last tiles), batch sizes through every column-tile width and into a second
    pass, k, score scales, duplicated rows, both twins -- against the MFMA-free exact path on the device (tools/gscan_fuzz.py runs more of them:
    profiles/r04/gscan_fuzz_48_cases.txt)"""
    rng = np.random.default_rng(77)
    g = torch.Generator(device="cuda").manual_seed(78)
    for c in range(16):
        N = int(rng.choice([65536, 65536 + int(rng.integers(1, 256)), int(rng.integers(66000, 400000)), int(rng.integers(400000, 1200000))]))
        B = int([rng.integers(97, 129), rng.integers(129, 193), rng.integers(193, 257), rng.integers(257, 385), rng.integers(385, 513),
                 rng.integers(513, 1025), rng.integers(1025, 1300)][c % 7])
        k = int(rng.choice([1, 5, 40, 40, 100, 256]))
        scale = float(rng.choice([1.0, 1.0, 0.05, 4.0]))
        slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
        for r0 in range(0, N, 200_000):
            n = min(200_000, N - r0)
            x = torch.randn((n, 768), generator=g, device="cuda")
            slab[r0 : r0 + n] = (x / x.norm(dim=1, keepdim=True) * scale).half()
        if c % 3 == 0:
            slab[N // 2 : N // 2 + 1000] = slab[:1000]
        q = torch.randn((B, 768), generator=g, device="cuda") * float(rng.choice([1.0, 0.3, 3.0]))
        idx = gpu_index_cls(certify_every=1 if c % 2 else 64)
        idx.init_embeddings([None] * 0)
        idx._set_slab(slab)
        idx.doc_map = {}
        s, i = idx._compute_scores_and_indices(q, k)
        st = dict(idx.last_search_stats)
        assert st["path"] == "scan" and st["plan"]["gemm_passes"] >= 1 and st["pmax_trusted"] is (c % 2 == 0), (c, st)
        es, ei = idx._exact_topk(q, k)
        assert torch.equal(s, es) and torch.equal(i, ei), (c, N, B, k, scale, st)
        del idx, slab


def test_batches_of_65_to_96_queries_on_a_large_shard_take_the_gemm_shaped_pass_and_agree_with_the_streaming_passes(gpu_index_cls):
    """from 4M rows on a batch of 65..96 queries is one 128-wide GEMM-shaped pass (atlas_hip.hip: GS_SMALL_BATCH_MIN_ROWS); its results must be
    those of the two halves searched on their own (64-query streaming passes) and of the MFMA-free exact path -- a size-independent property"""
    g = torch.Generator(device="cuda").manual_seed(4321)
    N = 6_500_000
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 500_000):
        x = torch.randn((500_000, 768), generator=g, device="cuda")
        slab[r0 : r0 + 500_000] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((80, 768), generator=g, device="cuda")
    idx = gpu_index_cls()
    idx.init_embeddings([None] * 0)
    idx._set_slab(slab)
    idx.doc_map = {}
    s, i = idx._compute_scores_and_indices(q, 40)
    st = dict(idx.last_search_stats)
    assert st["path"] == "scan" and st["fallback_queries"] == 0 and st["plan"]["gemm_passes"] == 1 and sum(st["plan"].values()) == 1, st
    for lo, hi in ((0, 40), (40, 80)):
        hs, hi_ = idx._compute_scores_and_indices(q[lo:hi], 40)
        assert idx.last_search_stats["plan"]["gemm_passes"] == 0
        assert torch.equal(s[lo:hi], hs) and torch.equal(i[lo:hi], hi_)
    es, ei = idx._exact_topk(q[:4], 40)
    assert torch.equal(s[:4], es) and torch.equal(i[:4], ei)
    del idx, slab
    torch.cuda.empty_cache()


def test_search_knn_over_rccl_world_size_1(gpu_index_cls, oracle_mod):
    """The distributed branch of search_knn is taken whenever a process group exists, even at W = 1 (index.py:134). With the
    `nccl` (= RCCL) backend this runs the device-side pack -> all_gather_into_tensor -> merge kernels and the fp16 query
    all-gather on GPU tensors; results must equal the single-process ones (and the oracle)."""
    import os
    import socket
    import torch.distributed as dist

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, k = 5000, 12
        P = synth.passages_f16(N, 768, 71)
        Q = synth.queries_f32(9, 768, 72)
        idx = gpu_index_cls()
        idx.init_embeddings([{"id": str(i), "title": f"t{i}", "text": f"x{i}"} for i in range(N)])
        idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
        docs, scores = idx.search_knn(torch.from_numpy(Q).cuda(), k)
        es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
        assert [[int(d["id"]) for d in row] for row in docs] == ei.tolist()
        assert scores == es.astype(np.float32).tolist()
        docs0, scores0 = idx.search_knn(torch.empty((0, 768), device="cuda"), k)      # an empty local batch is legal (evaluate.py:30-35)
        assert docs0 == [] and scores0 == []
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scale", [0.05, 4.0, 0.0005])
def test_other_exponent_buckets(scale, gpu_index_cls, oracle_mod):
    """un-normalised slabs (0.05 * randn, scores O(0.3)) and large / tiny magnitudes: fp16 scores land in other
    exponent buckets (coarser / finer ulps, subnormal scores), the certified margin scales with pmax"""
    N, B, k = 60000, 20, 40
    P = synth.normal_f32(N, 768, 81, scale).astype(np.float16)
    Q = synth.queries_f32(B, 768, 82)
    idx = _index(gpu_index_cls, P)
    s, i = _search(idx, Q, k)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    parity.assert_identical(s, i, es, ei, f"scale={scale}")
    assert idx.last_search_stats["max_err_over_eps"] < 0.25


def test_search_beside_a_kernel_that_holds_part_of_the_chip(gpu_index_cls):
    """The scan launches one workgroup per CU and its workgroups exchange thresholds inside the kernel: what if another
    stream's kernel owns CUs, so that some scan workgroups start late? Every in-kernel wait is bounded in wall-clock time (40 us a hop) and a
    missing partner only loosens a threshold: the result must stay bit-exact with no query sent to the exact path. The time is measured and
    reported; with static row ranges a workgroup that cannot start until a CU frees up costs about one more range time (a ~2x scan for
    any number of held CUs >= 1), so the bound asserted here is 3x -- not the 1.3x a range-stealing scan would meet."""
    import ctypes
    import time

    from atlas_amd import _lib

    T = _lib.lib(tuning=True)
    T.atlas_tune_spin.argtypes, T.atlas_tune_spin.restype = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p], ctypes.c_int
    N, B, k = 2_000_000, 64, 40
    g = torch.Generator(device="cuda").manual_seed(77)
    slab = torch.randn((N, 768), generator=g, device="cuda")
    slab = (slab / slab.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, 768), generator=g, device="cuda")
    idx = gpu_index_cls()
    idx._set_slab(slab)
    s0, i0 = idx._compute_scores_and_indices(q, k)
    s0, i0 = s0.clone(), i0.clone()

    def timed(n=5):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            s, i = idx._compute_scores_and_indices(q, k)
            ts.append(time.perf_counter() - t)
            assert torch.equal(s, s0) and torch.equal(i, i0), "result changed under contention"
            assert idx.last_search_stats["fallback_queries"] == 0, idx.last_search_stats
        return float(np.median(ts))

    quiet = timed()
    side = torch.cuda.Stream()
    report = []
    for held in (8, 32):
        assert T.atlas_tune_spin(held, 60_000, side.cuda_stream) == 0      # holds `held` CUs for 60 ms
        time.sleep(0.003)
        busy = timed()
        side.synchronize()
        report.append((held, busy / quiet))
    print("synchronous search at 2M rows: quiet %.3f ms; " % (quiet * 1e3) + "; ".join(f"{h} CUs held: {r:.2f}x" for h, r in report))
    assert all(r < 3.0 for _, r in report), report


def test_host_results_do_not_alias_across_searches(gpu_index_cls):
    """`_local_topk` hands out VIEWS of a reused pinned buffer (documented there); what the public calls return must not change when
    the next search overwrites that buffer -- search_knn's lists and _compute_scores_and_indices' device tensors are kept across a second search"""
    P = synth.passages_f16(30_000, 768, 301)
    idx = _index(gpu_index_cls, P)
    Q1 = torch.from_numpy(synth.queries_f32(8, 768, 302)).cuda()
    Q2 = torch.from_numpy(synth.queries_f32(8, 768, 303)).cuda()
    docs1, scores1 = idx.search_knn(Q1, 10)
    s1, i1 = idx._compute_scores_and_indices(Q1, 10)
    keep = ([[d["id"] for d in row] for row in docs1], [list(r) for r in scores1], s1.clone(), i1.clone())
    _, _, h_s, h_i = idx._local_topk(Q1, 10)
    h_before = h_s.copy()
    docs2, scores2 = idx.search_knn(Q2, 10)
    assert [[d["id"] for d in row] for row in docs1] == keep[0] and [list(r) for r in scores1] == keep[1]
    assert torch.equal(s1, keep[2]) and torch.equal(i1, keep[3])
    assert scores2 != scores1
    assert not np.array_equal(h_s, h_before)            # ... while the documented views DO follow the buffer: callers copy them at once


def test_workspace_of_a_65_to_96_query_batch_on_a_small_shard_is_small(gpu_index_cls):
    """atlas_scan_topk_workspace_bytes sizes the GEMM-shaped layout only for batches whose plan has a GEMM-shaped pass: 80 queries on a
    1M-row shard (one 96-query streaming pass) no longer reserve -- and zero-fill -- the 1024-wide layout"""
    from atlas_amd import _lib

    L = _lib.lib()
    small = int(L.atlas_scan_topk_workspace_bytes(1_000_000, 80, 768, 40))
    one = int(L.atlas_scan_topk_workspace_bytes(1_000_000, 64, 768, 40))
    big = int(L.atlas_scan_topk_workspace_bytes(4_000_000, 80, 768, 40))          # from 4M rows on: the 128-wide GEMM-shaped pass
    wide = int(L.atlas_scan_topk_workspace_bytes(1_000_000, 1024, 768, 40))
    w96 = int(L.atlas_scan_topk_workspace_bytes(1_000_000, 96, 768, 40))
    print("workspace bytes: 1M x 64:", one, " 1M x 80:", small, " 1M x 96:", w96, " 4M x 80:", big, " 1M x 1024:", wide)
    # (measured: 275 / 413 / 413 MB -- the paired-pass layout of the streaming scan is what sizes a 65..96-query workspace at 1M rows, above the
    #  GEMM-shaped layouts; the point here is that the size follows the plan: 80 and 96 queries plan alike, and nothing shrinks with more queries)
    assert one <= small == w96 <= wide
    P = synth.passages_f16(200_000, 768, 311)
    idx = _index(gpu_index_cls, P)
    Q = torch.from_numpy(synth.queries_f32(80, 768, 312)).cuda()
    s, i = idx._compute_scores_and_indices(Q, 40)
    assert idx.last_search_stats["plan"]["gemm_passes"] == 0
    es, ei = idx._exact_topk(Q[:8], 40)
    assert torch.equal(s[:8], es) and torch.equal(i[:8], ei)


def test_gemm_shaped_certifying_twin_measures_every_fragment_of_a_tile(gpu_index_cls, oracle_mod):
    """round 5: the certifying twin of the GEMM-shaped pass takes its row norms from the wave's register slots 0 and 1, which hold slab fragments
    2 wj and 2 wj + 1 through an XOR permutation of the LDS read addresses (gscan_kernel.h) -- so EVERY one of a tile's 16 fragments (8 per
    row half, two per wave) must be squared by somebody, and the filter epilogue must undo the permutation in the row it reports. A 3 x longer row
    is planted at each of the 16 fragment positions in turn (different tiles, different rows inside the fragment): the measured pmax must be
    that row's norm every time, and the results the oracle's (the long row is every query's best or worst passage: it is IN the results)."""
    N, B, k = 70_000, 128, 40
    P0 = synth.passages_f16(N, 768, 361)
    Q = synth.queries_f32(B, 768, 362)
    for width, nq in ((4, 256), (3, 150), (2, 128)):                      # FB = 4 | 3 | 2: column tiles of 256 | 192 | 128 queries
        Qw = synth.queries_f32(nq, 768, 363 + width)
        for pos in range(16):
            if width != 4 and pos % 5 != 0:                                # (every position on the 256-wide tile, a sample of them on the others)
                continue
            P = P0.copy()
            row = (37 + 11 * pos) * 256 + pos * 16 + (5 * pos + 3) % 16
            P[row] = (P[row].astype(np.float32) * 3.0).astype(np.float16)
            want = float(np.sqrt((P[row].astype(np.float64) ** 2).sum()))
            idx = gpu_index_cls(certify_every=1)
            idx.init_embeddings([{"id": str(i)} for i in range(N)], 768)
            idx.embeddings[:, :] = torch.from_numpy(P).cuda().T
            idx._pmax, idx._pmax_version = 1.002, idx._slab_version()      # a stale hint (the C-ABI caller without a certificate)
            s, i = _search(idx, Qw, k)
            st = idx.last_search_stats
            assert st["plan"]["gemm_passes"] >= 1 and st["reruns"] == 1, (width, pos, st)
            assert abs(st["pmax"] / want - 1.0) < 2e-3, (width, pos, st["pmax"], want)
            if pos % 4 == 0:
                es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Qw), P, k)
                parity.assert_identical(s, i, es, ei, f"long row at fragment position {pos}, column tile width {width}")
                assert (i == row).any()
