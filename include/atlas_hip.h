/*
 * atlas_hip.h — C-ABI of the MI355X (gfx950) retrieval hot path.
 *
 * This is the drop-in boundary for the exact-MIPS path of facebookresearch/atlas.
 * The reference has no FFI (it is pure Python on torch ops), so every entry point
 * below cites the reference *op sequence* it replaces; INTEGRATION.md shows the
 * ctypes stub a maintainer adds to src/index.py.
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary
 *   - every buffer is caller-owned DEVICE memory unless the name says `host`
 *   - nothing allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream)
 *   - re-entrant for distinct (stream, workspace) pairs. The library has no mutable global state, reads no
 *     environment variables and keeps no per-device caches: everything a call needs is in its arguments.
 *     (Kernel variants, GEMM configurations and cycle stamps can only be selected in the separate TUNING build of
 *     the same sources, libatlas_hip_tune.so = -DATLAS_TUNING=1, whose atlas_tune_* hooks are process-global and
 *     are not part of this interface; the product never loads it.)
 *   - return value: 0 = enqueued, >0 = hipError_t from a launch, <0 = ATLAS_E_*
 *
 * Canonical result (what "top-k" means here):
 *   score(q,p) = RNE_fp16( sum_k fp16(q_k) * p_k ) with the sum taken in double in a
 *                fixed 64-chain order (bit-reproducible; = the correctly rounded fp16 of
 *                the exact inner product except with probability ~1e-13 per score)
 *   order      = (score desc, passage row asc)          -- ties: lowest row first
 *   The reference computes fp16(fp32-accumulated sum) with backend-defined summation
 *   order and backend-defined tie order (src/index.py:117-118); the canonical result is
 *   the order/tie-independent representative of that family.
 */
#ifndef ATLAS_HIP_H
#define ATLAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATLAS_ABI_VERSION 9

#define ATLAS_WS_STATE_BYTES (2u << 20)   /* head of a scan workspace that must be zero before the workspace's first use */

/* negative return codes */
#define ATLAS_E_BADARG     (-1)  /* null pointer, B<=0, k<=0, d unsupported ...        */
#define ATLAS_E_WORKSPACE  (-2)  /* ws_bytes smaller than *_workspace_bytes()          */
#define ATLAS_E_UNSUPPORTED (-3) /* shape outside the fast path (use atlas_exact_topk) */

/* query element types accepted by the search entry points (the reference casts with
 * `.half()` inside _compute_scores_and_indices, src/index.py:117) */
#define ATLAS_DT_F16  0
#define ATLAS_DT_F32  1
#define ATLAS_DT_BF16 2

/* out_status layout: int32[ATLAS_STATUS_HEADER + B] */
#define ATLAS_STATUS_HEADER      8
#define ATLAS_ST_FLAGS           0   /* bit-or of ATLAS_F_*                                    */
#define ATLAS_ST_PMAX_BITS       1   /* float bits: largest passage-row L2 norm seen by scan    */
#define ATLAS_ST_N_FALLBACK      2   /* number of queries flagged ATLAS_Q_FALLBACK              */
#define ATLAS_ST_N_CANDIDATES    3   /* total candidates that reached the merge (diagnostic)    */
#define ATLAS_ST_N_RESCORED      4   /* total exact rescorings done in the merge (diagnostic)   */
#define ATLAS_ST_MAXERR_BITS     5   /* float bits: max |approx-exact| / eps over rescored rows */
#define ATLAS_ST_PLAN            6   /* the slab passes the call was made of (diagnostic): bits 0-7 single 64-query passes, 8-15 single
                                        96-query passes, 16-19 pairs of 64-query passes, 20-23 pairs of 96-query passes, 24-31 GEMM-shaped
                                        passes (<= 256 / 512 / 1024 queries each; batches above 96 queries) */
/* flags */
#define ATLAS_F_PMAX_VIOLATION   1   /* a row norm exceeded pmax_hint: results NOT certified;
                                        re-run with pmax_hint >= out_status[ATLAS_ST_PMAX_BITS] */
#define ATLAS_F_FALLBACK         2   /* >=1 query needs atlas_exact_topk (per-query word != 0)  */
#define ATLAS_F_EPS_VIOLATION    4   /* a rescored row had |approx-exact| > eps (model broken)  */
/* per-query status word (out_status[ATLAS_STATUS_HEADER + q]) */
#define ATLAS_Q_OK        0
#define ATLAS_Q_FALLBACK  1          /* candidate band overflowed (mass ties): rows not written */

/* ---- library info -------------------------------------------------------------- */
int         atlas_abi_version(void);
const char* atlas_build_info(void);          /* "gfx950 <compile date> ..." */

/* ---- search: fused MFMA scan + top-k (replaces src/index.py:113-120) -------------
 *
 * Replaces   scores = torch.matmul(allqueries.half(), self.embeddings)   index.py:117
 *            scores, indices = torch.topk(scores, topk, dim=1)           index.py:118
 * without materialising `scores`.
 *
 *   q          [B x d] row-major, element type q_dtype (converted to fp16 RNE = `.half()`)
 *   slab_f16   [N x d] row-major fp16 passage slab (the reference keeps (d,N); the Python
 *              class exposes slab.T so atlas.py:79 is unchanged)
 *   k          neighbours per query; rows beyond min(k,N) are filled with (-inf, -1)
 *   pmax_hint  upper bound on the L2 norm of any slab row (the certified error margin
 *              scales with it); if a larger row is met, ATLAS_F_PMAX_VIOLATION is raised
 *              and the measured maximum is reported so the caller can re-run once. Measuring every row's norm inside
 *              the scan is not free (one Gram MFMA per k-step: ~3 % of the scan time; 4 v_dot2 per k-step before round 4: 4.6-7.5 %): a caller that KNOWS
 *              its bound -- atlas_slab_pmax() taken after the last write to the slab -- passes ATLAS_SCAN_TRUST_PMAX to
 *              atlas_scan_topk_flags() and the scan takes pmax_hint as certified (ATLAS_ST_PMAX_BITS then reads 0)
 *   out_score  [B x k] fp16, canonical scores, descending
 *   out_idx    [B x k] int64 shard-local passage rows (same meaning as torch.topk indices)
 *   out_status int32[ATLAS_STATUS_HEADER + B]
 *   ws         workspace of >= atlas_scan_topk_workspace_bytes(N,B,d,k) bytes, 256-B aligned. ZERO-FILL IT ONCE before its
 *              first use (hipMemset); from then on it belongs to the library between calls: its first
 *              ATLAS_WS_STATE_BYTES hold per-query fallback flags (cleared again by the kernel that reads them), a call
 *              counter and the 8-byte {tag = counter + 1, value} granules of the in-kernel threshold exchange (a stale or
 *              garbage granule must not carry the current tag). One workspace serves one stream at a time.
 *
 * Fast path requires d == 768 (EMBEDDINGS_DIM, src/retrievers.py:13) and k <= 256;
 * otherwise returns ATLAS_E_UNSUPPORTED (callers then use atlas_exact_topk).
 * Any B >= 1 is accepted: up to 64 queries are one slab pass; a larger batch is a sequence of passes chosen by measured cost
 * (ATLAS_ST_PLAN reports it). Up to 64 queries the slab BYTES are the bound and the pass streams. Above that the matrix pipe takes over and the
 * passes are GEMM-shaped (csrc/gscan_kernel.h; shards of >= 65 536 rows; 65..96 queries from 4M rows on, a 96-query streaming pass below):
 * 256 slab rows x a column tile of 128, 192 or 256 queries per workgroup tile, both operands staged through LDS, up to 128 / 192 / 256 / 384 /
 * 512 / 1024 queries per pass for ONE slab read from HBM (the workgroups that score the same rows against different query tiles share them
 * through the L2); a sample launch gives every query its first threshold, a second scan launch runs with thresholds tightened by the
 * candidates of the first eighth of the slab. Costs in units of a 64-query pass: 1.10 / 1.30 / 1.50 / 2.32 / 2.69 / 5.02; 512 queries on a
 * 4M-row shard: 2.75 ms (0.46 of the f16 MFMA peak) against 5.7 ms for round 3's streaming passes (paired 64- / 96-query passes, which
 * remain for small shards). The workspace size depends on B: a workspace sized for a larger batch serves every smaller one; the state
 * words at its head do not move with B.
 *
 * One 64-query pass is two launches: the scan (which converts the queries itself, takes its initial pruning thresholds
 * from its own first tiles -- the workgroups exchange 8-byte granules inside the kernel; every wait is bounded, a value
 * that does not arrive in time only loosens a threshold -- and leaves per-workgroup candidate lists) and the merge
 * (exact rescoring of the candidate band, canonical order). Shards below 65 536 rows start without thresholds.
 * The last ~6 % of a large shard's rows are handed out to the scan's workgroups at run time (one ticket per 256-row tile
 * from a counter in the workspace state, put back by the merge): which workgroup scans which of those rows differs from
 * call to call, the result -- canonical order, exact scores -- does not.
 * Round 6: the scan of a pass of up to 64 queries on a shard of >= 65 536 rows is csrc/dscan_kernel.h -- the slab staged through LDS-DMA
 * (full 128-byte lines, nt policy), the queries in registers: 0.85-0.88 of the 8 TB/s HBM peak at 32M rows where the register-fed
 * scan_kernel.h (which remains for 96-query and paired passes and smaller shards) reached 0.77. Its workgroups take the slab's 256-row tiles in turn
 * (workgroup g: tiles g, g + G, ...: one ~100 MB window moves through the slab), so a shard of 128M rows (197 GB) streams at the same 0.88 as one of
 * 16M. Same workspace, same results bit for bit.
 */
size_t atlas_scan_topk_workspace_bytes(int64_t N, int B, int d, int k);
int atlas_scan_topk(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d,
                    int k, float pmax_hint, void* out_score_f16, int64_t* out_idx,
                    int32_t* out_status, void* ws, size_t ws_bytes, void* stream);
/* Same call with two optional hipEvent_t handles (NULL = skip) recorded on `stream` immediately
 * before and after the scan kernel of the first 64-query chunk: lets a harness time the dominant
 * kernel alone, on the stream it actually runs on (bench.py's roofline figure). */
int atlas_scan_topk_ex(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d,
                       int k, float pmax_hint, void* out_score_f16, int64_t* out_idx,
                       int32_t* out_status, void* ws, size_t ws_bytes, void* stream,
                       void* ev_scan_begin, void* ev_scan_end);
/* The same call with `flags` (0 = atlas_scan_topk_ex):
 *   ATLAS_SCAN_TRUST_PMAX   pmax_hint is a certified upper bound of the row norms: the scan does not re-measure them. What the call
 *                           still checks is every row its merge rescans (the candidates that reach the exact rescoring): one longer than
 *                           pmax_hint raises ATLAS_F_PMAX_VIOLATION with ATLAS_ST_PMAX_BITS = that row's norm -- a LOWER bound of the true
 *                           maximum; the caller's certificate was stale: take atlas_slab_pmax() again and repeat the call */
#define ATLAS_SCAN_TRUST_PMAX 1
int atlas_scan_topk_flags(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d,
                       int k, float pmax_hint, void* out_score_f16, int64_t* out_idx,
                       int32_t* out_status, void* ws, size_t ws_bytes, void* stream,
                       void* ev_scan_begin, void* ev_scan_end, int flags);

/* The same call that ALSO emits the winners as cross-shard packed candidates (see atlas_pack_candidates below): out_packed [B x k]
 * uint64, global_id = row * id_mul + id_add, padding entries 0. The distributed search_knn (src/index.py:134-151) hands these straight
 * to its one all-gather: no pack kernel between the merge and the collective. out_packed may be NULL (= atlas_scan_topk_flags).
 * Queries flagged ATLAS_Q_FALLBACK carry no valid packed row: the caller packs what atlas_exact_topk returns for them. */
int atlas_scan_topk_pack(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d,
                         int k, float pmax_hint, void* out_score_f16, int64_t* out_idx,
                         int32_t* out_status, void* ws, size_t ws_bytes, void* stream,
                         void* ev_scan_begin, void* ev_scan_end, int flags,
                         int64_t id_mul, int64_t id_add, uint64_t* out_packed);

/* ---- search: exact reference-order path (any d, any k <= 2048) --------------------
 * Same contract and same canonical result as atlas_scan_topk, computed without MFMA:
 * every score is formed in the canonical double order and selection is exact. Slower than the
 * scan (fp64 VALU) but bounded: ONE slab pass scores up to 8 queries (one wave per row, the 64 chains of
 * every query reduced by the canonical tree), then a 6-pass radix select over the 48 significant key
 * bits per query -- at 32M rows ~20 ms per 8 queries (measured 20.4 ms, profiles/r05/bench_32m_only_kernel_stats.csv). Used for queries flagged ATLAS_Q_FALLBACK, for
 * shapes outside the fast path (d <= 8192, k <= 2048), and as the on-device cross-check in tests.
 * Workspace: 8 keys of 8 bytes per row (atlas_exact_topk_workspace_bytes).
 */
size_t atlas_exact_topk_workspace_bytes(int64_t N, int B, int d, int k);
int atlas_exact_topk(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d,
                     int k, void* out_score_f16, int64_t* out_idx, void* ws, size_t ws_bytes,
                     void* stream);

/* ---- cross-shard merge (replaces the W*k torch.topk merge, src/index.py:151) -------
 * Packed candidate = uint64: (orderable_fp16(score) << 47) | (2^47-1 - global_id);
 * larger = better, so the canonical order is a plain descending integer order.
 *   atlas_pack_candidates: (score fp16, local row) -> packed, global_id = row*id_mul + id_add
 *                          (round-robin shards, src/index_io.py:41: id_mul=W, id_add=rank;
 *                           contiguous shards, src/index.py:95-99: id_mul=1, id_add=offset);
 *                          rows with idx < 0 pack to 0 (= worse than anything real)
 *   atlas_merge_packed:    in [W][B][k] (an all_gather of the per-rank [B][k]) -> out [B][k]
 *                          the k largest per query, descending
 */
int atlas_pack_candidates(const void* score_f16, const int64_t* idx, int64_t n, int64_t id_mul,
                          int64_t id_add, uint64_t* out_packed, void* stream);
int atlas_merge_packed(const uint64_t* gathered, int W, int B, int k, uint64_t* out_packed,
                       void* stream);

/* (The peer exchange -- atlas_xchg_*: a one-hop alternative to the all-gather + atlas_merge_packed above that has never run across two
 * devices -- is declared in include/atlas_hip_experimental.h, outside the product interface, until it has.) */

/* ---- index refresh epilogue (replaces src/retrievers.py:50-52 + src/atlas.py:79) ----
 * Masked mean pooling of the encoder's last hidden state, with the reference's fp16
 * double rounding (sum -> fp16, then / count -> fp16), written as contiguous rows
 * slab[row_offset + i, :] instead of the reference's stride-N column scatter.
 *   hidden_f16 [n x L x d] fp16, mask [n x L] int64 (HF attention_mask), slab [N x d] fp16
 */
int atlas_pool_write(const void* hidden_f16, const int64_t* mask, void* slab_f16, int64_t N,
                     int64_t row_offset, int n, int L, int d, void* stream);

/* ---- Contriever encoder (replaces src/retrievers.py:22-60 + src/modeling_bert.py) ---------------------------
 * Index refresh: the fp16 inference copy `copy.deepcopy(retriever).half().eval()` of Atlas.build_index
 * (src/atlas.py:54-59,78). Query embedding: the retriever in model precision (src/atlas.py:104; --precision
 * fp32 | fp16 | bf16). BERT-base encoder (12 x [QKV, attention with fp32 softmax, out-proj + residual + LayerNorm,
 * FFN with exact-erf GELU + residual + LayerNorm]; the reference's NON-standard LayerNorm, modeling_bert.py:104-114)
 * + masked mean pooling, every intermediate rounded to the model dtype where the reference materialises a tensor
 * of that dtype. Inference only (no autograd).
 * All weights are device pointers of dtype `dtype` (ATLAS_DT_*), Linear weights row-major [out][in] as in the HF
 * state dict; qkv_w = rows of query.weight | key.weight | value.weight ([2304][768]), qkv_b likewise.
 *   input_ids, attention_mask, token_type_ids (nullable): int64 [n x L] device tensors (HF tokenizer output);
 *            attention_mask is any 0/1 pattern. Only tokens with mask != 0 are computed (packed on the device, no
 *            host sync): cost follows the real token count, results do not depend on the padding.
 *   out:     [n x 768] rows of `dtype` (fp32 for ATLAS_POOL_SQRT), contiguous; for fp16 it may point into the passage slab
 *            (slab + row_offset*768), which makes the refresh write atlas.py:79 part of the pooling epilogue.
 *            A row whose mask is all zero is NaN (0/0), as in the reference.
 *   L <= 512, hidden 768, 12 heads, intermediate 3072 (ATLAS_E_UNSUPPORTED otherwise).
 */
#define ATLAS_BERT_MAX_LAYERS 24
#define ATLAS_POOL_AVERAGE 0
#define ATLAS_POOL_SQRT    1
#define ATLAS_POOL_CLS     2
typedef struct {
    const void *qkv_w, *qkv_b, *o_w, *o_b, *ln1_w, *ln1_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b, *ln2_w, *ln2_b;
} atlas_bert_layer;
typedef struct {
    int n_layers, n_heads, hidden, intermediate;
    float eps;                                   /* config.layer_norm_eps */
    int dtype;                                   /* ATLAS_DT_F16 | ATLAS_DT_BF16 | ATLAS_DT_F32: weights, activations, output */
    int pooling;                                 /* config.pooling, retrievers.py:51-56: ATLAS_POOL_AVERAGE (atlas' default) |
                                                    ATLAS_POOL_SQRT (sum / sqrt(count): the OUTPUT is fp32, as torch promotes) |
                                                    ATLAS_POOL_CLS (hidden state of position 0, zero if that token is masked) */
    int vocab_size, max_positions, type_vocab;   /* rows of word_emb / pos_emb / type_emb. L > max_positions -> ATLAS_E_BADARG;
                                                    token / type ids outside the tables are clamped into them (the reference
                                                    raises a device-side assert there; this library never reads out of bounds) */
    const void *word_emb, *pos_emb, *type_emb, *emb_ln_w, *emb_ln_b;
    atlas_bert_layer layers[ATLAS_BERT_MAX_LAYERS];
} atlas_bert_weights;
size_t atlas_contriever_workspace_bytes(int n, int L, int dtype);
int atlas_contriever_embed(const atlas_bert_weights* w /* host struct of device pointers */, const int64_t* input_ids,
                           const int64_t* attention_mask, const int64_t* token_type_ids, int n, int L, void* out,
                           void* ws, size_t ws_bytes, void* stream);
/* The same with a row map: passage b of the batch is written to out[out_rows[b]] (out_rows: int64 [n] on the device; NULL = b).
 * `out` is then the BASE of the destination (the passage slab): a refresh that batches passages by length instead of by position
 * (atlas_amd/token_store.py) still lands every embedding in its own slab row, inside the pooling epilogue (src/atlas.py:79). */
int atlas_contriever_embed_rows(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask,
                                const int64_t* token_type_ids, int n, int L, void* out, const int64_t* out_rows, void* ws,
                                size_t ws_bytes, void* stream);

/* ---- slab statistics ------------------------------------------------------------
 * out_pmax (device float): max L2 norm over rows [0,N). One streaming pass; lets a caller
 * obtain a certified pmax_hint up front instead of via the violation/re-run protocol.
 */
int atlas_slab_pmax(const void* slab_f16, int64_t N, int d, float* out_pmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATLAS_HIP_H */
