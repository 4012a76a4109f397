// atlas_hip.hip — gfx950 (MI355X, CDNA4) kernels + C-ABI for the Atlas exact-MIPS hot path.
//
// Kernels (DESIGN.md §4 has the roofline of each):
//   prep_queries_kernel   queries -> fp16 (== `.half()`, src/index.py:117), MFMA B-fragment
//                         order for LDS, row-major copy for rescoring, per-query eps
//   dscan_kernel          the hot one (round 6, dscan_kernel.h): streams the (N,768) fp16 slab once through LDS-DMA
//                         (full 128-byte lines, nt), 16x16x32 f16 MFMA against 64 queries held in REGISTERS,
//                         per-lane threshold filter, per-workgroup candidate lists with certified pruning
//                         margins. Scores never reach HBM (replaces matmul+topk, index.py:117-118)
//   scan_kernel           the same pass with the slab HBM -> VGPR and the queries in an LDS image (rounds 1-5:
//                         every pass; now 96-query and paired passes, shards below 65 536 rows)
//   gscan_kernel          the slab pass of 97..1024 queries, shaped like a GEMM (gscan_kernel.h)
//   merge_rescore_kernel  per query: k-th of all surviving candidates, exact (canonical
//                         double-order) rescoring of the candidate band, canonical sort
//   exact_* kernels       MFMA-free exact path for any (d,k): fallback + on-device cross-check
//   pack/merge_packed     cross-shard candidate packing and W*k -> k merge (index.py:151)
//   pool_write_kernel     masked mean pooling + contiguous slab row write
//                         (retrievers.py:50-52 + atlas.py:79)
//   slab_pmax_kernel      max row norm (certified eps needs an upper bound on |p|)
//
// gfx950 only. No CUDA paths, no compatibility layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifndef ATLAS_TUNING
#define ATLAS_TUNING 0
#endif

#include "common.h"
#include "scan_kernel.h"
#include "gscan_kernel.h"
#include "dscan_kernel.h"
#include "../../include/atlas_hip.h"
#include "../../include/atlas_hip_experimental.h"   // atlas_xchg_*: exported, outside the product interface

using namespace atlas;


#define K_FAST_MAX 256
#define K_EXACT_MAX 2048


// ------------------------------------------------------------------------------------------
// prep: one block per query slot (64 slots; slots >= nq are zero queries with eps 0)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
prep_queries_kernel(const void* __restrict__ q, int q_dtype, int q0, int nq, int d, float pmax,
                    uint16_t* __restrict__ qrow /*[64][d]*/, uint16_t* __restrict__ qfrag /*fragment order*/,
                    float* __restrict__ qeps /*[64]*/, uint32_t* __restrict__ scan_state /*spare[64] | qflag[64] | spare[64] or null*/,
                    int32_t* __restrict__ out_status /*or null*/) {
    const int j = blockIdx.x;
    // also resets the per-call state words (saves two memset dispatches per search)
    if (scan_state != nullptr) {
        if (j == 0 && threadIdx.x < 64) scan_state[threadIdx.x] = 0u;
        if (threadIdx.x == 0) { scan_state[64 + j] = 0u; scan_state[128 + j] = 0u; }
    }
    if (out_status != nullptr) {
        if (q0 == 0 && j == 0 && threadIdx.x < ATLAS_STATUS_HEADER) out_status[threadIdx.x] = 0;
        if (j < nq && threadIdx.x == 0) out_status[ATLAS_STATUS_HEADER + q0 + j] = 0;
    }
    __shared__ double red[256];
    double ss = 0.0;
    for (int k = threadIdx.x; k < d; k += 256) {
        uint16_t h = 0;
        if (j < nq) {
            const size_t off = (size_t)(q0 + j) * d + k;
            if (q_dtype == ATLAS_DT_F16) h = ((const uint16_t*)q)[off];
            else if (q_dtype == ATLAS_DT_F32) h = f32_to_f16_bits(((const float*)q)[off]);
            else h = bf16_bits_to_f16_bits(((const uint16_t*)q)[off]);
        }
        qrow[(size_t)j * d + k] = h;
        const double v = f16_bits_to_f64(h);
        ss += v * v;
        if (qfrag != nullptr) {
            // MFMA B operand of v_mfma_f32_16x16x32_f16: lane l holds B[k = 8*(l>>4)+e][col = l&15]
            const int s = k >> 5, g = (k >> 3) & 3, e = k & 7, qf = j >> 4, lane = (j & 15) + 16 * g;
            qfrag[((size_t)((s * 4 + qf) * 64 + lane) << 3) + e] = h;
        }
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // eps = GAMMA * |q| * pmax, rounded up
        const float nq2 = (float)sqrt(red[0]) * 1.000001f;
        qeps[j] = (j < nq) ? ATLAS_GAMMA * nq2 * pmax * 1.000001f : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// merge + exact rescoring: merge_kernel.h (merge_rescore_body). One block per query as a kernel of its own (small shards, two-kernel
// mode, tuning A/B); the coop scan runs the same body in its last workgroups (scan_kernel.h: fused merge)
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT)
merge_rescore_kernel(const MergeParams pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MergeParams p = pk;
    if (blockIdx.y != 0) {                                 // the second chunk of a paired pass
        auto st = [&](auto*& ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)((unsigned char*)ptr + pk.pair_state); };
        auto bk = [&](auto*& ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)((unsigned char*)ptr + pk.pair_bulk); };
        st(p.epoch); st(p.ticket); st(p.qflag);
        bk(p.lists); bk(p.list_cnt); bk(p.wg_stat);
        p.q0 = pk.q0 + pk.nq1;
        p.qbase = pk.qbase + pk.nq1;
    }
    if ((int)blockIdx.x >= (blockIdx.y ? pk.nq2 : pk.nq1)) return;
    merge_rescore_body<NT>(p, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// exact path (no MFMA): canonical keys for every row, then an exact radix select.
// Batched: ONE slab pass scores every row against up to EXACT_QB queries (keys[t][row]), and every selection kernel handles
// the whole batch (blockIdx.y = query of the batch). A slab row is read once per EXACT_QB queries instead of once per query.
// ------------------------------------------------------------------------------------------
#define EXACT_QB 8                // queries per slab pass
#define EXACT_PASSES 6            // radix passes: the canonical key uses bits 47..0 (16-bit score key | 32-bit inverted row)

// One wave per row. Lane j is chain j of the canonical order for ALL queries of the batch: c[t] += q_t[e] * p[e] for its elements
// e = j, j+64, ... in order (common.h exact_dot_f16). The 64 chains of each query are then combined by the canonical balanced tree
// c[j] += c[j+m], m = 1, 2, .. 32 -- as a reduce-scatter for the first three levels (each lane keeps half of its queries and
// hands the other half to its partner: 4 + 2 + 1 exchanges instead of 3 x 8), then a plain butterfly. A lane adds (own, partner)
// or (partner, own): IEEE addition is commutative, so every sum has the bits the serial tree gives (oracle/oracle.c).
// Afterwards lane j holds the score of query ((j&1)<<2)|(j&2)|((j>>2)&1).
template <bool D768>
__global__ void __launch_bounds__(256)
exact_keys_batch_kernel(const uint16_t* __restrict__ slab, int64_t N, int d, const uint16_t* __restrict__ qrow /*[nq][d]*/, int nq,
                        uint64_t* __restrict__ keys /*[EXACT_QB][N]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* qs = (uint16_t*)smem;                                // [EXACT_QB][d] fp16 bits (queries >= nq are zero)
    for (int i = threadIdx.x; i < EXACT_QB * d; i += 256) {
        const int t = i / d;
        qs[i] = (t < nq) ? qrow[i] : (uint16_t)0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double qreg[D768 ? EXACT_QB : 1][D768 ? D_FAST / 64 : 1];      // d == 768: the lane's 12 elements of every query live in registers
    if (D768) {
#pragma unroll
        for (int t = 0; t < EXACT_QB; ++t)
#pragma unroll
            for (int i = 0; i < D_FAST / 64; ++i) qreg[t][i] = (double)(float)__builtin_bit_cast(_Float16, qs[t * D_FAST + i * 64 + lane]);
    }
    const int mine = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N; r += (int64_t)gridDim.x * 4) {
        const uint16_t* prow = slab + (size_t)r * d;
        double c[EXACT_QB];
#pragma unroll
        for (int t = 0; t < EXACT_QB; ++t) c[t] = 0.0;
        if (D768) {
            uint16_t pv[D_FAST / 64];
#pragma unroll
            for (int i = 0; i < D_FAST / 64; ++i) pv[i] = prow[i * 64 + lane];          // all 12 loads in flight
#pragma unroll
            for (int i = 0; i < D_FAST / 64; ++i) {
                const double pd = (double)(float)__builtin_bit_cast(_Float16, pv[i]);
#pragma unroll
                for (int t = 0; t < EXACT_QB; ++t) c[t] += qreg[t][i] * pd;
            }
        } else {
            for (int e = lane; e < d; e += 64) {
                const double pd = (double)(float)__builtin_bit_cast(_Float16, prow[e]);
#pragma unroll
                for (int t = 0; t < EXACT_QB; ++t) c[t] += (double)(float)__builtin_bit_cast(_Float16, qs[t * d + e]) * pd;
            }
        }
        // levels m = 1, 2, 4 as a reduce-scatter over the queries
        {
            const bool up = lane & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double keep = up ? c[i + 4] : c[i], give = up ? c[i] : c[i + 4];
                c[i] = keep + __shfl_xor(give, 1);
            }
        }
        {
            const bool up = lane & 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double keep = up ? c[i + 2] : c[i], give = up ? c[i] : c[i + 2];
                c[i] = keep + __shfl_xor(give, 2);
            }
        }
        {
            const bool up = lane & 4;
            const double keep = up ? c[1] : c[0], give = up ? c[0] : c[1];
            c[0] = keep + __shfl_xor(give, 4);
        }
        double sc = c[0];
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) sc += __shfl_xor(sc, m);
        if (lane < EXACT_QB && mine < nq) keys[(size_t)mine * N + r] = local_key(f64_to_f16_bits_rto(sc), (uint32_t)r);
    }
}

// state derived from the histograms of the passes already done: (prefix, remaining rank). Pass ps looks at byte 5 - ps.
static __device__ __forceinline__ void radix_state(const uint32_t* __restrict__ hists, int npass_done, int k,
                                                   uint64_t& prefix, uint32_t& rem) {
    prefix = 0; rem = (uint32_t)k;
    for (int ps = 0; ps < npass_done; ++ps) {
        const uint32_t* h = hists + ps * 256;
        uint32_t acc = 0, b = 0;
        for (int bi = 255; bi >= 0; --bi) {
            if (acc + h[bi] >= rem) { b = (uint32_t)bi; break; }
            acc += h[bi];
        }
        rem -= acc;
        prefix |= (uint64_t)b << (40 - 8 * ps);
    }
}

// per-query state words of the selection, contiguous per query of the batch: hists[EXACT_PASSES][256] | nsel | pad
#define EXACT_STATE_WORDS (EXACT_PASSES * 256 + 64)

__global__ void __launch_bounds__(256)
exact_hist_kernel(const uint64_t* __restrict__ keys_all, int64_t N, int k, int pass, uint32_t* __restrict__ state_all) {
    const uint64_t* keys = keys_all + (size_t)blockIdx.y * N;
    uint32_t* hists = state_all + (size_t)blockIdx.y * EXACT_STATE_WORDS;
    __shared__ uint32_t lh[256];
    __shared__ uint64_t s_prefix;
    lh[threadIdx.x] = 0;
    if (threadIdx.x == 0) { uint64_t pf; uint32_t rem; radix_state(hists, pass, k, pf, rem); s_prefix = pf; }
    __syncthreads();
    const uint64_t prefix = s_prefix;
    const int shift = 40 - 8 * pass;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (int64_t)gridDim.x * 256) {
        const uint64_t key = keys[r];
        if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
            atomicAdd(&lh[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hists[pass * 256 + threadIdx.x], lh[threadIdx.x]);
}

// keys >= kth are exactly min(k,N) keys (unique): gather, rank by counting, write outputs
__global__ void __launch_bounds__(256)
exact_collect_kernel(const uint64_t* __restrict__ keys_all, int64_t N, int k, uint32_t* __restrict__ state_all,
                     uint64_t* __restrict__ sel_all /*[EXACT_QB][k]*/) {
    const uint64_t* keys = keys_all + (size_t)blockIdx.y * N;
    uint32_t* state = state_all + (size_t)blockIdx.y * EXACT_STATE_WORDS;
    uint32_t* nsel = state + EXACT_PASSES * 256;
    uint64_t* sel = sel_all + (size_t)blockIdx.y * k;
    __shared__ uint64_t s_kth;
    if (threadIdx.x == 0) {
        uint64_t pf; uint32_t rem;
        radix_state(state, EXACT_PASSES, k, pf, rem);
        s_kth = ((int64_t)k <= N) ? pf : 0ull;     // fewer than k rows: take everything
    }
    __syncthreads();
    const uint64_t kth = s_kth;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (int64_t)gridDim.x * 256) {
        const uint64_t key = keys[r];
        if (key >= kth) {
            const uint32_t s = atomicAdd(nsel, 1u);
            if (s < (uint32_t)k) sel[s] = key;
        }
    }
}

// one block per query of the batch
__global__ void __launch_bounds__(256)
exact_emit_kernel(const uint64_t* __restrict__ sel_all, const uint32_t* __restrict__ state_all, int k,
                  uint16_t* __restrict__ o_score_all, int64_t* __restrict__ o_idx_all) {
    const uint64_t* sel = sel_all + (size_t)blockIdx.x * k;
    uint16_t* o_score = o_score_all + (size_t)blockIdx.x * k;
    int64_t* o_idx = o_idx_all + (size_t)blockIdx.x * k;
    uint32_t n = state_all[(size_t)blockIdx.x * EXACT_STATE_WORDS + EXACT_PASSES * 256];
    if (n > (uint32_t)k) n = (uint32_t)k;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t ki = sel[i];
        uint32_t pos = 0;
        for (uint32_t j = 0; j < n; ++j) pos += (sel[j] > ki) ? 1u : 0u;
        o_score[pos] = f16_from_order_key((uint16_t)(ki >> 32));
        o_idx[pos] = (int64_t)(0xffffffffu - (uint32_t)ki);
    }
    for (uint32_t i = n + threadIdx.x; i < (uint32_t)k; i += 256) { o_score[i] = 0xfc00; o_idx[i] = -1; }
}

// ------------------------------------------------------------------------------------------
// cross-shard packing / merge
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_candidates_kernel(const uint16_t* __restrict__ score, const int64_t* __restrict__ idx, int64_t n,
                       int64_t id_mul, int64_t id_add, uint64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = idx[i];
    out[i] = (r < 0) ? 0ull : pack_candidate(score[i], (uint64_t)(r * id_mul + id_add));
}

// one block per query; W*k <= 8192 candidates staged in LDS, rank by counting
__global__ void __launch_bounds__(256)
merge_packed_kernel(const uint64_t* __restrict__ gathered, int W, int B, int k, uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* c = (uint64_t*)smem;
    const int q = blockIdx.x, n = W * k;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int w = i / k, j = i - w * k;
        c[i] = gathered[((size_t)w * B + q) * k + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t ki = c[i];
        // strict order with index tie-break so that padded zeros (equal keys) stay distinct
        uint32_t pos = 0;
        for (int j = 0; j < n; ++j) pos += (c[j] > ki || (c[j] == ki && j < i)) ? 1u : 0u;
        if (pos < (uint32_t)k) out[(size_t)q * k + pos] = ki;
    }
}

// ------------------------------------------------------------------------------------------
// Peer exchange (atlas_xchg_*): the one-hop alternative to the all-gather of the packed winners. Every rank owns an exchange buffer that
// its peers map (hipIpc): [2 sets][XCHG_MAXW flag lines of 128 B] | [2 sets][W slots of slot_entries u64]. A search with tag t uses set
// t & 1: rank r copies its [B][k] packed winners into slot r of EVERY peer's buffer and then stores t into flag r there (system-scope
// release); the merge of a rank waits -- bounded -- until all W flags of its own buffer carry t (acquire) and ranks the W * k candidates
// of a query. Two sets are enough: a rank passes the merge of tag t only when every peer has pushed t, i.e. has finished the merge of
// t - 1 (stream order), so set (t - 1) & 1 = (t + 1) & 1 is free when this rank pushes t + 1.
// NOT the default (index.py: exchange="rccl"): never run across two devices -- no multi-GPU box was attached; its logic is tested with two
// processes on one GPU (tests/test_gpu_peer_exchange.py).
// ------------------------------------------------------------------------------------------
#define XCHG_MAXW 64
#define XCHG_HDR (2 * XCHG_MAXW * 128)
struct XchgPeers { unsigned char* buf[XCHG_MAXW]; };

__global__ void __launch_bounds__(256)
xchg_push_kernel(const uint64_t* __restrict__ packed, int64_t n, XchgPeers peers, int rank, int W, int64_t slot_entries, uint32_t tag) {
    unsigned char* dst = peers.buf[blockIdx.x];
    const int set = (int)(tag & 1u);
    uint64_t* slot = (uint64_t*)(dst + XCHG_HDR + ((size_t)set * W + rank) * (size_t)slot_entries * 8);
    for (int64_t i = threadIdx.x; i < n; i += 256) slot[i] = packed[i];
    __threadfence_system();                                // this thread's stores are out before ...
    __syncthreads();
    if (threadIdx.x == 0)                                  // ... the flag says so
        __hip_atomic_store((unsigned long long*)(dst + ((size_t)set * XCHG_MAXW + rank) * 128), (unsigned long long)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one block per query: wait for the W flags of this call, then rank the W * k candidates (as merge_packed_kernel)
__global__ void __launch_bounds__(256)
xchg_merge_kernel(const unsigned char* __restrict__ own, int W, int B, int k, int64_t slot_entries, uint32_t tag, unsigned long long wait_ticks,
                  uint64_t* __restrict__ out, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* c = (uint64_t*)smem;
    __shared__ int s_late;
    const int q = blockIdx.x, n = W * k, set = (int)(tag & 1u);
    if (threadIdx.x == 0) s_late = 0;
    __syncthreads();
    if ((int)threadIdx.x < W) {
        const unsigned long long* f = (const unsigned long long*)(own + ((size_t)set * XCHG_MAXW + threadIdx.x) * 128);
        const unsigned long long end = wall_clock64() + wait_ticks;
        while ((uint32_t)__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
            if (wall_clock64() >= end) { s_late = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (s_late) {                                          // a peer did not deliver in time: the caller repeats the exchange another way
        if (threadIdx.x == 0) atomicOr((uint32_t*)status, 1u);
        return;
    }
    __threadfence_system();                                // acquire for every thread's loads of the slots
    const uint64_t* slots = (const uint64_t*)(own + XCHG_HDR + (size_t)set * W * (size_t)slot_entries * 8);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int w = i / k, j = i - w * k;
        c[i] = __hip_atomic_load(slots + (size_t)w * slot_entries + (size_t)q * k + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t ki = c[i];
        uint32_t pos = 0;
        for (int j = 0; j < n; ++j) pos += (c[j] > ki || (c[j] == ki && j < i)) ? 1u : 0u;
        if (pos < (uint32_t)k) out[(size_t)q * k + pos] = ki;
    }
}

// ------------------------------------------------------------------------------------------
// refresh epilogue: masked mean pooling + contiguous slab row write
// one block per passage; thread t owns columns t, t+256, ...
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pool_write_kernel(const uint16_t* __restrict__ hidden, const int64_t* __restrict__ mask, uint16_t* __restrict__ slab,
                  int64_t row_offset, int L, int d) {
    const int i = blockIdx.x;
    const int64_t* m = mask + (size_t)i * L;
    __shared__ int s_cnt;
    if (threadIdx.x == 0) {
        long c = 0;
        for (int l = 0; l < L; ++l) c += m[l];     // attention_mask.sum(dim=1), retrievers.py:52
        s_cnt = (int)c;
    }
    __syncthreads();
    const float cnt = (float)s_cnt;
    const uint16_t* h = hidden + (size_t)i * L * d;
    for (int c = threadIdx.x; c < d; c += 256) {
        double s = 0.0;                                   // exact: fp16 addends, L <= 512
        for (int l = 0; l < L; ++l)
            if (m[l] != 0) s += f16_bits_to_f64(h[(size_t)l * d + c]);   // masked_fill(~mask, 0), retrievers.py:50
        const uint16_t sum16 = f64_to_f16_bits(s);                       // .sum(dim=1) result is fp16
        const float quo = f16_bits_to_f32(sum16) / cnt;                  // fp16 / int64 -> fp32 opmath
        slab[(size_t)(row_offset + i) * d + c] = f32_to_f16_bits(quo);   // -> fp16, contiguous row
    }
}

__global__ void __launch_bounds__(256)
slab_pmax_kernel(const uint16_t* __restrict__ slab, int64_t N, int d, uint32_t* __restrict__ out_bits) {
    // one wave per row, 4 rows per block iteration
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float best = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N; r += (int64_t)gridDim.x * 4) {
        const uint16_t* p = slab + (size_t)r * d;
        float s = 0.f;
        for (int c = lane; c < d; c += 64) { const float v = f16_bits_to_f32(p[c]); s += v * v; }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        best = fmaxf(best, s);
    }
    if (lane == 0 && best > 0.f) atomicMax(out_bits, f32_bits(sqrtf(best * 1.002f)));   // >= what the scan will measure
}

// d == 768: a wave takes TWO rows per step with three 16-byte loads per lane (3072 contiguous bytes: row A | row B; the middle load
// is split between them at lane 32), up to four steps in flight. Streams at the scan's rate instead of the 2-byte loads' 1.7 TB/s:
// HipDistributedIndex runs this pass once per state of the slab before it lets the scan trust the bound.
__global__ void __launch_bounds__(256)
slab_pmax768_kernel(const uint16_t* __restrict__ slab, int64_t N, uint32_t* __restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t npairs = (N + 1) / 2;
    auto sumsq = [](const uint4 v) {
        float s = 0.f;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { const f16x2 h = __builtin_bit_cast(f16x2, w[i]); s = __builtin_amdgcn_fdot2(h, h, s, false); }
        return s;
    };
    float best = 0.f;
    constexpr int U = 4;
    for (int64_t p0 = wave * U; p0 < npairs; p0 += nwaves * U) {
        uint4 v[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pr = p0 + u < npairs ? p0 + u : npairs - 1;             // clamped: loads stay unconditional
            const uint4* base = (const uint4*)(slab + (size_t)pr * 2 * D_FAST);
            const bool has_b = 2 * pr + 1 < N;                                     // an odd N: the last pair has no second row
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = j * 64 + lane;                                       // 16-byte chunk of the pair; chunks >= 96 are row B
                v[u][j] = (c < 96 || has_b) ? base[c] : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float s0 = sumsq(v[u][0]), s1 = sumsq(v[u][1]), s2 = sumsq(v[u][2]);
            float a = s0 + (lane < 32 ? s1 : 0.f), b = (lane < 32 ? 0.f : s1) + s2;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            best = fmaxf(best, fmaxf(a, b));
        }
    }
    if (lane == 0 && best > 0.f) atomicMax(out_bits, f32_bits(sqrtf(best * 1.002f)));   // >= what the scan will measure
}

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// scan_kernel instantiations. The product library runs kVariants[0] and has no knobs, no environment reads and no mutable
// globals (include/atlas_hip.h). The TUNING build (-DATLAS_TUNING=1 -> libatlas_hip_tune.so, used by tools/ and by the
// configuration-equality tests only) adds the other instantiations and process-global hooks (atlas_tune_*).
struct ScanVariant { int nw, pf, ring; void (*kern)(const ScanParams); const char* name; };
const ScanVariant kVariants[] = {
    {16, 1, 8, scan_kernel<16, 1, 8>, "scan_kernel<16,1,8>"},
#if ATLAS_TUNING
    {8, 2, 8, scan_kernel<8, 2, 8>, "scan_kernel<8,2,8>"},
    {8, 4, 4, scan_kernel<8, 4, 4>, "scan_kernel<8,4,4>"},
    {16, 2, 4, scan_kernel<16, 2, 4>, "scan_kernel<16,2,4>"},
    {12, 2, 4, scan_kernel<12, 2, 4>, "scan_kernel<12,2,4>"},
    {16, 1, 8, scan_kernel<16, 1, 8, 2>, "scan_kernel<16,1,8,nt>"},
    {16, 1, 8, scan_kernel<16, 1, 8, 16>, "scan_kernel<16,1,8,sc1>"},
#endif
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
// the production variant without the row-norm measurement, for callers that pass a certified pmax (ATLAS_SCAN_TRUST_PMAX)
const ScanVariant kTrusted = {16, 1, 8, scan_kernel<16, 1, 8, 64>, "scan_kernel<16,1,8> (trusted pmax)"};
// 96 queries per slab pass (6 MFMA column groups): the passes of batches above 64 queries (scan_kernel.h: NQF; chunk_plan below)
const ScanVariant kWide = {16, 1, 8, scan_kernel<16, 1, 8, 0, 6>, "scan_kernel<16,1,8,96q>"};
const ScanVariant kWideTrusted = {16, 1, 8, scan_kernel<16, 1, 8, 64, 6>, "scan_kernel<16,1,8,96q> (trusted pmax)"};
constexpr int QWIDE = 96;
constexpr size_t PAIR_STATE = 1u << 20;
static_assert(ATLAS_WS_STATE_BYTES >= 2 * PAIR_STATE, "both chunks' state inside the zero-filled head");

constexpr int MERGE_NT = 1024;
constexpr int SAMPLE_MAX = 16384;
#if ATLAS_TUNING
int g_scan_variant = 0;                      // atlas_tune_set_scan_variant
unsigned long long* g_merge_dbg = nullptr;   // atlas_tune_set_merge_stamps
unsigned long long* g_scan_dbg = nullptr;    // atlas_tune_set_scan_stamps
int scan_variant_index() { return (g_scan_variant >= 0 && g_scan_variant < kNumVariants) ? g_scan_variant : 0; }
int g_scan_coop = 1;                         // atlas_tune_set_scan_coop: 0 = sample kernel + early threshold exchange (A/B)
bool scan_coop_enabled() { return g_scan_coop != 0; }
int g_scan_wide = 1;                         // atlas_tune_set_scan_wide: 0 = batches above 64 queries in 64-query passes only (A/B)
bool scan_wide_enabled() { return g_scan_wide != 0; }
int g_scan_pair = 1;                         // atlas_tune_set_scan_pair: 0 = no paired passes (A/B)
bool scan_pair_enabled() { return g_scan_pair != 0; }
int g_scan_fused = 0;                        // atlas_tune_set_scan_fused: 1 = the merge inside the scan (experiment, not adopted)
bool scan_fused_enabled() { return g_scan_fused != 0; }
int g_scan_gemm = 1;                         // atlas_tune_set_scan_gemm: 0 = no GEMM-shaped passes for big batches (A/B)
bool scan_gemm_enabled() { return g_scan_gemm != 0; }
int g_gscan_nt = 1;                          // atlas_tune_set_gscan_nt: 0 = the GEMM-shaped passes' slab DMA with the default cache policy (A/B)
bool gscan_nt_enabled() { return g_gscan_nt != 0; }
int g_dma_pool_tile = 256;                   // atlas_tune_set_dma_pool_tile: rows of a pool tile of the DMA-staged scan (64 | 128 | 256)
int dma_pool_tile_rows() { return g_dma_pool_tile; }
int g_dma_deal = 1;                          // atlas_tune_set_dma_deal: 0 = the DMA-staged scan with one contiguous range per workgroup (A/B)
int dma_deal() { return g_dma_deal; }
int g_scan_dma = 1;                          // atlas_tune_set_scan_dma: 0 = 64-query passes on scan_kernel.h (round 5's kernel; A/B), 2 = the DMA kernel with the default cache policy
int scan_dma_mode() { return g_scan_dma; }
#else
constexpr unsigned long long* g_merge_dbg = nullptr;
constexpr unsigned long long* g_scan_dbg = nullptr;
constexpr int scan_variant_index() { return 0; }
constexpr bool scan_coop_enabled() { return true; }
constexpr bool scan_wide_enabled() { return true; }
constexpr bool scan_pair_enabled() { return true; }
constexpr bool scan_gemm_enabled() { return true; }
constexpr int scan_dma_mode() { return 1; }
constexpr int dma_deal() { return 1; }
constexpr int dma_pool_tile_rows() { return 256; }
constexpr bool gscan_nt_enabled() { return true; }
#endif
// The 64-query pass with the slab staged through LDS-DMA (dscan_kernel.h, round 6): what every coop, unpaired pass of up to 64 queries runs.
// [0] measures every row norm (the C-ABI's default contract), [1] takes the caller's pmax as certified (ATLAS_SCAN_TRUST_PMAX)
struct DmaVariant { void (*kern)(const ScanParams); const char* name; };
const DmaVariant kDma[2] = {{dscan_kernel<2>, "dscan_kernel<nt>"}, {dscan_kernel<2 | 64>, "dscan_kernel<nt> (trusted pmax)"}};
#if ATLAS_TUNING
const DmaVariant kDmaDefaultPolicy[2] = {{dscan_kernel<0>, "dscan_kernel<default policy>"}, {dscan_kernel<64>, "dscan_kernel<default policy> (trusted pmax)"}};
#endif
// run-time tile pool at the end of the slab: share of a workgroup's tiles that is NOT pre-assigned, and its cap
#if ATLAS_TUNING
int g_pool_permille = 60, g_pool_max = 32;   // atlas_tune_set_scan_pool
int pool_permille() { return g_pool_permille; }
int pool_max_per_wg() { return g_pool_max; }
#else
constexpr int pool_permille() { return 60; }
constexpr int pool_max_per_wg() { return 32; }
#endif

struct ScanPlan {
    int G;               // workgroups
    int64_t rows_per_wg; // static range of every workgroup
    int64_t pool_begin;  // rows [pool_begin, N) are handed out at run time, a tile at a time (scan_kernel.h: fill_next_tile)
    int pool_rows, pool_tiles;
    int keep_max, cap, tile, buf_cap, flush_at;
    int nqp;             // most queries of one slab pass this plan is laid out for (64, or 96 for batches above 64)
    int buf_cap_wide, flush_at_wide; size_t scan_lds_wide;   // the 96-query pass: its image leaves 11.5 KiB for candidates
    int S; int64_t sample_stride;     // sample pre-pass: S rows (0 = none), tile j starts at j*sample_stride
    int key_cap;
    int total_cap;
    size_t bulk_begin, bulk_size;
    size_t off_qflag, off_epoch, off_fuse, off_gran, off_gmax, off_q16, off_sample, off_list_cnt, off_wg_stat, off_lists, total;
    size_t scan_lds, merge_lds;
};

int device_cus() {       // the current device's CU count, asked every time (no cached state; an attribute read, not a property dump)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
}

ScanPlan make_plan(int64_t N, int d, int k, int cus, const ScanVariant& v, int nqp = QCHUNK) {
    ScanPlan pl{};
    pl.nqp = nqp;
    // one workgroup per CU (the 96 KB query image allows exactly one resident workgroup);
    // every workgroup gets a contiguous, 16-row aligned range of equal size
    const int64_t frags = (N + 15) / 16;
    int64_t G = cus;
    if (G > (frags + v.nw - 1) / v.nw) G = (frags + v.nw - 1) / v.nw;
    if (G < 1) G = 1;
    if (G > 1024) G = 1024;
    pl.G = (int)G;
    pl.rows_per_wg = ((frags + G - 1) / G) * 16;
    pl.tile = v.nw * v.pf * 16;
    // The tail of the slab is not pre-assigned. Workgroups stream at rates that differ by a few per cent from call to call (which CUs
    // are slow changes: a table measured on earlier calls did not help, profiles/r02/scan_tail.txt), so with equal static ranges the
    // last workgroup ends 30-50 us after the median one at 4M rows. Each workgroup therefore gets pool_per_wg tiles LESS than its
    // share, and the tiles of the pool [pool_begin, N) go to whoever is ready for the next one (one ticket per tile, drawn one tile
    // ahead). Shards with fewer than 8 tiles per workgroup keep the static split.
    pl.pool_begin = N; pl.pool_rows = 0; pl.pool_tiles = 0;
    {
        const int64_t tiles_per_wg = N / ((int64_t)G * pl.tile);
        int64_t pool_per_wg = tiles_per_wg * pool_permille() / 1000;
        if (pool_per_wg < 1) pool_per_wg = 1;
        if (pool_per_wg > pool_max_per_wg()) pool_per_wg = pool_max_per_wg();
        // one buffer descriptor spans the pool (< 4 GiB), and virtual candidate rows have 26 bits
        const int64_t by_bytes = ((int64_t)0xe0000000ll / (D_FAST * 2)) / ((int64_t)G * pl.tile) - 1;
        if (pool_per_wg > by_bytes) pool_per_wg = by_bytes;
        if (pool_per_wg > tiles_per_wg - 1) pool_per_wg = tiles_per_wg - 1;     // every workgroup keeps a static tile: its first tile is the sample
        if (tiles_per_wg >= 8 && pool_permille() > 0) {
            pl.rows_per_wg = (tiles_per_wg - pool_per_wg) * pl.tile;
            pl.pool_begin = (int64_t)G * pl.rows_per_wg;
            pl.pool_rows = (int)(N - pl.pool_begin);          // < (pool_per_wg + 1) * G * tile + G * tile rows: far below 2^31
            const int pool_tile = (v.nw - 1) * v.pf * 16;     // pool tiles leave the ticket wave without rows (scan_kernel.h)
            pl.pool_tiles = (pl.pool_rows + pool_tile - 1) / pool_tile;
        }
    }
    pl.keep_max = (2 * k > k + 64) ? 2 * k : k + 64;
    // LDS candidate buffer: what fits next to the 96 KiB query image. A flush into the global lists (scattered stores, the ring of slab
    // loads drained and restarted: 15-20 us per workgroup, profiles/r02/scan_tail.txt) is requested at 3/4: a 4M-row shard collects ~2.3k
    // candidates per workgroup and an 8M-row shard ~4.6k, both now end without one
    pl.buf_cap = ATLAS_TUNING ? 7552 : 7680;             // 60 KiB of LDS: query image + buffer + 1.3 KiB of state = 159.3 of the 160 KiB
    pl.flush_at = pl.buf_cap * 3 / 4;
    // entries per (query, workgroup) list: what survives a compaction (keep_max) plus what one flush of the buffer normally adds to ONE
    // query (buf_cap / 64 = 120 on average). A list that overflows hands its query to the exact path (correct, slower): adversarial data only
    pl.cap = 2048;
    // sample pre-pass: ~N/64 rows in tiles of 64, spread evenly (distinct rows: stride >= 256)
    pl.S = 0; pl.sample_stride = 0;
    if (N >= 65536) {
        int64_t S = (N / 64 + 63) / 64 * 64;
        if (S < 4096) S = 4096;
        if (S > SAMPLE_MAX) S = SAMPLE_MAX;
        pl.S = (int)S;
        pl.sample_stride = (N / (S / 64)) & ~(int64_t)15;
    }
    size_t o = 0;
    // state that lives across calls (zero before the first use of a workspace, ATLAS_WS_STATE_BYTES): the per-query fallback flags
    // (cleared again by the merge block that reads them) and the call counter that tags the granules of coop scans
    // (laid out for 96 queries per pass whatever the batch: a workspace keeps its state words where they are when the batch size changes)
    pl.off_qflag = o;  o += 512;
    pl.off_epoch = o;  o += 256;
    pl.off_fuse = o;   o += 512;                         // fused merge: arrivals, tags, per-query states (scan_kernel.h)
    pl.off_gran = o;   o += 1024;
    pl.off_gmax = o;   o += (size_t)QWIDE * 1024 * 8;
    // everything above is state (< 1 MiB); the second chunk of a PAIRED pass (ScanParams::nq2) keeps its own copy PAIR_STATE bytes on,
    // inside the zero-filled head too (ATLAS_WS_STATE_BYTES = 2 MiB). The bulk arrays follow from 2 MiB on, the second chunk's behind
    // the first's (bulk_size on)
    pl.bulk_begin = o = PAIR_STATE * 2;
    pl.off_q16 = o;    o += (size_t)QCHUNK * D_FAST * 2;
    pl.off_sample = o; o += (size_t)QCHUNK * SAMPLE_MAX * 4;
    pl.off_list_cnt = o; o += align_up((size_t)pl.G * nqp * 4, 256);     // every scan workgroup overwrites its words: no reset
    pl.off_wg_stat = o;  o += align_up((size_t)pl.G * 2 * 4, 256);
    pl.total_cap = 131072;                // most candidates of one query the merge takes on (beyond: exact path)
    pl.off_lists = o;  o += (size_t)pl.G * nqp * pl.cap * 8;
    pl.bulk_size = align_up(o - pl.bulk_begin, 256);
    pl.total = pl.bulk_begin + pl.bulk_size;
    pl.scan_lds = (size_t)ScanSmem::buf_off + (size_t)pl.buf_cap * 8 + (ATLAS_TUNING ? 1024 : 0);     // tuning build: + per-tile stamps (its buffer is 128 entries shorter)
    pl.buf_cap_wide = (int)((160 * 1024 - (size_t)ScanSmemT<QWIDE>::buf_off) / 8) - (ATLAS_TUNING ? 128 : 0);
    pl.flush_at_wide = pl.buf_cap_wide * 3 / 4;
    pl.scan_lds_wide = (size_t)ScanSmemT<QWIDE>::buf_off + (size_t)pl.buf_cap_wide * 8 + (ATLAS_TUNING ? 1024 : 0);
    const size_t merge_fixed = align_up((size_t)d * 2, 16) + 64 * 4 + (size_t)MERGE_SMAX * (4 + 4 + 8) + (size_t)(MERGE_GMAX + 8) * 4;
    pl.key_cap = (int)((160 * 1024 - 1024 - merge_fixed) / 4);
    pl.merge_lds = merge_fixed + (size_t)pl.key_cap * 4;
    return pl;
}

// the ranges the scan kernel's 32-bit offsets and 26-bit candidate rows can express
bool scan_plan_supported(const ScanPlan& pl, const int row_bits = 26) {
    // per-lane byte offsets inside one workgroup's range are 32-bit (buffer voffset)
    if ((pl.rows_per_wg + 2 * pl.tile) * (int64_t)(D_FAST * 2) >= (int64_t)0xfff00000ll) return false;
    if (pl.rows_per_wg + 2 * pl.tile + pl.pool_rows >= (1 << row_bits)) return false;  // buffer entries carry 26-bit (96-query pass: 25-bit) virtual rows
    if ((int64_t)pl.pool_rows * (D_FAST * 2) >= (int64_t)0xfff00000ll) return false;     // one descriptor spans the pool
    return true;
}

struct ExactPlan { size_t off_qrow, off_qeps, off_state, off_sel, off_keys, total; };
ExactPlan make_exact_plan(int64_t N, int d, int k, int B) {
    ExactPlan e{}; size_t o = 0;
    const int qb = B < EXACT_QB ? (B > 0 ? B : 1) : EXACT_QB;          // queries of one slab pass: a single fallback query needs one key row, not eight
    e.off_qrow = o;  o += align_up((size_t)QCHUNK * d * 2, 256);
    e.off_qeps = o;  o += 256;
    e.off_state = o; o += align_up((size_t)EXACT_QB * EXACT_STATE_WORDS * 4, 256);
    e.off_sel = o;   o += align_up((size_t)EXACT_QB * k * 8, 256);
    e.off_keys = o;  o += align_up((size_t)qb * (size_t)(N > 0 ? N : 1) * 8, 256);
    e.total = o;
    return e;
}

// The GEMM-shaped pass of big batches (gscan_kernel.h). One pass takes up to GS_MAXQ queries as ncol = 1, 2 or 4 column tiles of 256.
constexpr int GS_MAXQ = 1024;
// Pass widths of the GEMM-shaped scan: column tiles of 128 / 192 / 256 queries (gscan_kernel<., 2 | 3 | 4>) x 1, 2 or 4 of them. A pass takes
// the narrowest width that holds its queries; GS_COSTS = measured cost in units of one 64-query pass (1.035 ms at 4M rows, 8.27 ms at 32M:
// profiles/r04/batch_gemm_pass_ab_4m.txt, _32m.txt). The 128-wide tile's k-tile streams the slab at ~5.3 TB/s: the HBM roof, not the MFMA one.
// (Four column tiles of 192 = 768 queries: measured 5.06 ms at 4M rows, a 512- plus a 256-query pass 4.85 ms -- not a width.)
constexpr int GS_NW = 6;
constexpr int GS_WIDTH[GS_NW] = {128, 192, 256, 384, 512, 1024};
constexpr int GS_CW[GS_NW] = {128, 192, 256, 192, 256, 256};
// Batches of 65..96 queries: the 96-query streaming pass and the 128-wide GEMM-shaped pass break even around 3-4M rows (4M: 1.173 / 1.156 ms,
// profiles/r04/batch_gemm_pass_ab_4m.txt); the GEMM-shaped one is 7-9 % faster from 8M rows on (32M: 8.39 / 9.14 ms) and carries ~0.1 ms of launches
// (sample, two threshold kernels, two scan launches) that a small shard does not amortise (1M rows: 0.36 / 0.32 ms, 2M: 0.62 / 0.60:
// profiles/r04/batch_65_96_crossover.txt): from GS_SMALL_BATCH_MIN_ROWS rows on.
#ifndef GS_SMALL_BATCH_MIN_ROWS
#define GS_SMALL_BATCH_MIN_ROWS 4000000
#endif
#ifndef GS_COSTS
#define GS_COSTS {1.10f, 1.30f, 1.50f, 2.32f, 2.69f, 5.02f}
#endif
constexpr size_t GS_OFF_QFLAG = 800u << 10;     // its per-query fallback flags: state words (zero between calls) behind the coop scan's granules
static_assert(GS_OFF_QFLAG >= 512 + 256 + 512 + 1024 + (size_t)QWIDE * 1024 * 8 && GS_OFF_QFLAG + GS_MAXQ * 4 <= PAIR_STATE, "inside the first chunk's state");
struct GPlan {
    bool ok;
    int G, ncol, ldq, cw;                           // cw: queries per column tile (128, 192 or 256)
    int64_t rows_per_range;
    int s_tiles, nmax; int64_t s_stride;
    int gcap;
    size_t off_q16, off_theta, off_gcnt, off_wgstat, off_smax, off_lists, total;
};
GPlan make_gplan(int64_t N, int nq, int cus) {
    GPlan g{};
    int ncol = 1;
    int wi = 0;
    while (wi + 1 < GS_NW && GS_WIDTH[wi] < nq) ++wi;
    g.cw = GS_CW[wi]; ncol = GS_WIDTH[wi] / g.cw;
    g.ncol = ncol; g.ldq = ncol * g.cw;
    g.G = cus / (8 * ncol) * (8 * ncol);
    const int64_t full_tiles = N / GS_TILE, tiles = (N + GS_TILE - 1) / GS_TILE;
    g.ok = nq >= 1 && nq <= GS_MAXQ && g.G >= 8 * ncol && g.G <= 1024 && full_tiles >= 256;
    if (!g.ok) return g;
    const int nranges = g.G / ncol;
    g.rows_per_range = (tiles + nranges - 1) / nranges * GS_TILE;
    if (g.rows_per_range >= (1 << 24)) { g.ok = false; return g; }      // candidate entries carry a 24-bit row relative to the range
    // the sample: ~1/64 of the slab in whole tiles, evenly spread (at least 128 tiles = 2 048 fragment maxima per query, at most 2 048 =
    // 32 768 maxima: what gtheta_kernel keeps in LDS)
    int64_t st = full_tiles / 64;
    st = st < 128 ? 128 : (st > 2048 ? 2048 : st);
    g.s_tiles = (int)st; g.nmax = g.s_tiles * GS_FRAG_PER_TILE;
    g.s_stride = full_tiles / st * GS_TILE;
    g.gcap = 32768;                                                      // = 1024 virtual segments of 32 entries for the merge
    size_t o = PAIR_STATE * 2;
    g.off_q16 = o;    o += align_up((size_t)g.ldq * D_FAST * 2, 256);
    g.off_theta = o;  o += align_up((size_t)g.ldq * 4, 256);
    g.off_gcnt = o;   o += align_up((size_t)g.ldq * 4, 256);
    g.off_wgstat = o; o += align_up((size_t)g.G * 8 * 2, 256);        // two scan launches per pass
    g.off_smax = o;   o += align_up((size_t)g.nmax * g.ldq * 4, 256);
    g.off_lists = o;  o += (size_t)g.ldq * g.gcap * 8;
    g.total = o;
    return g;
}

template <typename KernelT>
void allow_lds(KernelT kern) {   // opt in to the full 160 KiB of LDS (idempotent)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

#if ATLAS_TUNING
// a kernel that does nothing but HOLD `wgs` CUs for `usec` microseconds (1024 threads + all of the LDS per workgroup: nothing else fits
// beside it), for the contention test: what happens to a search when another stream owns part of the chip
__global__ void __launch_bounds__(1024) atlas_spin_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long until = wall_clock64() + ticks;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
    if (ticks == 0ull && smem[threadIdx.x] == 77) *sink = 1;
}
#endif

// The passes a batch of B queries is made of (pure host arithmetic: CPU tests call it through the tuning build's atlas_test_plan_word). `ws_bytes` = the caller's
// workspace (what does not fit is not planned).
struct Pass { int nq, nq2; bool wide; bool gemm; };
struct BatchPlan { int rc; ScanPlan single, pair; bool wide_ok; std::vector<Pass> passes; int plan_word; };
static BatchPlan plan_batch(const int64_t N, const int B, const int d, const int k, const int cus, const size_t ws_bytes, const ScanVariant& var) {
    // Wide (96-query) passes need the coop exchange (one granule slot per query and workgroup) and the production shape.
    BatchPlan bp{};
    ScanPlan pl = make_plan(N, d, k, cus, var);
    const bool wide_ok = B > QCHUNK && scan_variant_index() == 0 && pl.S > 0 && pl.G >= QWIDE && pl.G <= 256 && scan_coop_enabled() &&
                         scan_wide_enabled() && scan_plan_supported(pl, 25);
    if (wide_ok) pl = make_plan(N, d, k, cus, var, QWIDE);
    if (!scan_plan_supported(pl)) { bp.rc = ATLAS_E_UNSUPPORTED; return bp; }
    if (ws_bytes < pl.total) { bp.rc = ATLAS_E_WORKSPACE; return bp; }
    // Passes of the batch, in order. Items: one 64-query pass (cost 1), one 96-query pass (1.11), a PAIR of 64-query passes run concurrently
    // on half the chip each (1.62 for up to 128 queries: the second reader of a slab row hits the Infinity Cache), a pair of 96-query
    // passes (1.89 for up to 192); f(n) = min over the items of cost + f(n - size): 128 -> a pair of 64, 192 -> a pair of 96,
    // 512 -> 2 pairs of 96 + a pair of 64.
    // ... and GEMM-shaped passes (gscan_kernel.h) of up to 128 / 192 / 256 / 384 / 512 / 1024 queries (GS_WIDTH, GS_COSTS above) for batches above 96
    // queries -- above 64 on shards of GS_SMALL_BATCH_MIN_ROWS rows or more --: MFMA-bound instead of LDS-fed, one slab read from HBM per pass
    // whatever its width; both twins (pmax trusted / every row norm measured).
    std::vector<Pass>& passes = bp.passes;
    ScanPlan pp = pl;                           // the half-chip plan of paired passes
    const int half = cus / 2;
    bool pair_ok = B > QCHUNK && scan_variant_index() == 0 && scan_pair_enabled() && scan_coop_enabled() && half % 8 == 0 && half >= QCHUNK && half <= 256;
    if (pair_ok) {
        pp = make_plan(N, d, k, half, var, wide_ok ? QWIDE : QCHUNK);
        pair_ok = pp.S > 0 && pp.G == half && scan_plan_supported(pp, wide_ok ? 25 : 26) && ws_bytes >= pp.bulk_begin + 2 * pp.bulk_size;
    }
    const bool pair_wide_ok = pair_ok && wide_ok && half >= QWIDE;
    bool gemm_ok[GS_NW] = {}, any_gemm = false;
    if ((B > QWIDE || (B > QCHUNK && N >= GS_SMALL_BATCH_MIN_ROWS)) && scan_variant_index() == 0 && scan_gemm_enabled()) {
        for (int i = 0; i < GS_NW; ++i) {
            const GPlan g = make_gplan(N, GS_WIDTH[i], cus);
            gemm_ok[i] = g.ok && ws_bytes >= g.total;
            any_gemm = any_gemm || gemm_ok[i];
        }
    }
    if (B <= QCHUNK || (!wide_ok && !pair_ok && !any_gemm)) {
        for (int r = B; r > 0; r -= QCHUNK) passes.push_back({r < QCHUNK ? r : QCHUNK, 0, false, false});
    } else {
        // costs in units of one 64-query pass (measured at 4M and 32M rows: profiles/r03/batch_paired_pass_ab.txt, profiles/r04/batch_gemm_pass_ab.txt)
        constexpr int NI = 4 + GS_NW;
        const float gcost[GS_NW] = GS_COSTS;
        float cost[NI] = {1.0f, 1.11f, 1.62f, 1.89f};
        int size[NI] = {QCHUNK, QWIDE, 2 * QCHUNK, 2 * QWIDE};
        bool ok[NI] = {true, wide_ok, pair_ok, pair_wide_ok};
        for (int i = 0; i < GS_NW; ++i) { cost[4 + i] = gcost[i]; size[4 + i] = GS_WIDTH[i]; ok[4 + i] = gemm_ok[i]; }
        std::vector<float> f((size_t)B + 1, 0.f);
        std::vector<unsigned char> take((size_t)B + 1, 0);
        for (int n = 1; n <= B; ++n) {
            float best = 1e30f;
            for (int it = 0; it < NI; ++it) {
                if (!ok[it]) continue;
                if ((it == 2 || it == 3) && n <= size[it - 2]) continue;     // a pair needs more queries than one pass of its kind takes
                if (it > 4 && ok[it - 1] && n <= size[it - 1]) continue;     // a wider pass than the queries need
                if (it == 4 && n <= QWIDE && wide_ok && N < GS_SMALL_BATCH_MIN_ROWS) continue;   // (up to 96 queries on a small shard: the streaming pass)
                const float c = cost[it] + f[n > size[it] ? n - size[it] : 0];
                if (c < best) { best = c; take[n] = (unsigned char)it; }
            }
            f[n] = best;
        }
        for (int r = B; r > 0;) {
            const int it = take[r], m = r < size[it] ? r : size[it];
            if (it < 2) passes.push_back({m, 0, it == 1, false});
            else if (it < 4) passes.push_back({(m + 1) / 2, m / 2, it == 3, false});
            else passes.push_back({m, 0, false, true});
            r -= m;
        }
    }
    // ATLAS_ST_PLAN: five counters in one word, each SATURATING at its field's width (8 | 8 | 4 | 4 | 8 bits): a very large batch on a shard the
    // GEMM-shaped pass does not take must not carry into the next field
    int cnt[5] = {0, 0, 0, 0, 0};       // 64-query passes | 96-query passes | pairs of 64 | pairs of 96 | GEMM-shaped passes
    for (const Pass& ps : passes) ++cnt[ps.gemm ? 4 : ps.nq2 > 0 ? (ps.wide ? 3 : 2) : (ps.wide ? 1 : 0)];
    const int sat[5] = {255, 255, 15, 15, 255}, shift[5] = {0, 8, 16, 20, 24};
    for (int i = 0; i < 5; ++i) bp.plan_word |= (cnt[i] < sat[i] ? cnt[i] : sat[i]) << shift[i];
    bp.single = pl; bp.pair = pp; bp.wide_ok = wide_ok;
    return bp;
}

extern "C" {

#if ATLAS_TUNING
// Test hooks and knobs: the TUNING build only. The product library exports exactly what include/atlas_hip.h declares (tests/test_capi_symbols.py
// compares `nm -D` with the header, both ways).
// test hook: the launch geometry and workspace layout of one GEMM-shaped pass of nq queries (make_gplan), as
// 17 words {ok, G, ncol, cw, ldq, rows_per_range, s_tiles, s_stride, nmax, gcap, off_q16, off_theta, off_gcnt, off_wgstat, off_smax, off_lists, total}
void atlas_test_gplan(int64_t N, int nq, int cus, int64_t* out) {
    const GPlan g = make_gplan(N, nq, cus);
    const int64_t v[17] = {g.ok ? 1 : 0, g.G, g.ncol, g.cw, g.ldq, g.rows_per_range, g.s_tiles, g.s_stride, g.nmax, g.gcap, (int64_t)g.off_q16,
                           (int64_t)g.off_theta, (int64_t)g.off_gcnt, (int64_t)g.off_wgstat, (int64_t)g.off_smax, (int64_t)g.off_lists, (int64_t)g.total};
    for (int i = 0; i < 17; ++i) out[i] = v[i];
}

// test hook: the ATLAS_ST_PLAN word a batch of B queries on N rows would report on a device of `cus`
// compute units with a workspace of atlas_scan_topk_workspace_bytes' size (negative: the error atlas_scan_topk would return); no device needed
int atlas_test_plan_word(int64_t N, int B, int k, int cus) {
    if (B <= 0 || k <= 0 || N < 0) return ATLAS_E_BADARG;
    if (k > K_FAST_MAX || N >= (int64_t)0xffffffffll) return ATLAS_E_UNSUPPORTED;
    const BatchPlan bp = plan_batch(N, B, D_FAST, k, cus, (size_t)-1, kTrusted);
    return bp.rc != 0 ? bp.rc : bp.plan_word;
}

// test hook: the device's double -> fp16 conversions, elementwise
// (out[2i] = software single-rounding path, out[2i+1] = hardware round-to-odd path)
__global__ void dbg_f64_to_f16_kernel(const double* in, uint16_t* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { out[2 * i] = f64_to_f16_bits(in[i]); out[2 * i + 1] = f64_to_f16_bits_rto(in[i]); }
}
int atlas_dbg_f64_to_f16(const double* in, uint16_t* out, int n, void* stream) {
    hipLaunchKernelGGL(dbg_f64_to_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, n);
    return (int)hipGetLastError();
}

// tuning build only (not in include/atlas_hip.h): scan variant, device buffers for cycle stamps
void atlas_tune_set_scan_variant(int v) { g_scan_variant = v; }
void atlas_tune_set_scan_coop(int c) { g_scan_coop = c; }
void atlas_tune_set_scan_dma(int m) { g_scan_dma = m; }
void atlas_tune_set_dma_deal(int d) { g_dma_deal = d; }
void atlas_tune_set_dma_pool_tile(int rows) { g_dma_pool_tile = (rows == 64 || rows == 128) ? rows : 256; }
void atlas_tune_set_gscan_nt(int m) { g_gscan_nt = m; }
void atlas_tune_set_scan_fused(int f) { g_scan_fused = f; }
void atlas_tune_set_scan_wide(int f) { g_scan_wide = f; }
void atlas_tune_set_scan_pair(int f) { g_scan_pair = f; }
void atlas_tune_set_scan_gemm(int f) { g_scan_gemm = f; }
void atlas_tune_set_scan_pool(int permille, int max_per_wg) { g_pool_permille = permille; g_pool_max = max_per_wg; }
// the launch plan of a scan over N rows on a device with `cus` CUs, for host-side checks of its invariants (no GPU needed):
// out = {G, rows_per_wg, pool_begin, pool_rows, pool_tiles, tile, pool_tile, supported (the range checks of atlas_scan_topk)}
void atlas_tune_scan_plan(int64_t N, int k, int cus, int64_t* out) {
    const ScanVariant& var = kVariants[scan_variant_index()];
    const ScanPlan pl = make_plan(N, D_FAST, k, cus, var);
    out[0] = pl.G; out[1] = pl.rows_per_wg; out[2] = pl.pool_begin; out[3] = pl.pool_rows; out[4] = pl.pool_tiles; out[5] = pl.tile;
    out[6] = (var.nw - 1) * var.pf * 16;
    out[7] = scan_plan_supported(pl) ? 1 : 0;
}
void atlas_tune_set_merge_stamps(unsigned long long* p) { g_merge_dbg = p; }
int atlas_tune_spin(int wgs, long long usec, void* stream) {
    allow_lds(atlas_spin_kernel);
    hipLaunchKernelGGL(atlas_spin_kernel, dim3(wgs), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned long long)usec * 100ull, (int*)nullptr);
    return (int)hipGetLastError();
}
void atlas_tune_set_scan_stamps(unsigned long long* p) { g_scan_dbg = p; }
#endif

int atlas_abi_version(void) { return ATLAS_ABI_VERSION; }
const char* atlas_build_info(void) {
#if ATLAS_TUNING
    static char buf[160];
    snprintf(buf, sizeof(buf), "atlas_hip gfx950 %s " __DATE__ " " __TIME__ " tuning", kVariants[scan_variant_index()].name);
    return buf;
#else
    return "atlas_hip gfx950 dscan_kernel<nt> (64-query passes) + scan_kernel<16,1,8> " __DATE__ " " __TIME__;
#endif
}

size_t atlas_scan_topk_workspace_bytes(int64_t N, int B, int d, int k) {
    if (N < 0 || d <= 0 || k <= 0) return 0;
    size_t mx = 0;     // the variant is a run-time choice: size for the largest
    for (int v = 0; v < kNumVariants; ++v) {
        const size_t t = make_plan(N, d, k, device_cus(), kVariants[v], B > QCHUNK ? QWIDE : QCHUNK).total;
        mx = t > mx ? t : mx;
        if (B > QCHUNK) {              // paired passes: two half-chip plans side by side
            const ScanPlan pp = make_plan(N, d, k, device_cus() / 2, kVariants[v], QWIDE);
            const size_t t2 = pp.bulk_begin + 2 * pp.bulk_size;
            mx = t2 > mx ? t2 : mx;
        }
    }
    if (B > QCHUNK && d == D_FAST && k <= K_FAST_MAX) {
        // GEMM-shaped passes (gscan_kernel.h): sized from the passes plan_batch really makes of this batch with room for all of them (the
        // widest pass it chose) -- not from "B > 64": 65..96 queries on a shard below GS_SMALL_BATCH_MIN_ROWS rows never take one, and the
        // 1024-wide layout (~128 MiB of sample maxima + 256 MiB of lists, zero-filled by the caller) is only paid by batches that use it
        const BatchPlan bp = plan_batch(N, B, d, k, device_cus(), (size_t)-1, kTrusted);
        if (bp.rc == 0)
            for (const Pass& ps : bp.passes)
                if (ps.gemm) {
                    const GPlan g = make_gplan(N, ps.nq, device_cus());
                    if (g.ok && g.total > mx) mx = g.total;
                }
    }
    return mx;
}

int atlas_scan_topk(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                    float pmax_hint, void* out_score_f16, int64_t* out_idx, int32_t* out_status, void* ws,
                    size_t ws_bytes, void* stream_) {
    return atlas_scan_topk_ex(q, q_dtype, slab_f16, N, B, d, k, pmax_hint, out_score_f16, out_idx, out_status, ws,
                              ws_bytes, stream_, nullptr, nullptr);
}

int atlas_scan_topk_ex(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                       float pmax_hint, void* out_score_f16, int64_t* out_idx, int32_t* out_status, void* ws,
                       size_t ws_bytes, void* stream_, void* ev_scan_begin, void* ev_scan_end) {
    return atlas_scan_topk_flags(q, q_dtype, slab_f16, N, B, d, k, pmax_hint, out_score_f16, out_idx, out_status, ws,
                                 ws_bytes, stream_, ev_scan_begin, ev_scan_end, 0);
}

int atlas_scan_topk_flags(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                          float pmax_hint, void* out_score_f16, int64_t* out_idx, int32_t* out_status, void* ws,
                          size_t ws_bytes, void* stream_, void* ev_scan_begin, void* ev_scan_end, int flags) {
    return atlas_scan_topk_pack(q, q_dtype, slab_f16, N, B, d, k, pmax_hint, out_score_f16, out_idx, out_status, ws, ws_bytes, stream_,
                                ev_scan_begin, ev_scan_end, flags, 1, 0, nullptr);
}

int atlas_scan_topk_pack(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                         float pmax_hint, void* out_score_f16, int64_t* out_idx, int32_t* out_status, void* ws,
                         size_t ws_bytes, void* stream_, void* ev_scan_begin, void* ev_scan_end, int flags,
                         int64_t id_mul, int64_t id_add, uint64_t* out_packed) {
    if (flags & ~ATLAS_SCAN_TRUST_PMAX) return ATLAS_E_BADARG;
    if (!q || (!slab_f16 && N > 0) || !out_score_f16 || !out_idx || !out_status || !ws) return ATLAS_E_BADARG;
    if (B <= 0 || k <= 0 || N < 0 || q_dtype < 0 || q_dtype > 2 || !(pmax_hint >= 0.f)) return ATLAS_E_BADARG;
    if (d != D_FAST || k > K_FAST_MAX || N >= (int64_t)0xffffffffll) return ATLAS_E_UNSUPPORTED;
    // same shape, same plan; the trusted twin exists for the production variant only
    const bool trusted = (flags & ATLAS_SCAN_TRUST_PMAX) && scan_variant_index() == 0;
    const ScanVariant& var = trusted ? kTrusted : kVariants[scan_variant_index()];
    const BatchPlan bp = plan_batch(N, B, d, k, device_cus(), ws_bytes, var);
    if (bp.rc != 0) return bp.rc;
    const ScanPlan& pl = bp.single;
    const ScanPlan& pp = bp.pair;                // the half-chip plan of paired passes
    const bool wide_ok = bp.wide_ok;
    const std::vector<Pass>& passes = bp.passes;
    const int plan_word = bp.plan_word;
    const ScanVariant& wide = trusted ? kWideTrusted : kWide;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* w = (unsigned char*)ws;

    auto merge = merge_rescore_kernel<MERGE_NT>;
    allow_lds(var.kern);
    if (wide_ok) allow_lds(wide.kern);
    allow_lds(merge);
    allow_lds(sample_scores_kernel);
    const DmaVariant& dma = kDma[trusted ? 1 : 0];
    allow_lds(dma.kern);
#if ATLAS_TUNING
    allow_lds(kDmaDefaultPolicy[trusted ? 1 : 0].kern);
#endif

    hipError_t e = hipSuccess;
    const ScanPlan single = pl;
    int q0 = 0;
    for (size_t ci = 0; ci < passes.size(); q0 += passes[ci].nq + passes[ci].nq2, ++ci) {
        const int nq = passes[ci].nq, nq2 = passes[ci].nq2;
        const bool is_wide = passes[ci].wide, paired = nq2 > 0;
        if (passes[ci].gemm) {
            // GEMM-shaped pass: queries -> fp16 rows, sample launch -> fragment maxima, thresholds, scan launch, merge (flat lists)
            const GPlan g = make_gplan(N, nq, device_cus());
            auto gsample = g.cw == 256 ? gscan_kernel<1, 4> : g.cw == 192 ? gscan_kernel<1, 3> : gscan_kernel<1, 2>;
            auto gscan = g.cw == 256 ? (trusted ? gscan_kernel<0, 4> : gscan_kernel<2, 4>)          // <2, .>: the twin that measures every row norm
                       : g.cw == 192 ? (trusted ? gscan_kernel<0, 3> : gscan_kernel<2, 3>)
                                     : (trusted ? gscan_kernel<0, 2> : gscan_kernel<2, 2>);
            if (g.ncol == 1 && gscan_nt_enabled())        // one column tile: every slab line is read once -> nt on the slab DMA (gscan_kernel.h: NT)
                gscan = g.cw == 256 ? (trusted ? gscan_kernel<0, 4, 1> : gscan_kernel<2, 4, 1>)
                      : g.cw == 192 ? (trusted ? gscan_kernel<0, 3, 1> : gscan_kernel<2, 3, 1>)
                                    : (trusted ? gscan_kernel<0, 2, 1> : gscan_kernel<2, 2, 1>);
            const size_t g_lds = GS_LDS_BYTES;
            allow_lds(gsample); allow_lds(gscan); allow_lds(gtheta_kernel);
            uint16_t* q16 = (uint16_t*)(w + g.off_q16);
            uint32_t* gcnt = (uint32_t*)(w + g.off_gcnt);
            hipLaunchKernelGGL(gprep_kernel, dim3(nq), dim3(96), 0, stream, q, q_dtype, q0, (uint4*)q16, gcnt, out_status);
            GScanParams gs{};
            gs.slab = (const uint16_t*)slab_f16; gs.N = N; gs.q16 = q16; gs.nq = nq; gs.ncol = g.ncol; gs.rows_per_range = g.rows_per_range;
            gs.s_tiles = g.s_tiles; gs.s_stride = g.s_stride; gs.theta = (const float*)(w + g.off_theta); gs.smax = (float*)(w + g.off_smax);
            gs.lists = (uint2*)(w + g.off_lists); gs.gcnt = gcnt; gs.qflag = (uint32_t*)(w + GS_OFF_QFLAG); gs.gcap = g.gcap;
            gs.wg_stat = (uint32_t*)(w + g.off_wgstat); gs.pmax2_hint = pmax_hint * pmax_hint;
            gs.dbg = g_scan_dbg;
            hipLaunchKernelGGL(gsample, dim3(g.G), dim3(512), g_lds, stream, gs);
            hipLaunchKernelGGL(gtheta_kernel, dim3(g.ldq), dim3(256), GTHETA_LDS(g.nmax), stream, (const float*)gs.smax, g.nmax, g.ldq, (const uint16_t*)q16, nq,
                               pmax_hint, k, (float*)(w + g.off_theta), (const uint2*)nullptr, (const uint32_t*)nullptr, 0);
            // the scan, in two launches: the first eighth of every row range with the sample's thresholds, then -- thresholds tightened by what
            // that found (the k-th best of N / 8 rows instead of the k-th best of the N / 64 sampled ones) -- the rest: ~4 x fewer candidates
            // per query, and the filter epilogue of most tiles finds nothing to do
            const int tiles_per_range = (int)(g.rows_per_range / GS_TILE);
            const int t1 = tiles_per_range >= 16 ? (tiles_per_range + 7) / 8 : tiles_per_range;
            if (q0 == 0 && ev_scan_begin) (void)hipEventRecord((hipEvent_t)ev_scan_begin, stream);
            gs.tile_begin = 0; gs.tile_end = t1;
            hipLaunchKernelGGL(gscan, dim3(g.G), dim3(512), g_lds, stream, gs);
            if (t1 < tiles_per_range) {
                hipLaunchKernelGGL(gtheta_kernel, dim3(g.ldq), dim3(256), GTHETA_LDS(GTHETA_MAXKEYS), stream, (const float*)gs.smax, g.nmax, g.ldq, (const uint16_t*)q16, nq,
                                   pmax_hint, k, (float*)(w + g.off_theta), (const uint2*)gs.lists, (const uint32_t*)gcnt, g.gcap);
                gs.tile_begin = t1; gs.tile_end = tiles_per_range; gs.wg_stat += (size_t)g.G * 2;      // (one word pair per workgroup and launch)
                hipLaunchKernelGGL(gscan, dim3(g.G), dim3(512), g_lds, stream, gs);
            }
            if (q0 == 0 && ev_scan_end) (void)hipEventRecord((hipEvent_t)ev_scan_end, stream);
            MergeParams mp{};
            mp.slab = (const uint16_t*)slab_f16; mp.N = N; mp.d = d;
            mp.q = q16; mp.q_dtype = ATLAS_DT_F16; mp.qbase = 0; mp.pmax = pmax_hint; mp.pmax_trusted = trusted ? 1 : 0;
            mp.lists = gs.lists; mp.list_cnt = nullptr; mp.wg_stat = (uint32_t*)(w + g.off_wgstat); mp.G = 1024; mp.cap = g.gcap / 1024;
            mp.flat_cnt = gcnt; mp.nstat = (t1 < tiles_per_range ? 2 : 1) * g.G;
            mp.total_cap = g.gcap; mp.epoch = (uint32_t*)(w + single.off_epoch); mp.ticket = (uint32_t*)(w + single.off_epoch + 128);
            mp.qflag = gs.qflag; mp.k = k; mp.q0 = q0; mp.key_cap = single.key_cap;
            mp.dbg = g_merge_dbg;
            mp.out_score = (uint16_t*)out_score_f16; mp.out_idx = out_idx; mp.out_status = out_status;
            mp.out_packed = out_packed; mp.id_mul = id_mul; mp.id_add = id_add;
            mp.nq1 = nq; mp.nq2 = 0; mp.pair_state = PAIR_STATE; mp.pair_bulk = 0; mp.plan_word = plan_word;
            hipLaunchKernelGGL(merge, dim3(nq, 1), dim3(MERGE_NT), single.merge_lds, stream, mp);
            continue;
        }
        const ScanPlan& pl = paired ? pp : single;                          // (shadows the whole-chip plan inside the loop)
        // initial thresholds: the scan derives them from the tile maxima of an evenly spread sample; small shards
        // start at -inf. Whoever runs first clears the per-call state (per-query fallback flags, status header).
        // coop scan: the workgroups' first tiles are the sample (one launch less); needs one granule per (query, workgroup) lane slot
        const bool coop = pl.S > 0 && pl.G >= (is_wide ? QWIDE : QCHUNK) && pl.G <= 256 && scan_coop_enabled();
        if (coop) {
            // nothing to launch: workgroup 0 of the scan clears the status header, the flags were cleared by the previous merge
        } else if (pl.S > 0) {
            SampleParams sm{};
            sm.slab = (const uint16_t*)slab_f16; sm.N = N; sm.q = q; sm.q_dtype = q_dtype; sm.q0 = q0; sm.nq = nq;
            sm.top2 = (float*)(w + pl.off_sample); sm.S = pl.S; sm.stride_rows = pl.sample_stride;
            sm.qflag = (uint32_t*)(w + pl.off_qflag); sm.theta_gran = (unsigned long long*)(w + pl.off_gran); sm.q16 = (uint4*)(w + pl.off_q16); sm.out_status = out_status;
            hipLaunchKernelGGL(sample_scores_kernel, dim3(pl.S / 64), dim3(SAMPLE_NT), (size_t)QIMG_U4 * 16, stream, sm);
        } else {
            e = hipMemsetAsync(w + pl.off_qflag, 0, 512, stream);
            if (e == hipSuccess && q0 == 0) e = hipMemsetAsync(out_status, 0, ATLAS_STATUS_HEADER * 4, stream);
            if (e != hipSuccess) return (int)e;
        }
        ScanParams sp{};
        sp.slab = (const uint16_t*)slab_f16; sp.N = N;
        // after a sample pass the queries of this pass exist as fp16 rows in the workspace (its blocks wrote them): half the bytes, no conversion
        const bool have_q16 = !coop && pl.S > 0 && pl.S / 64 >= QCHUNK;
        sp.q = have_q16 ? (const void*)(w + pl.off_q16) : q; sp.q_dtype = have_q16 ? ATLAS_DT_F16 : q_dtype; sp.q0 = have_q16 ? 0 : q0;
        sp.pmax = pmax_hint;
        sp.top2 = (!coop && pl.S > 0 && pl.G >= QCHUNK) ? (const float*)(w + pl.off_sample) : nullptr; sp.sample_blocks = pl.S / 64;
        sp.theta_gran = (unsigned long long*)(w + pl.off_gran);
        sp.coop = coop ? 1 : 0; sp.gran_max = (unsigned long long*)(w + pl.off_gmax); sp.epoch = (const uint32_t*)(w + pl.off_epoch);
        sp.out_status = out_status;
        sp.lists = (uint2*)(w + pl.off_lists);
        sp.list_cnt = (uint32_t*)(w + pl.off_list_cnt); sp.wg_stat = (uint32_t*)(w + pl.off_wg_stat);
        sp.qflag = (uint32_t*)(w + pl.off_qflag);
        sp.pool_begin = pl.pool_begin; sp.pool_rows = pl.pool_rows; sp.pool_tiles = pl.pool_tiles; sp.ticket = (uint32_t*)(w + pl.off_epoch + 128);
        sp.rows_per_wg = pl.rows_per_wg; sp.nq = nq; sp.k = k; sp.cap = pl.cap; sp.keep_max = pl.keep_max;
        sp.buf_cap = is_wide ? pl.buf_cap_wide : pl.buf_cap; sp.flush_at = is_wide ? pl.flush_at_wide : pl.flush_at;
        sp.pmax2_hint = pmax_hint * pmax_hint;
        sp.nq2 = nq2; sp.pair_state = PAIR_STATE; sp.pair_bulk = pl.bulk_size;
        sp.dbg = g_scan_dbg;
        if (q0 == 0 && ev_scan_begin) (void)hipEventRecord((hipEvent_t)ev_scan_begin, stream);
        MergeParams mp{};
        mp.slab = (const uint16_t*)slab_f16; mp.N = N; mp.d = d;
        mp.q = sp.q; mp.q_dtype = sp.q_dtype; mp.qbase = sp.q0; mp.pmax = pmax_hint; mp.pmax_trusted = trusted ? 1 : 0;
        mp.lists = sp.lists; mp.list_cnt = sp.list_cnt; mp.wg_stat = sp.wg_stat; mp.G = pl.G; mp.cap = pl.cap;
        mp.total_cap = pl.total_cap; mp.epoch = (uint32_t*)(w + pl.off_epoch); mp.ticket = sp.ticket;
        mp.qflag = sp.qflag; mp.k = k; mp.q0 = q0; mp.key_cap = pl.key_cap;
        mp.dbg = g_merge_dbg;
        mp.out_score = (uint16_t*)out_score_f16; mp.out_idx = out_idx; mp.out_status = out_status;
        mp.out_packed = out_packed; mp.id_mul = id_mul; mp.id_add = id_add;
        mp.nq1 = nq; mp.nq2 = nq2; mp.pair_state = PAIR_STATE; mp.pair_bulk = pl.bulk_size; mp.plan_word = plan_word;
#if ATLAS_TUNING
        // experiment (scan_kernel.h, ScanParams::fused): the merge inside the scan's last nq workgroups
        const bool fused = !paired && !is_wide && coop && var.nw * 64 == MERGE_NT && pl.G >= nq && pl.merge_lds <= pl.scan_lds && scan_fused_enabled();
        sp.fused = fused ? 1 : 0; sp.fuse = (uint32_t*)(w + pl.off_fuse); sp.mp = mp;
#else
        constexpr bool fused = false;
#endif
        const dim3 sgrid(pl.G, paired ? 2 : 1);
        // coop, unpaired, 64-query passes of the production shape: the DMA-staged kernel (dscan_kernel.h). Its tiles are 256 rows, pool tiles
        // included (all eight waves take rows: the ticket is one lane's returning atomic, not a wave's job), and its candidate buffer is what
        // four 32 KiB stages leave of the LDS
        const bool use_dma = coop && !paired && !is_wide && !fused && scan_variant_index() == 0 && scan_dma_mode() != 0 && pl.tile == DS_TILE;
        if (use_dma) {
            sp.deal = dma_deal();
            sp.pool_tile_rows = dma_pool_tile_rows();
            sp.pool_tiles = (pl.pool_rows + sp.pool_tile_rows - 1) / sp.pool_tile_rows;
            sp.buf_cap = DS_BUF_CAP; sp.flush_at = DS_BUF_CAP * 3 / 4;
#if ATLAS_TUNING
            const DmaVariant& dk = scan_dma_mode() == 2 ? kDmaDefaultPolicy[trusted ? 1 : 0] : dma;
#else
            const DmaVariant& dk = dma;
#endif
            hipLaunchKernelGGL(dk.kern, dim3(pl.G), dim3(DS_NW * 64), (size_t)DScanSmem::total, stream, sp);
        } else if (is_wide) hipLaunchKernelGGL(wide.kern, sgrid, dim3(wide.nw * 64), pl.scan_lds_wide, stream, sp);
        else hipLaunchKernelGGL(var.kern, sgrid, dim3(var.nw * 64), pl.scan_lds, stream, sp);
        if (q0 == 0 && ev_scan_end) (void)hipEventRecord((hipEvent_t)ev_scan_end, stream);
        if (!fused) hipLaunchKernelGGL(merge, dim3(nq, paired ? 2 : 1), dim3(MERGE_NT), pl.merge_lds, stream, mp);
    }
    return (int)hipGetLastError();
}

size_t atlas_exact_topk_workspace_bytes(int64_t N, int B, int d, int k) {
    if (N < 0 || d <= 0 || k <= 0 || B <= 0) return 0;
    return make_exact_plan(N, d, k, B).total;
}

int atlas_exact_topk(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                     void* out_score_f16, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream_) {
    if (!q || (!slab_f16 && N > 0) || !out_score_f16 || !out_idx || !ws) return ATLAS_E_BADARG;
    if (B <= 0 || k <= 0 || N < 0 || d <= 0 || q_dtype < 0 || q_dtype > 2) return ATLAS_E_BADARG;
    if (k > K_EXACT_MAX || d > 8192 || N >= (int64_t)0xffffffffll) return ATLAS_E_UNSUPPORTED;
    const ExactPlan pl = make_exact_plan(N, d, k, B);
    if (ws_bytes < pl.total) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* w = (unsigned char*)ws;
    const int cus = device_cus();
    int grid = (int)((N + 255) / 256);
    if (grid > cus * 8) grid = cus * 8;
    if (grid < 1) grid = 1;
    int grid_rows = (int)((N + 3) / 4 > (int64_t)cus * 8 ? (int64_t)cus * 8 : (N + 3) / 4);
    if (grid_rows < 1) grid_rows = 1;
    const size_t keys_lds = align_up((size_t)EXACT_QB * d * 2, 16);
    allow_lds(exact_keys_batch_kernel<true>);
    allow_lds(exact_keys_batch_kernel<false>);
    for (int q0 = 0; q0 < B; q0 += QCHUNK) {
        const int nq = (B - q0 < QCHUNK) ? (B - q0) : QCHUNK;
        // prep reuses the fast path's converter (no fragment image: qfrag == nullptr)
        hipLaunchKernelGGL(prep_queries_kernel, dim3(QCHUNK), dim3(256), 0, stream, q, q_dtype, q0, nq, d, 0.f,
                           (uint16_t*)(w + pl.off_qrow), (uint16_t*)nullptr, (float*)(w + pl.off_qeps),
                           (uint32_t*)nullptr, (int32_t*)nullptr);
        for (int j0 = 0; j0 < nq; j0 += EXACT_QB) {
            const int nb = (nq - j0 < EXACT_QB) ? (nq - j0) : EXACT_QB;
            hipError_t e = hipMemsetAsync(w + pl.off_state, 0, (size_t)EXACT_QB * EXACT_STATE_WORDS * 4, stream);
            if (e != hipSuccess) return (int)e;
            const uint16_t* qb = (const uint16_t*)(w + pl.off_qrow) + (size_t)j0 * d;
            if (d == D_FAST)
                hipLaunchKernelGGL(exact_keys_batch_kernel<true>, dim3(grid_rows), dim3(256), keys_lds, stream,
                                   (const uint16_t*)slab_f16, N, d, qb, nb, (uint64_t*)(w + pl.off_keys));
            else
                hipLaunchKernelGGL(exact_keys_batch_kernel<false>, dim3(grid_rows), dim3(256), keys_lds, stream,
                                   (const uint16_t*)slab_f16, N, d, qb, nb, (uint64_t*)(w + pl.off_keys));
            for (int pass = 0; pass < EXACT_PASSES; ++pass)
                hipLaunchKernelGGL(exact_hist_kernel, dim3(grid, nb), dim3(256), 0, stream,
                                   (const uint64_t*)(w + pl.off_keys), N, k, pass, (uint32_t*)(w + pl.off_state));
            hipLaunchKernelGGL(exact_collect_kernel, dim3(grid, nb), dim3(256), 0, stream,
                               (const uint64_t*)(w + pl.off_keys), N, k, (uint32_t*)(w + pl.off_state),
                               (uint64_t*)(w + pl.off_sel));
            hipLaunchKernelGGL(exact_emit_kernel, dim3(nb), dim3(256), 0, stream, (const uint64_t*)(w + pl.off_sel),
                               (const uint32_t*)(w + pl.off_state), k,
                               (uint16_t*)out_score_f16 + (size_t)(q0 + j0) * k, out_idx + (size_t)(q0 + j0) * k);
        }
    }
    return (int)hipGetLastError();
}

int atlas_pack_candidates(const void* score_f16, const int64_t* idx, int64_t n, int64_t id_mul, int64_t id_add,
                          uint64_t* out_packed, void* stream_) {
    if (!score_f16 || !idx || !out_packed || n < 0) return ATLAS_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_candidates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (const uint16_t*)score_f16, idx, n, id_mul, id_add, out_packed);
    return (int)hipGetLastError();
}

int atlas_merge_packed(const uint64_t* gathered, int W, int B, int k, uint64_t* out_packed, void* stream_) {
    if (!gathered || !out_packed || W <= 0 || B <= 0 || k <= 0) return ATLAS_E_BADARG;
    if ((size_t)W * k > 8192) return ATLAS_E_UNSUPPORTED;
    hipLaunchKernelGGL(merge_packed_kernel, dim3(B), dim3(256), (size_t)W * k * 8, (hipStream_t)stream_, gathered, W,
                       B, k, out_packed);
    return (int)hipGetLastError();
}

size_t atlas_xchg_bytes(int W, int64_t slot_entries) {
    if (W <= 0 || W > XCHG_MAXW || slot_entries <= 0) return 0;
    return (size_t)XCHG_HDR + (size_t)2 * W * (size_t)slot_entries * 8;
}

int atlas_xchg_create(int W, int64_t slot_entries, void** buf, unsigned char* handle64) {
    if (!buf || !handle64 || atlas_xchg_bytes(W, slot_entries) == 0) return ATLAS_E_BADARG;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, atlas_xchg_bytes(W, slot_entries));
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, atlas_xchg_bytes(W, slot_entries));
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI carries IPC handles as 64 bytes");
    memcpy(handle64, &h, 64);
    *buf = p;
    return 0;
}

int atlas_xchg_open(const unsigned char* handle64, void** peer_buf) {
    if (!handle64 || !peer_buf) return ATLAS_E_BADARG;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    return (int)hipIpcOpenMemHandle(peer_buf, h, hipIpcMemLazyEnablePeerAccess);
}

int atlas_xchg_close(void* peer_buf) { return peer_buf ? (int)hipIpcCloseMemHandle(peer_buf) : ATLAS_E_BADARG; }
int atlas_xchg_destroy(void* buf) { return buf ? (int)hipFree(buf) : ATLAS_E_BADARG; }

int atlas_xchg_push(const uint64_t* packed, int64_t n, void* const* peer_bufs, int W, int rank, int64_t slot_entries, uint32_t tag, void* stream_) {
    if (!packed || !peer_bufs || W <= 0 || W > XCHG_MAXW || rank < 0 || rank >= W || n < 0 || n > slot_entries || tag == 0) return ATLAS_E_BADARG;
    XchgPeers pp{};
    for (int i = 0; i < W; ++i) { if (!peer_bufs[i]) return ATLAS_E_BADARG; pp.buf[i] = (unsigned char*)peer_bufs[i]; }
    hipLaunchKernelGGL(xchg_push_kernel, dim3(W), dim3(256), 0, (hipStream_t)stream_, packed, n, pp, rank, W, slot_entries, tag);
    return (int)hipGetLastError();
}

int atlas_xchg_merge(const void* own_buf, int W, int B, int k, int64_t slot_entries, uint32_t tag, int wait_ms, uint64_t* out_packed,
                     int32_t* status, void* stream_) {
    if (!own_buf || !out_packed || !status || W <= 0 || W > XCHG_MAXW || B <= 0 || k <= 0 || (int64_t)B * k > slot_entries || tag == 0 || wait_ms <= 0)
        return ATLAS_E_BADARG;
    if ((size_t)W * k > 8192) return ATLAS_E_UNSUPPORTED;
    hipLaunchKernelGGL(xchg_merge_kernel, dim3(B), dim3(256), (size_t)W * k * 8, (hipStream_t)stream_, (const unsigned char*)own_buf, W, B, k,
                       slot_entries, tag, (unsigned long long)wait_ms * 100000ull, out_packed, status);
    return (int)hipGetLastError();
}

int atlas_pool_write(const void* hidden_f16, const int64_t* mask, void* slab_f16, int64_t N, int64_t row_offset,
                     int n, int L, int d, void* stream_) {
    if (!hidden_f16 || !mask || !slab_f16 || n < 0 || L <= 0 || d <= 0) return ATLAS_E_BADARG;
    if (row_offset < 0 || row_offset + n > N) return ATLAS_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pool_write_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)hidden_f16,
                       mask, (uint16_t*)slab_f16, row_offset, L, d);
    return (int)hipGetLastError();
}

int atlas_slab_pmax(const void* slab_f16, int64_t N, int d, float* out_pmax, void* stream_) {
    if ((!slab_f16 && N > 0) || !out_pmax || d <= 0 || N < 0) return ATLAS_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(out_pmax, 0, 4, stream);
    if (e != hipSuccess) return (int)e;
    if (N == 0) return 0;
    int grid = (int)((N + 3) / 4 > 65536 ? 65536 : (N + 3) / 4);
    if (grid > device_cus() * 16) grid = device_cus() * 16;
    if (d == D_FAST && ((uintptr_t)slab_f16 & 15) == 0)
        hipLaunchKernelGGL(slab_pmax768_kernel, dim3(device_cus() * 8), dim3(256), 0, stream, (const uint16_t*)slab_f16, N, (uint32_t*)out_pmax);
    else
        hipLaunchKernelGGL(slab_pmax_kernel, dim3(grid), dim3(256), 0, stream, (const uint16_t*)slab_f16, N, d,
                           (uint32_t*)out_pmax);
    return (int)hipGetLastError();
}

}  // extern "C"
