// scan_kernel.h — the fused MFMA scan + candidate-list kernel, as a template so that the
// product (atlas_hip.hip) and the tuning harness (microbench.hip) instantiate the same source.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "common.h"
#include "../../include/atlas_hip.h"

#ifndef ATLAS_TUNING
#define ATLAS_TUNING 0
#endif
// wall-clock stamps exist in the tuning build only: the product kernel carries no trace of them
#if ATLAS_TUNING
#define ATLAS_SCAN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define ATLAS_SCAN_STAMP_LANE0(i) do { if (p.dbg && lane == 0) p.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATLAS_SCAN_STAMP(i) do { } while (0)
#define ATLAS_SCAN_STAMP_LANE0(i) do { } while (0)
#endif

namespace atlas {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define D_FAST 768               // EMBEDDINGS_DIM, src/retrievers.py:13
#define KSTEPS (D_FAST / 32)     // 24 MFMA k-steps of 32
#define QCHUNK 64                // queries per slab pass (4 MFMA column groups of 16)
#define QROW_U4 98               // uint4 per query row of the LDS image: 96 of data + 2 of padding
#define QIMG_U4 (QCHUNK * QROW_U4)   // 100 352 bytes
// Every in-kernel wait for another workgroup is bounded in WALL-CLOCK time (100 MHz ticks): 40 us per hop, ten times what a hop takes when
// all workgroups are resident (~3 us). A workgroup that is not running yet (CUs held by another stream's kernel) then costs the others 40 us
// and a looser threshold, not milliseconds (the bound used to be 4 000 polls, and a poll under load is a ~1.5 us round trip)
#define ATLAS_SPIN_TICKS 4000ull

static __device__ __forceinline__ float neg_inf() { return bits_f32(0xff800000u); }
static __device__ __forceinline__ float pos_inf() { return bits_f32(0x7f800000u); }

// raw workgroup barrier that orders LDS only: prefetched global loads stay in flight
// (a __syncthreads() here would drain vmcnt once per tile; cdna guide §5 "Pipelining across barriers")
static __device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// The query image. Every kernel that needs the queries on the matrix cores builds it ITSELF in LDS from the caller's
// query tensor (fp32 | fp16 | bf16, converted RNE = `.half()`, src/index.py:117): there is no preparation kernel and no
// image in global memory. Layout: row-major fp16, one 1568-byte row per query (768 halfs + 32 bytes of padding). The MFMA B
// operand of v_mfma_f32_16x16x32_f16 for lane l = (query l&15 of a 16-query group, k-group l>>4) at k-step s is the 16 bytes
// at row[64 s + 16 (l>>4)]: one ds_read_b128; with the 98-uint4 row pitch the 16 lanes of every LDS service group hit 16
// different 4-bank windows (bank = 8 (l&15) + 4 (l>>4) mod 64): conflict-free.
// ------------------------------------------------------------------------------------------
struct QRaw { uint4 a, b; };      // 8 consecutive query elements as loaded (fp32: both words; 16-bit types: a only)
static __device__ __forceinline__ QRaw load_q8(const void* __restrict__ q, const int q_dtype, const size_t elem) {
    QRaw r; r.b = make_uint4(0, 0, 0, 0);
    if (q_dtype == ATLAS_DT_F32) { const uint4* p = (const uint4*)((const float*)q + elem); r.a = p[0]; r.b = p[1]; }
    else r.a = *(const uint4*)((const uint16_t*)q + elem);
    return r;
}
static __device__ __forceinline__ uint32_t pack_h2(const float x, const float y) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)x) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)y) << 16);   // v_cvt_f16_f32: RNE
}
static __device__ __forceinline__ uint4 q8_to_f16(const QRaw r, const int q_dtype) {
    if (q_dtype == ATLAS_DT_F16) return r.a;
    if (q_dtype == ATLAS_DT_F32)
        return make_uint4(pack_h2(bits_f32(r.a.x), bits_f32(r.a.y)), pack_h2(bits_f32(r.a.z), bits_f32(r.a.w)),
                          pack_h2(bits_f32(r.b.x), bits_f32(r.b.y)), pack_h2(bits_f32(r.b.z), bits_f32(r.b.w)));
    const uint32_t w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};               // bf16 -> f32 is exact, then RNE to fp16
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_h2(bits_f32(w[i] << 16), bits_f32(w[i] & 0xffff0000u));
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// queries [q0, q0 + nq) of `q` -> s_q; rows >= nq are zero. All loads of a batch are in flight before the first conversion: the element
// type is switched on ONCE per phase, outside the chunk loops -- with the switch inside load_q8 / q8_to_f16 every fp32 chunk was its own
// basic block (second half loaded, WAITED for and converted, then the first half requested): six serialised trips to the L2 per thread.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_first_loads` runs once, when the first batch of loads has been issued and before anything waits for them
template <int NT, class Hook = NoHook, int NQ = QCHUNK>
static __device__ __forceinline__ void fill_query_image(uint4* __restrict__ s_q, const void* __restrict__ q, const int q_dtype,
                                                        const int q0, const int nq, const int tid, Hook after_first_loads = Hook()) {
    constexpr int CH = NQ * (D_FAST / 8);              // 6144 chunks of 8 elements (64 queries)
    constexpr int PER = CH / NT;                       // 6 per thread with 1024 threads, 24 with 256
    constexpr int BATCH = PER % 6 == 0 ? 6 : PER % 4 == 0 ? 4 : PER % 3 == 0 ? 3 : PER % 2 == 0 ? 2 : 1;   // loads in flight per thread
    static_assert(CH % NT == 0 && PER % BATCH == 0, "chunks must divide over the threads");
#pragma unroll 1
    for (int b0 = 0; b0 < PER; b0 += BATCH) {
        QRaw raw[BATCH];
        size_t elem[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int c = tid + (b0 + u) * NT, qi = c / (D_FAST / 8), kc = c - qi * (D_FAST / 8);
            const int qs = qi < nq ? qi : 0;                                         // clamped: loads stay unconditional
            elem[u] = (size_t)(q0 + qs) * D_FAST + (size_t)kc * 8;
        }
        if (q_dtype == ATLAS_DT_F32) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) { const uint4* p = (const uint4*)((const float*)q + elem[u]); raw[u].a = p[0]; raw[u].b = p[1]; }
        } else {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) { raw[u].a = *(const uint4*)((const uint16_t*)q + elem[u]); raw[u].b = make_uint4(0, 0, 0, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (b0 == 0) after_first_loads();
        // every load of the batch is issued and waited for HERE, together: hipcc otherwise requests an fp32 chunk's second half, waits,
        // converts it (two registers instead of four) and only then requests the first half -- a trip to the L2 per chunk
        auto pin = [](uint4& v) {
            u32x4 t = {v.x, v.y, v.z, v.w};
            asm volatile("" : "+v"(t));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        };
        if (q_dtype == ATLAS_DT_F32) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) { pin(raw[u].a); pin(raw[u].b); }
        } else {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) pin(raw[u].a);
        }
        uint4 h[BATCH];
        if (q_dtype == ATLAS_DT_F32) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) h[u] = q8_to_f16(raw[u], ATLAS_DT_F32);
        } else if (q_dtype == ATLAS_DT_F16) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) h[u] = raw[u].a;
        } else {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) h[u] = q8_to_f16(raw[u], ATLAS_DT_BF16);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int c = tid + (b0 + u) * NT, qi = c / (D_FAST / 8), kc = c - qi * (D_FAST / 8);
            s_q[qi * QROW_U4 + kc] = qi < nq ? h[u] : make_uint4(0, 0, 0, 0);
        }
    }
}
// sum of squares of query row qi of the image (one wave; every lane returns the total). fp32 accumulation of 768 exact products:
// relative error <= 768 * 2^-24 = 4.6e-5, covered by the margin in query_eps
static __device__ __forceinline__ float image_row_sumsq(const uint4* __restrict__ s_q, const int qi, const int lane) {
    float ss = 0.f;
    for (int c = lane; c < D_FAST / 8; c += 64) {
        const uint4 v = s_q[qi * QROW_U4 + c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 h = __builtin_bit_cast(f16x2, w[i]);
            ss = __builtin_amdgcn_fdot2(h, h, ss, false);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    return ss;
}
// eps = GAMMA * |q| * pmax, rounded up (common.h: the certified bound on |approximate - exact| score)
static __device__ __forceinline__ float query_eps(const float sumsq, const float pmax) {
    return ATLAS_GAMMA * (sqrtf(sumsq) * 1.0001f) * pmax * 1.000001f;
}

// k-th largest of KPL register keys per lane across ONE wave (keys == 0 are padding): greedy bit
// search below the wave's common key prefix, counting with v_cmp + s_bcnt1 only (no LDS, no barrier).
// Returns the largest v (to `res_bits` bits below the first differing bit) with count(keys >= v) >= kk.
template <int KPL>
static __device__ __forceinline__ uint32_t wave_kth_key(const uint32_t (&key)[KPL], const uint32_t kk, const int res_bits) {
    uint32_t kmax = 0, kmin = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < KPL; ++u) { kmax = key[u] > kmax ? key[u] : kmax; kmin = key[u] < kmin ? key[u] : kmin; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmax, o), b = __shfl_xor(kmin, o);
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
    }
    const uint32_t diff = kmax ^ kmin;
    const int top = diff ? 31 - __builtin_clz(diff) : -1;
    uint32_t prefix = (top < 0) ? kmax : ((top >= 31) ? 0u : (kmax & ~((2u << top) - 1u)));
    const int stop = top - res_bits > 0 ? top - res_bits : 0;
    for (int bit = top; bit >= stop; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < KPL; ++u) c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(key[u] >= cand));
        if (c >= kk) prefix = cand;
    }
    return prefix;
}

// Certified initial threshold of one query from the sample pre-pass (one wave): the k-th largest of the 2 * nblk tile maxima
// written by sample_scores_kernel (nblk <= 256 -> 8 keys per lane), entirely in registers. All of them are approximate scores
// of distinct slab rows, so prune_threshold(k-th, eps) is a valid threshold for the scan; fewer than k of them: -inf.
static __device__ __forceinline__ void load_sample_maxima(const float* __restrict__ top2_q /*[nblk][2]*/, const int nblk, const int lane,
                                                          float (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = lane + u * 64; v[u] = top2_q[2 * (i < nblk ? i : nblk - 1)]; }   // unconditional (clamped): all in flight
}
// The k-th largest of the <= 256 tile MAXIMA (the runner-up of each tile is left out: any subset of real scores gives a valid, slightly
// lower threshold -- the k-th of the maxima is about the (1.08 k)-th of the whole sample -- and the search is half as long), 14 bits below
// the first bit in which they differ (a threshold 2^-14 of the score range lower lets ~0.1 % more candidates through).
static __device__ __forceinline__ float initial_theta(const float (&v)[4], const int nblk, const int k, const float eps, const int lane) {
    uint32_t key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = (lane + u * 64 < nblk && v[u] > neg_inf()) ? f32_order_key(v[u]) : 0u;   // 0 = padding, below every real key
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) valid += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(key[u] != 0u));
    const uint32_t kth = wave_kth_key<4>(key, (uint32_t)k, 14);
    return (valid >= (uint32_t)k) ? prune_threshold(f32_from_order_key(kth), eps) : neg_inf();
}

#include "merge_kernel.h"

// ------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------
struct ScanParams {
    const uint16_t* slab;     // [N][768] fp16
    int64_t N;
    const void* q;            // the caller's queries [B][768], element type q_dtype; this pass takes rows [q0, q0 + nq)
    int q_dtype, q0;
    float pmax;               // upper bound on the slab's row norms (eps = GAMMA |q| pmax)
    const float* top2;        // [64][2 * sample_blocks] tile maxima of the sample pre-pass, or null: thresholds start at -inf
    int sample_blocks;
    unsigned long long* theta_gran;   // [64] {tag, threshold bits}: workgroup q publishes query q's initial threshold, everybody collects
                                      // all 64 (two-kernel mode: cleared by the sample kernel that ran before; tag = 1)
    int coop;                 // 1: no sample kernel -- the workgroups' own FIRST TILES are the sample (see first_tile_exchange below)
    unsigned long long* gran_max;     // coop: [64][G] {tag, tile maximum bits}: workgroup g's best first-tile score for query q
    const uint32_t* epoch;    // coop: per-workspace call counter (bumped by the merge kernel); granules of this call carry tag = epoch + 1
    int32_t* out_status;      // coop: workgroup 0 clears the status header of the call's first 64-query pass
    uint2* lists;             // [64][G][cap]  {f32 bits of approx score, row}: the candidates of workgroup g for query q
    uint32_t* list_cnt;       // [64][G] entries of lists[q][g] at the end of the scan (plain stores: every workgroup writes its 64)
    uint32_t* wg_stat;        // [G][2]  per workgroup: largest row sum of squares seen (float bits) | ATLAS_F_* flags
    uint32_t* qflag;          // [64] per-query fallback flag (band overflow; plain idempotent stores)
    int64_t rows_per_wg;      // rows of every workgroup's STATIC range [g * rows_per_wg, +rows_per_wg) (clipped to N)
    // the tail of the slab, [pool_begin, N), is not pre-assigned: it is handed out in tiles at run time (pool_tiles == 0: no pool).
    // Pool tile g is workgroup g's first one; further tiles come from the ticket counter (tile = G + ticket), see the main loop.
    int64_t pool_begin;
    int pool_rows, pool_tiles;
    int deal;                 // dscan_kernel.h only: 1 = the static part of the slab is dealt to the workgroups tile by tile (the product), 0 = one contiguous range per workgroup (A/B)
    int pool_tile_rows;       // dscan_kernel.h only: rows of a pool tile (<= 256; rows past it are not fetched: a short tile is short in HBM time)
    uint32_t* ticket;         // per-workspace counter, zero at launch (the merge kernel puts it back)
    int nq, k, cap, keep_max;
    int buf_cap;              // entries of the LDS candidate buffer
    int flush_at;             // buffer fill at which a flush into the global lists is requested
    float pmax2_hint;
    unsigned long long* dbg;  // tuning build only (atlas_tune_set_scan_stamps): 8 wall-clock stamps (100 MHz, common to all XCDs) per workgroup; null in production
    // PAIRED PASSES (grid.y == 2, round 3): two query chunks of a batch are scanned CONCURRENTLY, each by half of the chip (grid.x = CUs / 2
    // workgroups per chunk, the same row ranges in both halves). The second reader of a slab row is served by the Infinity Cache / L2
    // instead of HBM: the two half-chip scans deliver ~7.6 TB/s to the CUs for one slab of HBM reads, x 1.17-1.24 against the two passes
    // one after the other (profiles/r03/concurrent_chunks_experiment.txt). Chunk 1 = queries [q0 + nq, q0 + nq + nq2); its per-call state
    // lives pair_state bytes behind chunk 0's, its lists pair_bulk bytes behind (same layout; atlas_hip.hip: make_plan).
    int nq2;
    size_t pair_state, pair_bulk;
#if ATLAS_TUNING
    // EXPERIMENT, tuning build only (atlas_tune_set_scan_fused; tools/fused_timeline.py): the merge inside the scan -- the LAST nq workgroups
    // to finish their ranges stay and run the merge of one query each once every workgroup has handed over. Measured and NOT adopted:
    // the kernel boundary it removes costs 2.3 us (last hand-over -> first merge instruction), the in-kernel hand-off that replaces it
    // (L2 write-back, returning arrival atomic, published tag, poll, invalidate) 6.5-10 us: profiles/r03/fused_timeline.txt
    int fused;                // 0: the caller launches merge_rescore_kernel behind the scan
    uint32_t* fuse;           // per-workspace words: [0] arrivals (zero between calls) [1] tag of the last call in which a workgroup gave its query up
                              // [2] tag of the last call whose hand-over is complete; [64 + q] state of query q (FUSE_*, FREE between calls)
    MergeParams mp;
#endif
};
enum : uint32_t { FUSE_FREE = 0u, FUSE_ABANDONED = 1u, FUSE_MERGING = 2u };
#define ATLAS_FUSE_WAIT_TICKS 10000ull   // 100 us: how long a finished workgroup waits for the slowest one before it leaves its query to it

template <int NQ>
struct ScanSmemT {  // byte offsets into dynamic LDS; NQ = queries per slab pass (64, or 96 for the second half of big batches)
    static constexpr int q_off = 0;                       // the query image (fill_query_image): NQ rows of 1568 bytes
    static constexpr int theta_off = NQ * QROW_U4 * 16;   // NQ f32
    static constexpr int cnt_off = theta_off + NQ * 4;    // NQ u32
    static constexpr int flag_off = cnt_off + NQ * 4;     // 64 B: [0],[1] flush/compaction request by tile parity, [2] buffer fill, [8..15] tile tickets
    static constexpr int aux_off = flag_off + 64;         // [0, 256) per-wave norm maxima (final hand-over), [256, 256 + 4 NQ) per-query eps
    static constexpr int buf_off = aux_off + 256 + NQ * 4 + (NQ == 64 ? 256 : 0);   // buf_cap x {u32 score bits, u32 (query << QSHIFT) | row}   (64: the layout of rounds 1-2)
};
typedef ScanSmemT<64> ScanSmem;

// AUX & 31 = cache-policy bits of the slab loads (0 = default, 2 = nt: rows are read once by one CU)
// AUX & 64 = the caller's pmax is certified (ATLAS_SCAN_TRUST_PMAX): the row norms are not measured (one Gram MFMA per k-step less -- see
//            `gram` below; as 4 v_dot2 per k-step, rounds 2-3, the measurement was 4.6-7.5 % of the kernel's time, now ~3 %)
// NQF = 16-query fragments per slab k-step: 4 (64 queries per pass: every search of up to 64 queries) or 6 (96 queries per pass, round 3:
//       batches above 64 queries -- a rank of an N-GPU search scores ALL gathered queries -- read the slab once per 96 instead of once
//       per 64. The stream is 10 % slower under 6 MFMAs + 6 LDS reads per 16-byte load -- the kernel sits at the board's power limit --
//       for 50 % more queries per byte: x 1.36, profiles/r03/scan_96_query_proxy.txt. The 147 KiB image leaves 11.5 KiB of candidate
//       buffer (1 480 entries instead of 7 680: a flush per ~1 100 candidates), and candidate entries carry a 7-bit query + 25-bit row)
template <int NW, int PF, int RING, int AUX = 0, int NQF = 4>
__global__ void __launch_bounds__(NW * 64)
scan_kernel(const ScanParams pk) {
    ScanParams p = pk;
    if (blockIdx.y != 0) {                                 // the second chunk of a paired pass (uniform)
        auto st = [&](auto*& ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)((unsigned char*)ptr + pk.pair_state); };
        auto bk = [&](auto*& ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)((unsigned char*)ptr + pk.pair_bulk); };
        st(p.theta_gran); st(p.gran_max); st(p.epoch); st(p.qflag); st(p.ticket);
        bk(p.lists); bk(p.list_cnt); bk(p.wg_stat);
        p.q0 = pk.q0 + pk.nq;
        p.nq = pk.nq2;
    }
    constexpr int NQ = 16 * NQF;                           // queries per pass
    constexpr int QSHIFT = NQ > 64 ? 25 : 26;              // candidate entry: (query << QSHIFT) | virtual row
    constexpr uint32_t ROWMASK = (1u << QSHIFT) - 1u;
    typedef ScanSmemT<NQ> ScanSmem;
    // RING slots of PF fragments: RING-1 k-steps of loads in flight while one slot is consumed
    static_assert(KSTEPS % RING == 0 && RING >= 2, "prefetch ring must divide the k-steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_q = (uint4*)(smem + ScanSmem::q_off);
    float* s_theta = (float*)(smem + ScanSmem::theta_off);
    uint32_t* s_cnt = (uint32_t*)(smem + ScanSmem::cnt_off);
    uint32_t* s_flag = (uint32_t*)(smem + ScanSmem::flag_off);   // plain LDS words; ordered by wg_barrier_lds()
    uint2* s_buf = (uint2*)(smem + ScanSmem::buf_off);

    const int tid = threadIdx.x;
    ATLAS_SCAN_STAMP(0);        // [0] entry
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lgrp = lane >> 4;
    constexpr int TILE = NW * PF * 16;       // rows per workgroup tile
    constexpr int ROWB = D_FAST * 2;         // bytes per slab row
    constexpr int RPT = KSTEPS / RING;       // ring revolutions per tile

    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    int64_t r_end = r_begin + p.rows_per_wg;
    if (r_end > p.N) r_end = p.N;
    const int ntiles = (r_end > r_begin) ? (int)((r_end - r_begin + TILE - 1) / TILE) : 0;   // workgroup-uniform
    // lists are [query][workgroup][cap]: everything that belongs to one query -- what its merge block gathers -- lies within G * cap
    // entries (a few 2 MiB pages), and all workgroups' stores go to the same 64 regions
    const size_t qstride = (size_t)gridDim.x * p.cap;
    uint2* my_lists = p.lists + (size_t)blockIdx.x * p.cap;

    // Passage rows stream HBM -> VGPR through buffer loads (cdna guide T8). ONE descriptor per
    // wave spans [first row of this wave's first tile, N). The hardware bounds check covers
    // voffset + immediate only (not soffset), so everything that selects a ROW lives in the
    // per-lane voffset (one VGPR per fragment, bumped once per tile) and rows at or past the end of
    // the range read as zero; the k-step (< one row) rides in the scalar offset. No address VALU in the k-loop.
    //   lane l loads row (l & 15) of fragment pf, bytes [64*s + 16*(l>>4), +16)   (MFMA A operand)
    // The descriptor ends at the END OF THIS WORKGROUP'S RANGE, not at N: a workgroup runs whole tiles, so the waves of its last
    // tile that have no rows left, and the fill cursor running ahead of the last tile, would otherwise read the NEXT workgroup's
    // rows for real -- bytes that count against HBM and are thrown away (PMC: 4.8 % at 1M rows, 1.6 % at 4M). Out-of-range
    // buffer loads return zero without a memory request.
    const int64_t wrow0 = r_begin + (int64_t)wave * PF * 16;
    int64_t span = (wrow0 < r_end) ? (r_end - wrow0) * (int64_t)ROWB : 0;
    if (span > 0xfffffff0ll) span = 0xfffffff0ll;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)p.slab + (span > 0 ? wrow0 : 0) * (int64_t)ROWB), 0, (int)span, 0x00020000);

    // fill cursor: (rows of the tile being fetched -> vo[], k-step -> fill_step); it runs RING-1
    // steps ahead of the consumer. Past the last tile it keeps walking forward: those loads fall
    // out of the descriptor's bounds (return 0, no memory traffic).
    unsigned vo[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) vo[pf] = (unsigned)((pf * 16 + lrow) * ROWB + lgrp * 16);
    int fill_step = 0;
    // The tile sequence of a workgroup: its static tiles 0 .. ntiles-1, then pool tiles. The fill cursor enters tile f_seq + 1 seven
    // k-steps before the consumer leaves tile f_seq, so it is the fill cursor that finds out where the next tile is:
    //   static tile          the next TILE rows of the wave's descriptor;
    //   first pool tile      pool tile blockIdx.x (pre-assigned);
    //   every further one    G + the ticket that wave NW-1 drew at the START of the tile before (a plain returning atomic) and posted
    //                        in LDS as {sequence number, ticket}; readers spin on the LDS word (lgkmcnt only -- the ring of slab
    //                        loads is not disturbed), which in practice is there 10 us earlier.
    // Pool tiles have POOL_TILE = (NW-1) * PF * 16 rows: wave NW-1 takes none. A returning atomic waits behind everything its CU has
    // in flight (~5 us under a full ring) and the compiler drains the wave's own ring for it; with rows of its own the ticket wave
    // reached every tile barrier last (measured: +4 us per pool tile). Its loads fall out of bounds instead (zeros, no traffic).
    // A ticket past the last pool tile ends the sequence: the remaining fills fall out of bounds as well.
    constexpr int POOL_TILE = (NW - 1) * PF * 16;
    // (an LDS-address-space pointer: through a generic pointer the poll below would be a FLAT load, which counts in vmcnt and would
    //  drain the ring of slab loads once per tile)
    typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;
    lds_vu64* s_tk = (lds_vu64*)(smem + ScanSmem::flag_off + 32);   // [4] ring of posted {sequence number << 32 | ticket}
    const uint32_t vpool = (uint32_t)p.rows_per_wg;        // virtual row index of pool row 0 in this workgroup's candidate entries
    int f_seq = 0;                  // tile the fill cursor is in
    uint32_t n_pt = 0;              // pool tile the fill cursor last entered
    bool n_ok = false;              //   ... and whether it exists
    auto fill_next_tile = [&]() {   // scalar code, once per tile
        fill_step = 0;
        ++f_seq;
        if (f_seq < ntiles) {
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) vo[pf] += (unsigned)(TILE * ROWB);
            return;
        }
        uint32_t pt = blockIdx.x;
        if (f_seq > ntiles) {
            unsigned long long e = s_tk[f_seq & 3];
            while ((uint32_t)(e >> 32) != (uint32_t)f_seq) { __builtin_amdgcn_s_sleep(1); e = s_tk[f_seq & 3]; }
            pt = gridDim.x + (uint32_t)e;
        }
        pt = __builtin_amdgcn_readfirstlane(pt);
        n_pt = pt;
        n_ok = pt < (uint32_t)p.pool_tiles;
        const int64_t pool_bytes = (int64_t)p.pool_rows * ROWB;                  // < 2^32 - 16 (host plan)
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.slab + p.pool_begin * (int64_t)ROWB), 0, (int)pool_bytes, 0x00020000);
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
            vo[pf] = (n_ok && wave < NW - 1) ? (unsigned)(((pt * (uint32_t)POOL_TILE + (uint32_t)(wave * PF * 16 + pf * 16 + lrow)) * (uint32_t)ROWB) + (uint32_t)lgrp * 16u)
                                             : 0xfffffff0u;
    };
    auto fill_advance = [&](const bool may_wrap) {   // the cursor wraps at ring slot 0 only: KSTEPS % RING == 0 and it runs RING-1 steps ahead
        ++fill_step;
        if (may_wrap && fill_step == KSTEPS) fill_next_tile();
    };

    u32x4 abuf[RING][PF];
    auto ring_prologue = [&]() {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) {
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
                abuf[s][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, AUX & 31);
            fill_advance(false);
            // keep issue order == ring order: hipcc's waitcnt for slot 0 is the minimum over the loop
            // entry and the back edge, so a shuffled prologue would cost ring depth on every revolution
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // The query image is built in LDS straight from the caller's tensor, and its loads go out FIRST; the ring's first slab loads follow
    // them (the image loader's hook below) and are in flight while the image is converted. A wave's loads return in order: with the ring
    // first (rounds 1-2) the image sat behind 29 MB of cold slab requests of all 256 workgroups, and nothing streamed while it was
    // converted: image in LDS 12.5 us after entry. Now 11.4 us (1M-row step -1.8 us, profiles/r03/scan_startup.txt): what is left is the
    // CU's own path -- 196 KiB of fp32 queries + 112 KiB of ring per CU at the ~23 GB/s a CU gets when all 256 ask at once (the queries
    // do not survive in L2 from call to call: a scan streams gigabytes through it); reading the image in a rotated order per CU (L2
    // channel spread) measured no different. Then every wave derives eps and the initial threshold of its share of the queries.
    // Initial thresholds. Deriving one (the k-th largest of 512 sample maxima, a bit search with ballots) costs ~1 us for one wave --
    // but 64 of them on every CU were 30 us of start-up (4 waves share a SIMD). So the last wave of workgroup q derives the threshold of
    // query q alone -- its inputs (the query's row, for eps, and the sample maxima) are requested before the image loads and it works
    // on them while those land -- and publishes it as one 8-byte {tag, value} granule (one write-through store); the last wave of every
    // workgroup then collects the 64 granules. Workgroups 0..63 are dispatched first; the wait is bounded, and a threshold that has not
    // arrived in time simply starts at -inf (slower, never wrong).
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* gran = (gu64*)p.theta_gran;
    const bool exchange = p.top2 != nullptr && !p.coop;
    const uint32_t tag = p.coop ? *p.epoch + 1u : 1u;              // (uniform: a scalar load)
    if (p.coop && blockIdx.x == 0 && p.q0 == 0 && tid < ATLAS_STATUS_HEADER) p.out_status[tid] = 0;   // the merge accumulates into it
    const bool theta_wave = exchange && wave == NW - 1 && (int)blockIdx.x < p.nq;       // wave-uniform
    float tv[4];
    QRaw trow[2];                                       // the 768 elements of query blockIdx.x: 12 per lane = two 8-element pieces, 4 unused
    if (theta_wave) {
        load_sample_maxima(p.top2 + (size_t)blockIdx.x * 2 * p.sample_blocks, p.sample_blocks, lane, tv);
        const size_t base = (size_t)(p.q0 + (int)blockIdx.x) * D_FAST;
        trow[0] = load_q8(p.q, p.q_dtype, base + (size_t)lane * 8);                         // chunks 0..63
        trow[1] = load_q8(p.q, p.q_dtype, base + (size_t)(64 + (lane & 31)) * 8);           // chunks 64..95 (lanes >= 32 repeat them)
    }
    auto image_hook = [&]() {
        // EVERY wave's image loads are queued before ANY wave's slab loads (a raw barrier: nothing is waited for). The CU's L1 serves its
        // queue in order and holds a bounded number of misses: slab loads (cold HBM, all 256 workgroups at once) queued in front of
        // another wave's image loads (L2 hits) kept the image barrier waiting until ~10 us after entry (tools stamps, round 3)
        __builtin_amdgcn_s_barrier();
        ring_prologue();
        ATLAS_SCAN_STAMP(1);    // [1] image loads and ring prologue issued
        if (!theta_wave) return;
        float ss = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint4 v = q8_to_f16(trow[h], p.q_dtype);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            if (h == 0 || lane < 32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const f16x2 x = __builtin_bit_cast(f16x2, w[i]); ss = __builtin_amdgcn_fdot2(x, x, ss, false); }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float th = initial_theta(tv, p.sample_blocks, p.k, query_eps(ss, p.pmax), lane);
        if (lane == 0) __hip_atomic_store(gran + blockIdx.x, (1ull << 32) | (unsigned long long)f32_bits(th), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ATLAS_SCAN_STAMP_LANE0(6);   // [6] threshold of query blockIdx.x published
    };
    fill_query_image<NW * 64, decltype(image_hook), NQ>(s_q, p.q, p.q_dtype, p.q0, p.nq, tid, image_hook);
    if (tid < NQ) s_cnt[tid] = 0;
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; s_flag[2] = 0; }
    if (tid < 4) s_tk[tid] = 0ull;      // sequence number 0 is never asked for (LDS keeps the previous kernel's words)
    __syncthreads();
    float* s_eps = (float*)(smem + ScanSmem::aux_off + 256);
    if (wave == NW - 1) ATLAS_SCAN_STAMP_LANE0(1);   // [1] (re-stamped) image barrier passed
    for (int qq = wave; qq < NQ; qq += NW) {
        const float eps = qq < p.nq ? query_eps(image_row_sumsq(s_q, qq, lane), p.pmax) : 0.f;
        if (lane == 0) s_eps[qq] = eps;
    }
    if (exchange) {
        if (wave == NW - 1) {
            unsigned long long g = 0ull;
            const bool want = lane < p.nq;
            for (const unsigned long long spin_end = wall_clock64() + ATLAS_SPIN_TICKS; ; ) {
                if (want && (g >> 32) == 0ull) g = __hip_atomic_load(gran + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(want && (g >> 32) == 0ull) == 0ull) break;
                if (wall_clock64() >= spin_end) break;
                __builtin_amdgcn_s_sleep(2);
            }
            s_theta[lane] = want ? ((g >> 32) != 0ull ? bits_f32((uint32_t)g) : neg_inf()) : pos_inf();
            ATLAS_SCAN_STAMP_LANE0(7);   // [7] all thresholds collected
        }
    } else if (tid < NQ) {
        s_theta[tid] = (tid < p.nq) ? neg_inf() : pos_inf();      // query slots beyond nq never collect anything
    }
    __syncthreads();
    ATLAS_SCAN_STAMP(2);        // [2] query image, eps and thresholds in LDS

    f32x4 acc[PF][NQF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int qf = 0; qf < NQF; ++qf) acc[pf][qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // row norms of the certifying twin: the GRAM matrix of the wave's 16 rows, one more MFMA per k-step with the slab fragment as BOTH
    // operands (A[i][k] and B[k][j] of v_mfma_16x16x32 have the same register layout: D = A A^T) -- its diagonal is the rows' sums of
    // squares, accumulated in fp32 like everything else. It replaces 4 dependent v_dot2 per k-step, which cost the kernel 7.5 % (0.724 vs
    // 0.782 of the HBM peak at 32M rows) although the matrix pipe is only ~20 % busy: it is the wave's instruction stream that is full.
    f32x4 gram[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) gram[pf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pm = 0.0f;   // running max of row sum-of-squares seen by this lane's row group

    // rows relative to r_begin fit 32 bits (plan guarantees rows_per_wg * 1536 < 2^32)
    // Candidate entries carry a 26-bit VIRTUAL row: [0, rows_per_wg) = the static range, from vpool on = pool rows (host plan keeps
    // vpool + pool_rows below 2^26); global_row() turns it into the shard-local row when entries leave for the global lists
    int nrows = (int)(r_end > r_begin ? r_end - r_begin : 0);      // end of the valid virtual rows of the current tile's region
    const uint32_t gbase = (uint32_t)r_begin;          // shard-local row ids are < 2^32
    const uint32_t pbase = (uint32_t)p.pool_begin - vpool;
    auto global_row = [&](const uint32_t v) -> uint32_t { return v + (v >= vpool ? pbase : gbase); };
    int row0 = wave * PF * 16;   // first (virtual) row of this wave's current tile
    int c_seq = 0;               // tile the consumer is in
    int par = 0;               // tile parity (double-buffers the compaction-request flag)
    int cstep = 0;             // consumer k-step inside the tile

    // Flush the LDS buffer into the per-query global lists, then compact every list that crossed
    // keep_max (k-th largest -> certified threshold -> in-place prune). Called by ALL waves between
    // two barriers, rarely: buffer half full, direct-store overflow, or end of the scan.
    auto flush_and_compact = [&](const int parity, const bool final_flush) {
        wg_barrier_lds();                       // everybody has read the request words
        const uint32_t nbuf = s_flag[2] < (uint32_t)p.buf_cap ? s_flag[2] : (uint32_t)p.buf_cap;
        for (uint32_t i = tid; i < nbuf; i += NW * 64) {
            const uint2 e = s_buf[i];
            const uint32_t qq = e.y >> QSHIFT;
            const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
            if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(e.x, global_row(e.y & ROWMASK));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);     // list stores complete before anyone reads them back
        wg_barrier_lds();
        if (tid == 0) { s_flag[parity] = 0u; s_flag[2] = 0u; }
        for (int qq = wave; qq < p.nq; qq += NW) {
            const uint32_t n = s_cnt[qq] < (uint32_t)p.cap ? s_cnt[qq] : (uint32_t)p.cap;
            // mid-scan: tighten every list that can be pruned; at the end only the oversized ones
            if (n <= (uint32_t)(final_flush ? p.keep_max : p.k)) continue;
            uint2* L = my_lists + (size_t)qq * qstride;
            // k-th largest approximate score: greedy bit search below the common prefix of the keys,
            // stopping 2^-15 short of exact (any lower bound of the k-th is a valid T)
            uint32_t kmax = 0, kmin = 0xffffffffu;
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t key = f32_order_key(bits_f32(L[i].x));
                kmax = key > kmax ? key : kmax;
                kmin = key < kmin ? key : kmin;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t a = __shfl_xor(kmax, o), b = __shfl_xor(kmin, o);
                kmax = a > kmax ? a : kmax;
                kmin = b < kmin ? b : kmin;
            }
            const uint32_t diff = kmax ^ kmin;
            const int top = diff ? 31 - __builtin_clz(diff) : -1;
            uint32_t prefix = (top < 0) ? kmax : ((top >= 31) ? 0u : (kmax & ~((2u << top) - 1u)));
            const int stop = top - 22 > 0 ? top - 22 : 0;
            for (int bit = top; bit >= stop; --bit) {
                const uint32_t cand = prefix | (1u << bit);
                uint32_t c = 0;
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool ge = (i < n) && (f32_order_key(bits_f32(L[i].x)) >= cand);
                    c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ge));
                }
                if (c >= (uint32_t)p.k) prefix = cand;
            }
            const float theta = prune_threshold(f32_from_order_key(prefix), s_eps[qq]);
            // in-place stable compaction of entries with score > theta
            uint32_t kept = 0;
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                uint2 e = make_uint2(0, 0);
                bool keep = false;
                if (i < n) { e = L[i]; keep = bits_f32(e.x) > theta; }
                const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
                const uint32_t pos = kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (keep) L[pos] = e;
                kept += (uint32_t)__popcll(m);
            }
            if (lane == 0) {
                if (kept > (uint32_t)p.keep_max || s_cnt[qq] > (uint32_t)p.cap) {
                    // candidate band wider than the list (mass ties): hand this query to the
                    // exact path and stop collecting for it
                    p.qflag[qq] = 1u;
                    s_cnt[qq] = 0;
                    s_theta[qq] = pos_inf();
                } else {
                    s_cnt[qq] = kept;
                    s_theta[qq] = theta;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        wg_barrier_lds();
    };

    // coop mode -- the sample is the scan's own first tiles (256 workgroups x 256 rows = 65 536 evenly spread rows, no extra bytes, no
    // extra kernel). At the end of its first tile a workgroup publishes its best score per query (64 granules {tag, value}); workgroup q
    // collects the G maxima of query q, takes their k-th largest (scores of distinct rows -> a certified threshold) and publishes it; every
    // workgroup collects the 64 thresholds, then filters the tile it is still holding in registers. Two hops of ~3 us. Every wait is
    // bounded: a maximum or threshold that has not arrived counts as -inf (slower, never wrong), so nothing depends on co-residency.
    auto first_tile_exchange = [&](const float (&tm)[NQF]) {
        float* s_tmax = (float*)s_buf;                     // [NW][64] scratch: the candidate buffer is still empty
        gu64* gmax = (gu64*)p.gran_max;
        const uint32_t G = gridDim.x;
        if (lgrp == 0) {
#pragma unroll
            for (int qf = 0; qf < NQF; ++qf) s_tmax[wave * NQ + qf * 16 + lrow] = tm[qf];
        }
        wg_barrier_lds();
        if (tid < p.nq) {
            float m = s_tmax[tid];
            for (int w = 1; w < NW; ++w) m = fmaxf(m, s_tmax[w * NQ + tid]);
            uint32_t to = (uint32_t)tid;                   // the granule's address is formed here, not hoisted over the slab loop into scratch
            asm volatile("" : "+v"(to));
            __hip_atomic_store(gmax + (size_t)to * G + blockIdx.x, ((unsigned long long)tag << 32) | (unsigned long long)f32_bits(m),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == NW - 1) {
            // (granule addresses are formed HERE, from a lane index hipcc cannot trace back: hoisted over the slab loop they end up in scratch)
            uint32_t lane_o = (uint32_t)lane;
            asm volatile("" : "+v"(lane_o));
            if ((int)blockIdx.x < p.nq) {                  // this workgroup derives the threshold of query blockIdx.x
                const int qq = blockIdx.x;
                unsigned long long g[4] = {0ull, 0ull, 0ull, 0ull};
                for (const unsigned long long spin_end = wall_clock64() + ATLAS_SPIN_TICKS; ; ) {
                    bool missing = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = lane_o + 64u * u;
                        if (i < G && (uint32_t)(g[u] >> 32) != tag) {
                            g[u] = __hip_atomic_load(gmax + (size_t)qq * G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            missing |= (uint32_t)(g[u] >> 32) != tag;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                    if (wall_clock64() >= spin_end) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ((uint32_t)(g[u] >> 32) == tag) ? bits_f32((uint32_t)g[u]) : neg_inf();
                const float th = initial_theta(v, (int)G, p.k, s_eps[qq], lane);
                if (lane == 0) __hip_atomic_store(gran + qq, ((unsigned long long)tag << 32) | (unsigned long long)f32_bits(th), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ATLAS_SCAN_STAMP_LANE0(6);   // [6] threshold of query blockIdx.x published
            }
            constexpr int QR = (NQ + 63) / 64;             // rounds of 64 queries
            unsigned long long g[QR];
#pragma unroll
            for (int u = 0; u < QR; ++u) g[u] = 0ull;
            for (const unsigned long long spin_end = wall_clock64() + ATLAS_SPIN_TICKS; ; ) {
                bool missing = false;
#pragma unroll
                for (int u = 0; u < QR; ++u) {
                    const int qi = (int)lane_o + 64 * u;
                    if (qi < p.nq && (uint32_t)(g[u] >> 32) != tag) {
                        g[u] = __hip_atomic_load(gran + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        missing |= (uint32_t)(g[u] >> 32) != tag;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                if (wall_clock64() >= spin_end) break;
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int u = 0; u < QR; ++u) {
                const int qi = lane + 64 * u;
                if (qi < NQ) s_theta[qi] = qi < p.nq ? ((uint32_t)(g[u] >> 32) == tag ? bits_f32((uint32_t)g[u]) : neg_inf()) : pos_inf();
            }
            ATLAS_SCAN_STAMP_LANE0(7);   // [7] all thresholds collected
        }
        wg_barrier_lds();
    };
    if (p.coop && ntiles == 0) {               // a workgroup without rows still owes the others its (empty) maxima
        float none[NQF];
#pragma unroll
        for (int qf = 0; qf < NQF; ++qf) none[qf] = neg_inf();
        first_tile_exchange(none);
    }

    // One flat loop over ring revolutions of all tiles: the ring rotation is the same every
    // iteration (no register shuffling at tile boundaries), and the per-tile work (filter,
    // barrier) hangs off every RPT-th revolution.
    if (ntiles > 0)
#pragma unroll 1
    for (;;) {
        // B operands of this revolution: query rows (16 qf + lrow), 16 bytes at k-step * 64 + 16 * lgrp (two bases: the offset of
        // query group 3 does not fit the 16-bit immediate of ds_read_b128)
        const uint4* bq0 = s_q + lrow * QROW_U4 + lgrp + cstep * 4;
        const uint4* bq2 = bq0 + 32 * QROW_U4;
        const uint4* bq4 = bq0 + 64 * QROW_U4;             // (NQF == 6)
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            // refill the slot freed by the previous step first (its loads stay in flight for
            // RING-1 steps), then consume slot j. sched_barrier pins that order: left alone,
            // hipcc sinks the loads to the loop end and waits vmcnt(0) at the top.
            const int fill = (j + RING - 1) % RING;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
                abuf[fill][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, AUX & 31);
            fill_advance(j == 0);
            __builtin_amdgcn_sched_barrier(0);
            uint4 b[NQF];
#pragma unroll
            for (int qf = 0; qf < NQF; ++qf) b[qf] = (qf < 2 ? bq0 : qf < 4 ? bq2 : bq4)[(qf & 1) * 16 * QROW_U4 + j * 4];
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const u32x4 a = abuf[j][pf];
                const f16x8 av = __builtin_bit_cast(f16x8, a);
#pragma unroll
                for (int qf = 0; qf < NQF; ++qf)
                    acc[pf][qf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        av, __builtin_bit_cast(f16x8, b[qf]), acc[pf][qf], 0, 0, 0);
                // row sums of squares (certifies pmax_hint): the diagonal of the fragment's Gram matrix
                if constexpr (!(AUX & 64)) gram[pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, av, gram[pf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cstep += RING;
        if (cstep < KSTEPS) continue;

        // ------------------------- end of a tile: filter --------------------------------
        cstep = 0;
        const bool have_rows = row0 < nrows;        // wave-uniform
        if (have_rows) {
            // full row norms: lane (column lrow, rows 4 lgrp + r) of the Gram matrix holds the diagonal element of row lrow iff lgrp == lrow >> 2
            if constexpr (!(AUX & 64)) {
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) {
                    const f32x4 gm = gram[pf];
                    const int r = lrow & 3;
                    const float x = r == 0 ? gm[0] : r == 1 ? gm[1] : r == 2 ? gm[2] : gm[3];
                    pm = fmaxf(pm, lgrp == (lrow >> 2) ? x : 0.f);
                }
            }
        }
        // rows past the end of this workgroup's range never become candidates (a per-row flag that the maxima and the filter look at: writing
        // -inf into the accumulators of a partial tile made hipcc keep two copies of them -- with 24 accumulators, in scratch)
        bool vr[PF][4];
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int r = 0; r < 4; ++r) vr[pf][r] = row0 + pf * 16 + lgrp * 4 + r < nrows;
        if (p.coop && c_seq == 0) {            // workgroup-uniform: the end of every wave's FIRST tile
            float tm[NQF];
#pragma unroll
            for (int qf = 0; qf < NQF; ++qf) {
                float m = neg_inf();
                if (have_rows) {
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) m = fmaxf(m, vr[pf][r] ? acc[pf][qf][r] : neg_inf());
                }
                m = fmaxf(m, __shfl_xor(m, 16));
                tm[qf] = fmaxf(m, __shfl_xor(m, 32));
            }
            first_tile_exchange(tm);
        }
        if (have_rows) {
            // threshold filter: lane l owns query 16*qf + (l&15) in acc[.][qf]
            float th[NQF];
#pragma unroll
            for (int qf = 0; qf < NQF; ++qf) th[qf] = s_theta[qf * 16 + lrow];
            bool any = false;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                for (int qf = 0; qf < NQF; ++qf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) any |= vr[pf][r] && acc[pf][qf][r] > th[qf];
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                // Candidates go to a workgroup-wide LDS buffer: LDS traffic is counted in lgkmcnt, so the
                // ring of HBM loads (vmcnt) is not disturbed. (gfx9 counts loads and stores in one vmcnt
                // and retires them out of order with respect to each other: a single pending global store
                // would force vmcnt(0) in front of every ring slot.) Only when the buffer is full do entries
                // go straight to the global lists, and that branch drains itself.
                const uint32_t rrel = (uint32_t)row0 + (uint32_t)lgrp * 4u;
                // (everything below that depends on the lane's query is formed HERE, from a value hipcc cannot trace back to the lane id:
                //  otherwise the four per-lane list pointers and the shifted query tags are hoisted out of the slab loop and, at the
                //  128-register cap, kept in scratch -- and a kernel with a private segment pays ~12 us per launch for it)
                uint32_t lrow_o = (uint32_t)lrow;
                asm volatile("" : "+v"(lrow_o));
                bool spilled = false;
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                    for (int qf = 0; qf < NQF; ++qf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[pf][qf][r];
                            if (vr[pf][r] && v > th[qf]) {
                                const uint32_t qq = (uint32_t)(qf * 16) + lrow_o;
                                const uint32_t slot = atomicAdd(&s_flag[2], 1u);
                                if (slot < (uint32_t)p.buf_cap) {
                                    s_buf[slot] = make_uint2(f32_bits(v), (qq << QSHIFT) | (rrel + (uint32_t)(pf * 16 + r)));
                                    if (slot >= (uint32_t)p.flush_at) s_flag[par] = 1u;   // request a flush
                                } else {
                                    const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
                                    if (gs < (uint32_t)p.cap)
                                        my_lists[qq * qstride + gs] =
                                            make_uint2(f32_bits(v), global_row(rrel + (uint32_t)(pf * 16 + r)));
                                    s_flag[par] = 1u;
                                    spilled = true;
                                }
                            }
                        }
                if (__builtin_amdgcn_ballot_w64(spilled) != 0ull)
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): drain the direct stores (builtin: visible to the waitcnt pass)
            }
        }
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            gram[pf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qf = 0; qf < NQF; ++qf) acc[pf][qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        wg_barrier_lds();
        if (c_seq == 0) ATLAS_SCAN_STAMP(3);   // [3] first tile done
#if ATLAS_TUNING
        // per-tile end stamps of this workgroup, collected in LDS (no global store in the loop) and dumped after the hand-over
        if (p.dbg && tid == 0 && c_seq < 120) ((unsigned long long*)(smem + ScanSmem::buf_off + (size_t)p.buf_cap * 8))[c_seq] = wall_clock64();
#endif
        // the request word of this tile's parity cannot change until every wave has passed the next
        // barrier, so all waves take the same branch
        if (s_flag[par] != 0u) flush_and_compact(par, false);
        par ^= 1;
        // on to tile c_seq + 1 (the fill cursor entered it seven k-steps ago and knows where it is)
        ++c_seq;
        if (c_seq < ntiles) row0 += TILE;
        else {
            if (!n_ok) break;                                   // workgroup-uniform: every wave read the same ticket
            nrows = (int)vpool + p.pool_rows;
            row0 = (wave < NW - 1) ? (int)(vpool + n_pt * (uint32_t)POOL_TILE) + wave * PF * 16 : nrows;     // the ticket wave has no rows
            // the ticket of the tile after this one, drawn now (see fill_next_tile)
            if (wave == NW - 1 && lane == 0) {
                const uint32_t t = atomicAdd(p.ticket, 1u);
                s_tk[(c_seq + 1) & 3] = ((unsigned long long)(uint32_t)(c_seq + 1) << 32) | (unsigned long long)t;
            }
        }
    }

    ATLAS_SCAN_STAMP(4);        // [4] last tile done
    // Final hand-over, WITHOUT global atomics: the buffered candidates join this workgroup's own per-query lists (slots from the LDS
    // counters), the 64 list lengths and the workgroup's norm / flag word go out as plain stores, and the merge kernel gathers the G
    // segments of its query. (It used to reserve ranges of one shared per-query array with 64 atomicAdds per workgroup, plus one
    // atomicMax per wave: 20k atomics on three cache lines as the workgroups finish. The loads of every workgroup still scanning
    // queued behind them at that L2 channel: its last tiles took 25-45 us instead of 7-17, ~30 us of every scan whatever its size --
    // profiles/r02/scan_wg_times_wallclock.txt, scan_tail_modes.txt.)
    {
        wg_barrier_lds();
        const uint32_t nbuf = s_flag[2] < (uint32_t)p.buf_cap ? s_flag[2] : (uint32_t)p.buf_cap;
        for (uint32_t i = tid; i < nbuf; i += NW * 64) {
            const uint2 e = s_buf[i];
            const uint32_t qq = e.y >> QSHIFT;
            const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
            if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(e.x, global_row(e.y & ROWMASK));
        }
        // largest row norm^2 of the workgroup (x1.001: the Gram MFMA accumulates in fp32): waves -> LDS -> one word
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
        float* s_pm = (float*)(smem + ScanSmem::aux_off);
        if (lane == 0) s_pm[wave] = pm;
        wg_barrier_lds();
        if (tid < NQ) {
            uint32_t c = (tid < p.nq) ? s_cnt[tid] : 0u;
            if (c > (uint32_t)p.cap) { p.qflag[tid] = 1u; c = 0u; }          // a list overflowed -> exact path
            p.list_cnt[(size_t)tid * gridDim.x + blockIdx.x] = c;
        }
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < NW; ++w) m = fmaxf(m, s_pm[w]);
            m *= 1.001f;
            p.wg_stat[(size_t)blockIdx.x * 2 + 0] = f32_bits(m);
            p.wg_stat[(size_t)blockIdx.x * 2 + 1] = (m > p.pmax2_hint) ? (uint32_t)ATLAS_F_PMAX_VIOLATION : 0u;
        }
    }
    ATLAS_SCAN_STAMP(5);        // [5] hand-over done
#if ATLAS_TUNING
    if (p.dbg && tid < 120) p.dbg[2048 + blockIdx.x * 120 + tid] = (tid < c_seq) ? ((unsigned long long*)(smem + ScanSmem::buf_off + (size_t)p.buf_cap * 8))[tid] : 0ull;
#endif
#if ATLAS_TUNING
    // Fused merge (experiment, see ScanParams). Arrival = one returning atomic per workgroup, behind a release fence that covers every list entry, length and
    // norm / flag word this workgroup wrote. Arrival number t decides: the first G - nq workgroups leave; workgroup t >= G - nq owns
    // query t - (G - nq) and waits until the last arriver has published the call's tag. The wait is BOUNDED (a workgroup that has not
    // even started -- CUs held by another stream's kernel -- can be a whole scan away): a workgroup that gives up marks its query
    // ABANDONED and leaves, and the last arriver, which by construction finds everything complete, merges the abandoned queries itself
    // after its own. abandon: CAS(FREE -> ABANDONED), note the call's tag in fuse[1], re-check fuse[2] (the last arriver may have looked
    // before the mark: then this workgroup takes its query back, CAS(ABANDONED -> MERGING)); last arriver: publish the tag in fuse[2],
    // THEN read fuse[1], THEN CAS(ABANDONED -> MERGING) per query: exactly one of the two merges every query. (Tags are unique per
    // call: nothing but the arrival counter and the states of merged queries has to be put back.)
    if constexpr (NW * 64 == 1024 && NQF == 4) {
        if (p.fused) {
            uint32_t* s_fz = (uint32_t*)(smem + ScanSmem::flag_off);       // [4], [5]: free from here on (the tile tickets used [8..15])
            const uint32_t G = gridDim.x, first = G - (uint32_t)p.nq;
            // release: every wave waits for ITS stores to reach the L2, then ONE thread writes the L2 back and arrives (a fence in every
            // thread is 16 L2 write-backs per workgroup, all of them in the last microseconds of the scan: +160 us per search, measured)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                s_fz[4] = atomicAdd(&p.fuse[0], 1u);
            }
            __syncthreads();
            const uint32_t t = s_fz[4];
            if (t < first || t >= G) return;                               // (>= G: a workspace whose state words were not zero before its first use)
            const bool last = t == G - 1u;
            const int myq = (int)(t - first);
            if (tid == 0) {
                uint32_t mine = 1u;
                if (last) {
                    p.fuse[0] = 0u;                                        // (nobody else arrives in this call)
                    __hip_atomic_store(&p.fuse[2], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const unsigned long long end = wall_clock64() + ATLAS_FUSE_WAIT_TICKS;
                    while (__hip_atomic_load(&p.fuse[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) {
                        if (wall_clock64() >= end) {
                            uint32_t expect = FUSE_FREE;
                            __hip_atomic_compare_exchange_strong(&p.fuse[64 + myq], &expect, (uint32_t)FUSE_ABANDONED, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_exchange(&p.fuse[1], tag, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                            if (__hip_atomic_load(&p.fuse[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == tag) {
                                expect = FUSE_ABANDONED;
                                mine = __hip_atomic_compare_exchange_strong(&p.fuse[64 + myq], &expect, (uint32_t)FUSE_MERGING, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
                            } else mine = 0u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (mine) __threadfence();                                 // acquire: the other workgroups' lists (one invalidate for the CU)
                s_fz[5] = mine;
            }
            __syncthreads();
            if (s_fz[5] == 0u) return;
            // (one copy of the merge in the kernel: the last arriver comes round again for the queries whose workgroups gave up)
            for (int qq = myq, scan_from = 0; ; ) {
                merge_rescore_body<1024>(p.mp, qq, smem);
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_store(&p.fuse[64 + qq], (uint32_t)FUSE_FREE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int next = -1;
                    if (last && __hip_atomic_load(&p.fuse[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == tag) {
                        for (int c = scan_from; c < p.nq && next < 0; ++c) {
                            uint32_t expect = FUSE_ABANDONED;
                            if (c != myq && __hip_atomic_compare_exchange_strong(&p.fuse[64 + c], &expect, (uint32_t)FUSE_MERGING, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                next = c;
                        }
                    }
                    s_fz[4] = (uint32_t)next;
                }
                __syncthreads();
                const int next = (int)s_fz[4];
                if (next < 0) return;
                qq = next;
                scan_from = next + 1;
                __syncthreads();
            }
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------
// sample pre-pass: approximate scores of S evenly spread rows -> the two best per 64-row tile and query,
// from which every scan workgroup derives a certified initial threshold per query (initial_theta). Without it
// every workgroup starts at -inf and pays a "cold start" (its first tile passes entirely).
//   grid = S/64 blocks of 256 threads; wave w of block j scores rows row(j) + 16w .. +16
// ------------------------------------------------------------------------------------------
struct SampleParams {
    const uint16_t* slab; int64_t N;
    const void* q; int q_dtype, q0, nq;   // the caller's queries: every block converts them into its own LDS image (fill_query_image)
    float* top2;              // [64][S/64][2]: the two best approximate scores of each 64-row sample tile
    int S;                    // multiple of 64
    int64_t stride_rows;      // first row of sample tile j = j * stride_rows (multiple of 16, >= 64)
    uint32_t* qflag;          // [64] per-query fallback flags of the scan that follows: cleared here (block 0)
    unsigned long long* theta_gran;   // [64] threshold granules of the scan that follows: cleared here (block 0)
    uint4* q16;               // [64][96] the queries of this pass as fp16 rows: block j < 64 writes row j of its image (the scan and the
                              // merge then read 2-byte queries whatever the caller's dtype)
    int32_t* out_status;      // status header: cleared by block 0 when this is the first 64-query pass of the call
};

// 512 threads: waves 0..3 score 16 rows each, all 8 waves convert the queries (the block's 192 KiB of fp32 queries in two batches of
// loads per thread instead of four: the conversion, not the 64 rows, is what this kernel's time is made of)
#define SAMPLE_NT 512
__global__ void __launch_bounds__(SAMPLE_NT)
sample_scores_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_q = (uint4*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool scorer = wave < 4;                 // wave-uniform
    const int lrow = lane & 15, lgrp = lane >> 4;
    // per-call state of the kernels that follow (the scan sets qflag, the merge accumulates into the status header)
    if (blockIdx.x == 0) {
        if (tid < QCHUNK) { p.qflag[tid] = 0u; p.theta_gran[tid] = 0ull; }
        if (p.q0 == 0 && tid < ATLAS_STATUS_HEADER) p.out_status[tid] = 0;
    }
    const int64_t row0 = (int64_t)blockIdx.x * p.stride_rows + (wave & 3) * 16;
    int64_t r = row0 + lrow;
    if (r >= p.N) r = p.N - 1;
    const uint4* src = (const uint4*)((const unsigned char*)p.slab + r * (int64_t)(D_FAST * 2) + lgrp * 16);
    // all 24 fragments of a scoring wave's 16 rows go out first (HBM latency overlaps the query conversion)
    uint4 a[KSTEPS];
    if (scorer) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) a[s] = src[s * 4];
    }
    fill_query_image<SAMPLE_NT>(s_q, p.q, p.q_dtype, p.q0, p.nq, tid);
    __syncthreads();
    if (blockIdx.x < QCHUNK && tid < D_FAST / 8) p.q16[blockIdx.x * (D_FAST / 8) + tid] = s_q[blockIdx.x * QROW_U4 + tid];
    f32x4 acc[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) acc[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (scorer) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const f16x8 av = __builtin_bit_cast(f16x8, a[s]);
#pragma unroll
            for (int qf = 0; qf < 4; ++qf)
                acc[qf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    av, __builtin_bit_cast(f16x8, s_q[(qf * 16 + lrow) * QROW_U4 + s * 4 + lgrp]), acc[qf], 0, 0, 0);
        }
    }
    // two best scores per query over this block's 64 rows: per lane (4 rows) -> across the 4 lane groups
    // that share a query column (xor 16, 32) -> across the 4 waves through LDS. Any subset of real
    // scores gives a certified threshold; with 256 tiles, missing a 3rd score of one tile is rare.
    auto merge2 = [](float& a1, float& a2, const float b1, const float b2) {
        const float hi = fmaxf(a1, b1), lo = fminf(a1, b1);
        a2 = fmaxf(lo, fmaxf(a2, b2));
        a1 = hi;
    };
    float* s_t2 = (float*)smem;                      // reuse the query image: [4 waves][64 queries][2]
    float t1[4], t2[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) {
        t1[qf] = neg_inf(); t2[qf] = neg_inf();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float v = (row0 + lgrp * 4 + rr < p.N) ? acc[qf][rr] : neg_inf();
            merge2(t1[qf], t2[qf], v, neg_inf());
        }
        merge2(t1[qf], t2[qf], __shfl_xor(t1[qf], 16), __shfl_xor(t2[qf], 16));
        merge2(t1[qf], t2[qf], __shfl_xor(t1[qf], 32), __shfl_xor(t2[qf], 32));
    }
    __syncthreads();                                  // everyone is done reading the query image
    if (scorer && lgrp == 0) {
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
            s_t2[(wave * 64 + qf * 16 + lrow) * 2 + 0] = t1[qf];
            s_t2[(wave * 64 + qf * 16 + lrow) * 2 + 1] = t2[qf];
        }
    }
    __syncthreads();
    if (tid < 64) {
        float a1 = s_t2[tid * 2], a2 = s_t2[tid * 2 + 1];
#pragma unroll
        for (int w = 1; w < 4; ++w) merge2(a1, a2, s_t2[(w * 64 + tid) * 2], s_t2[(w * 64 + tid) * 2 + 1]);
        const int nblk = p.S / 64;
        p.top2[((size_t)tid * nblk + blockIdx.x) * 2 + 0] = a1;
        p.top2[((size_t)tid * nblk + blockIdx.x) * 2 + 1] = a2;
    }
}

}  // namespace atlas
