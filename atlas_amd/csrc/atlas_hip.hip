// atlas_hip.hip — gfx950 (MI355X, CDNA4) kernels + C-ABI for the Atlas exact-MIPS hot path.
//
// Kernels (DESIGN.md §4 has the roofline of each):
//   prep_queries_kernel   queries -> fp16 (== `.half()`, src/index.py:117), MFMA B-fragment
//                         order for LDS, row-major copy for rescoring, per-query eps
//   scan_kernel           the hot one: streams the (N,768) fp16 slab once, 16x16x32 f16 MFMA
//                         against 64 LDS-resident queries, per-lane threshold filter,
//                         per-workgroup candidate lists with certified pruning margins.
//                         Scores never reach HBM (replaces matmul+topk, index.py:117-118)
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

#include "common.h"
#include "../../include/atlas_hip.h"

using namespace atlas;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define D_FAST 768               // EMBEDDINGS_DIM, src/retrievers.py:13
#define KSTEPS (D_FAST / 32)     // 24 MFMA k-steps of 32
#define QCHUNK 64                // queries per slab pass (4 MFMA column groups of 16)
#define QFRAG_U4 (KSTEPS * 4 * 64)   // uint4 elements of the fragment-ordered query image
#define K_FAST_MAX 256
#define K_EXACT_MAX 2048
#define MERGE_SMAX 2048          // max candidates rescored per query in the merge

static __device__ __forceinline__ float neg_inf() { return bits_f32(0xff800000u); }
static __device__ __forceinline__ float pos_inf() { return bits_f32(0x7f800000u); }

// raw workgroup barrier that orders LDS only: prefetched global loads stay in flight
// (a __syncthreads() here would drain vmcnt once per tile; cdna guide §5 "Pipelining across barriers")
static __device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// prep: one block per query slot (64 slots; slots >= nq are zero queries with eps 0)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
prep_queries_kernel(const void* __restrict__ q, int q_dtype, int q0, int nq, int d, float pmax,
                    uint16_t* __restrict__ qrow /*[64][d]*/, uint16_t* __restrict__ qfrag /*fragment order*/,
                    float* __restrict__ qeps /*[64]*/) {
    const int j = blockIdx.x;
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
// scan
// ------------------------------------------------------------------------------------------
struct ScanParams {
    const uint16_t* slab;     // [N][768] fp16
    int64_t N;
    const uint4* qfrag;       // [24][4][64] uint4
    const float* qeps;        // [64]
    uint2* lists;             // [G][64][cap]  {f32 bits of approx score, row}
    uint32_t* counts;         // [G][64]
    uint32_t* gstat;          // [0] max row sumsq (float bits, atomicMax)  [1] flags
    uint32_t* qflag;          // [64] per-query fallback flag (band overflow)
    int64_t rows_per_wg;
    int nq, k, cap, keep_max;
    float pmax2_hint;
};

struct ScanSmem {   // byte offsets into dynamic LDS
    static constexpr int q_off = 0;                       // 98304 B
    static constexpr int theta_off = QFRAG_U4 * 16;       // 64 f32
    static constexpr int cnt_off = theta_off + 256;       // 64 u32
    static constexpr int flag_off = cnt_off + 256;        // 64 B (word 0 used)
    static constexpr int keys_off = flag_off + 64;        // NW * cap u32
};

template <int NW, int PF, int RING>
__global__ void __launch_bounds__(NW * 64)
scan_kernel(const ScanParams p) {
    // RING slots of PF fragments: RING-1 k-steps of loads in flight while one slot is consumed
    static_assert(KSTEPS % RING == 0 && RING >= 2, "prefetch ring must divide the k-steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_q = (uint4*)(smem + ScanSmem::q_off);
    float* s_theta = (float*)(smem + ScanSmem::theta_off);
    uint32_t* s_cnt = (uint32_t*)(smem + ScanSmem::cnt_off);
    uint32_t* s_flag = (uint32_t*)(smem + ScanSmem::flag_off);   // plain LDS words; ordered by wg_barrier_lds()
    uint32_t* s_keys = (uint32_t*)(smem + ScanSmem::keys_off);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lgrp = lane >> 4;
    constexpr int TILE = NW * PF * 16;       // rows per workgroup tile
    constexpr int ROWB = D_FAST * 2;         // bytes per slab row
    constexpr int RPT = KSTEPS / RING;       // ring revolutions per tile

    for (int i = tid; i < QFRAG_U4; i += NW * 64) s_q[i] = p.qfrag[i];
    if (tid < 64) {
        s_theta[tid] = (tid < p.nq) ? neg_inf() : pos_inf();
        s_cnt[tid] = 0;
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();

    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    int64_t r_end = r_begin + p.rows_per_wg;
    if (r_end > p.N) r_end = p.N;
    const int ntiles = (r_end > r_begin) ? (int)((r_end - r_begin + TILE - 1) / TILE) : 0;   // workgroup-uniform
    uint2* my_lists = p.lists + (size_t)blockIdx.x * 64 * p.cap;

    // Passage rows stream HBM -> VGPR through buffer loads (cdna guide T8). ONE descriptor per
    // wave spans [first row of this wave's first tile, N). The hardware bounds check covers
    // voffset + immediate only (not soffset), so everything that selects a ROW lives in the
    // per-lane voffset (one VGPR per fragment, bumped once per tile) and rows at or past N read
    // as zero; the k-step (< one row) rides in the scalar offset. No address VALU in the k-loop.
    //   lane l loads row (l & 15) of fragment pf, bytes [64*s + 16*(l>>4), +16)   (MFMA A operand)
    const int64_t wrow0 = r_begin + (int64_t)wave * PF * 16;
    int64_t span = (wrow0 < p.N) ? (p.N - wrow0) * (int64_t)ROWB : 0;
    if (span > 0xfffffff0ll) span = 0xfffffff0ll;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)p.slab + (span > 0 ? wrow0 : 0) * (int64_t)ROWB), 0, (int)span, 0x00020000);

    // fill cursor: (rows of the tile being fetched -> vo[], k-step -> fill_step); it runs RING-1
    // steps ahead of the consumer. Past the last tile it keeps walking forward: those loads hit
    // rows of the next workgroup's range (harmless) or fall out of bounds (return 0).
    unsigned vo[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) vo[pf] = (unsigned)((pf * 16 + lrow) * ROWB + lgrp * 16);
    int fill_step = 0;
    auto fill_advance = [&]() {
        ++fill_step;
        if (fill_step == KSTEPS) {          // scalar condition: next tile of this wave
            fill_step = 0;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) vo[pf] += (unsigned)(TILE * ROWB);
        }
    };

    u32x4 abuf[RING][PF];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
            abuf[s][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, 0);
        fill_advance();
        // keep issue order == ring order: hipcc's waitcnt for slot 0 is the minimum over the loop
        // entry and the back edge, so a shuffled prologue would cost ring depth on every revolution
        __builtin_amdgcn_sched_barrier(0);
    }

    f32x4 acc[PF][4];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) acc[pf][qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float nrm[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) nrm[pf] = 0.f;
    float pm = 0.0f;   // running max of row sum-of-squares seen by this lane's row group

    // rows relative to r_begin fit 32 bits (plan guarantees rows_per_wg * 1536 < 2^32)
    const int nrows = (int)(r_end > r_begin ? r_end - r_begin : 0);
    const uint32_t gbase = (uint32_t)r_begin;          // shard-local row ids are < 2^32
    int row0 = wave * PF * 16;   // first row (relative) of this wave's current tile
    int par = 0;               // tile parity (double-buffers the compaction-request flag)
    int cstep = 0;             // consumer k-step inside the tile

    // One flat loop over ring revolutions of all tiles: the ring rotation is the same every
    // iteration (no register shuffling at tile boundaries), and the per-tile work (filter,
    // barrier) hangs off every RPT-th revolution.
#pragma unroll 1
    for (int rev = 0; rev < ntiles * RPT; ++rev) {
        const uint4* bq = s_q + lane + cstep * (4 * 64);
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            // refill the slot freed by the previous step first (its loads stay in flight for
            // RING-1 steps), then consume slot j. sched_barrier pins that order: left alone,
            // hipcc sinks the loads to the loop end and waits vmcnt(0) at the top.
            const int fill = (j + RING - 1) % RING;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
                abuf[fill][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, 0);
            fill_advance();
            __builtin_amdgcn_sched_barrier(0);
            uint4 b[4];
#pragma unroll
            for (int qf = 0; qf < 4; ++qf) b[qf] = bq[(j * 4 + qf) * 64];
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const u32x4 a = abuf[j][pf];
                const f16x8 av = __builtin_bit_cast(f16x8, a);
#pragma unroll
                for (int qf = 0; qf < 4; ++qf)
                    acc[pf][qf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        av, __builtin_bit_cast(f16x8, b[qf]), acc[pf][qf], 0, 0, 0);
                // row sum of squares (certifies pmax_hint): 4 x v_dot2_f32_f16
                // (element copies first: __builtin_bit_cast straight on an ext-vector element
                //  reads element 0 for every swizzle on ROCm 7.2's clang)
                const unsigned ax = a.x, ay = a.y, az = a.z, aw = a.w;
                const f16x2 h0 = __builtin_bit_cast(f16x2, ax), h1 = __builtin_bit_cast(f16x2, ay);
                const f16x2 h2 = __builtin_bit_cast(f16x2, az), h3 = __builtin_bit_cast(f16x2, aw);
                nrm[pf] = __builtin_amdgcn_fdot2(h0, h0, nrm[pf], false);
                nrm[pf] = __builtin_amdgcn_fdot2(h1, h1, nrm[pf], false);
                nrm[pf] = __builtin_amdgcn_fdot2(h2, h2, nrm[pf], false);
                nrm[pf] = __builtin_amdgcn_fdot2(h3, h3, nrm[pf], false);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cstep += RING;
        if (cstep < KSTEPS) continue;

        // ------------------------- end of a tile: filter --------------------------------
        cstep = 0;
        if (row0 < nrows) {        // wave-uniform
            // full row norms: the 4 lanes {l, l+16, l+32, l+48} hold the 4 k-groups of row l&15
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                float x = nrm[pf];
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                pm = fmaxf(pm, x);
            }
            // rows past the end of this workgroup's range never become candidates
            if (row0 + PF * 16 > nrows) {
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + pf * 16 + lgrp * 4 + r >= nrows) {
#pragma unroll
                            for (int qf = 0; qf < 4; ++qf) acc[pf][qf][r] = neg_inf();
                        }
            }
            // threshold filter: lane l owns query 16*qf + (l&15) in acc[.][qf]
            float th[4];
#pragma unroll
            for (int qf = 0; qf < 4; ++qf) th[qf] = s_theta[qf * 16 + lrow];
            bool any = false;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                for (int qf = 0; qf < 4; ++qf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) any |= acc[pf][qf][r] > th[qf];
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                // rare path: append candidates to the workgroup's per-query lists
                const uint32_t rbase = gbase + (uint32_t)row0 + (uint32_t)lgrp * 4u;
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                    for (int qf = 0; qf < 4; ++qf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[pf][qf][r];
                            if (v > th[qf]) {
                                const int qq = qf * 16 + lrow;
                                const uint32_t slot = atomicAdd(&s_cnt[qq], 1u);
                                if (slot < (uint32_t)p.cap)
                                    my_lists[(uint32_t)(qq * p.cap) + slot] =
                                        make_uint2(f32_bits(v), rbase + (uint32_t)(pf * 16 + r));
                                if (slot >= (uint32_t)p.keep_max) s_flag[par] = 1u;
                            }
                        }
                // Drain the list stores here, with the builtin (hipcc's waitcnt pass sees it, unlike
                // inline asm): gfx9 counts stores and loads in one vmcnt and assumes they retire out
                // of order, so a store left pending on this rare path would force vmcnt(0) in front
                // of every ring slot of the hot loop.
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
            }
        }
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            nrm[pf] = 0.f;
#pragma unroll
            for (int qf = 0; qf < 4; ++qf) acc[pf][qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        wg_barrier_lds();
        if (s_flag[par] != 0u) {
            // compaction: some list crossed keep_max. Every wave has drained its list stores
            // (above); each wave compacts the queries it owns (lists are private to this CU).
            wg_barrier_lds();
            if (tid == 0) s_flag[par] = 0u;
            for (int qq = wave; qq < p.nq; qq += NW) {
                const uint32_t n = s_cnt[qq];
                if (n <= (uint32_t)p.k) continue;                 // nothing can be pruned yet
                uint2* L = my_lists + (size_t)qq * p.cap;
                uint32_t* K = s_keys + wave * p.cap;
                for (uint32_t i = lane; i < n; i += 64) K[i] = f32_order_key(bits_f32(L[i].x));
                // k-th largest key: greedy bit search (largest v with count(keys >= v) >= k)
                uint32_t prefix = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = prefix | (1u << bit);
                    uint32_t c = 0;
                    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        const bool ge = (i < n) && (K[i] >= cand);
                        c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ge));
                    }
                    if (c >= (uint32_t)p.k) prefix = cand;
                }
                const float T = f32_from_order_key(prefix);
                const float theta = prune_threshold(T, p.qeps[qq]);
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
                    if (kept > (uint32_t)p.keep_max) {
                        // candidate band wider than the list (mass ties): hand this query to
                        // the exact path and stop collecting for it
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
        }
        row0 += TILE;
        par ^= 1;
    }

    __syncthreads();
    if (tid < 64) p.counts[(size_t)blockIdx.x * 64 + tid] = (tid < p.nq) ? s_cnt[tid] : 0u;
    // publish the largest row norm^2 seen (x1.001: v_dot2 accumulates in fp32)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
    if (lane == 0 && pm > 0.f) {
        pm *= 1.001f;
        atomicMax(&p.gstat[0], f32_bits(pm));
        if (pm > p.pmax2_hint) atomicOr(&p.gstat[1], (uint32_t)ATLAS_F_PMAX_VIOLATION);
    }
}

// ------------------------------------------------------------------------------------------
// merge + exact rescoring: one block per query
// ------------------------------------------------------------------------------------------
struct MergeParams {
    const uint16_t* slab; int64_t N; int d;
    const uint16_t* qrow;        // [64][d]
    const float* qeps;           // [64]
    const uint2* lists; const uint32_t* counts; int G; int cap;
    const uint32_t* gstat; const uint32_t* qflag;
    int k, q0;                   // q0: first query of this chunk (output row offset)
    uint16_t* out_score; int64_t* out_idx; int32_t* out_status;
};

static __device__ __forceinline__ double exact_dot_dev(const uint16_t* __restrict__ qs /*LDS*/,
                                                       const uint16_t* __restrict__ prow, int d) {
    // canonical order (common.h exact_dot_f16): chain j takes elements j, j+8, ...; fixed combine tree.
    // Chains live in named registers (a runtime-indexed array would go to scratch).
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    int i = 0;
    if ((d & 7) == 0) {
        const uint4* p4 = (const uint4*)prow;
        const uint4* q4 = (const uint4*)qs;
        for (; i < d; i += 8) {
            const f16x8 pv = __builtin_bit_cast(f16x8, p4[i >> 3]);
            const f16x8 qv = __builtin_bit_cast(f16x8, q4[i >> 3]);
            c0 += (double)(float)qv[0] * (double)(float)pv[0];
            c1 += (double)(float)qv[1] * (double)(float)pv[1];
            c2 += (double)(float)qv[2] * (double)(float)pv[2];
            c3 += (double)(float)qv[3] * (double)(float)pv[3];
            c4 += (double)(float)qv[4] * (double)(float)pv[4];
            c5 += (double)(float)qv[5] * (double)(float)pv[5];
            c6 += (double)(float)qv[6] * (double)(float)pv[6];
            c7 += (double)(float)qv[7] * (double)(float)pv[7];
        }
    } else {
#define ATLAS_TERM(j) (f16_bits_to_f64(qs[i + j]) * f16_bits_to_f64(prow[i + j]))
        for (; i + 8 <= d; i += 8) {
            c0 += ATLAS_TERM(0); c1 += ATLAS_TERM(1); c2 += ATLAS_TERM(2); c3 += ATLAS_TERM(3);
            c4 += ATLAS_TERM(4); c5 += ATLAS_TERM(5); c6 += ATLAS_TERM(6); c7 += ATLAS_TERM(7);
        }
        const int rem = d - i;
        if (rem > 0) c0 += ATLAS_TERM(0);
        if (rem > 1) c1 += ATLAS_TERM(1);
        if (rem > 2) c2 += ATLAS_TERM(2);
        if (rem > 3) c3 += ATLAS_TERM(3);
        if (rem > 4) c4 += ATLAS_TERM(4);
        if (rem > 5) c5 += ATLAS_TERM(5);
        if (rem > 6) c6 += ATLAS_TERM(6);
#undef ATLAS_TERM
    }
    return ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
}

template <int NT>
__global__ void __launch_bounds__(NT)
merge_rescore_kernel(const MergeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: qs[d] u16 (padded to 16 B) | hist[256] | misc[8] | s_row[SMAX] u32 | s_app[SMAX] f32 | s_key[SMAX] u64
    uint16_t* qs = (uint16_t*)smem;
    const int qbytes = ((p.d * 2 + 15) / 16) * 16;
    uint32_t* hist = (uint32_t*)(smem + qbytes);
    uint32_t* misc = hist + 256;            // [0] prefix [1] remaining k [2] nsurv [3] total [4] maxerr bits [5] flags
    uint32_t* s_row = misc + 8;
    float* s_app = (float*)(s_row + MERGE_SMAX);
    uint64_t* s_key = (uint64_t*)(s_app + MERGE_SMAX);

    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = NT / 64;
    const int k = p.k;
    uint16_t* o_score = p.out_score + (size_t)(p.q0 + q) * k;
    int64_t* o_idx = p.out_idx + (size_t)(p.q0 + q) * k;
    int32_t* o_qst = p.out_status + ATLAS_STATUS_HEADER + p.q0 + q;

    if (p.qflag[q] != 0u) {   // scan overflowed this query's band -> exact path will fill the row
        if (tid == 0) {
            *o_qst = ATLAS_Q_FALLBACK;
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_FALLBACK);
            atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_FALLBACK], 1u);
        }
        return;
    }
    for (int i = tid; i < p.d; i += NT) qs[i] = p.qrow[(size_t)q * p.d + i];
    if (tid < 8) misc[tid] = 0;
    if (tid == 1) misc[1] = (uint32_t)k;
    __syncthreads();

    // ---- k-th largest approximate score over all lists: 4 x 8-bit radix passes ----
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += NT) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = misc[0];
        uint32_t local_total = 0;
        for (int w = wave; w < p.G; w += NWV) {
            const uint32_t n = p.counts[(size_t)w * 64 + q];
            const uint2* L = p.lists + ((size_t)w * 64 + q) * p.cap;
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t key = f32_order_key(bits_f32(L[i].x));
                if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                    atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            local_total += n;
        }
        if (pass == 0 && lane == 0) atomicAdd(&misc[3], local_total);
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = misc[1], b = 255, acc = 0;
            // from the top bin down: first bin where the cumulative count reaches rem
            for (int bi = 255; bi >= 0; --bi) {
                if (acc + hist[bi] >= rem) { b = (uint32_t)bi; break; }
                acc += hist[bi];
                if (bi == 0) b = 0;
            }
            misc[1] = rem - acc;          // rank inside the chosen bin
            misc[0] = prefix | (b << shift);
        }
        __syncthreads();
    }
    const uint32_t total = misc[3];
    float theta = neg_inf();
    const float eps = p.qeps[q];
    if (total >= (uint32_t)k) theta = prune_threshold(f32_from_order_key(misc[0]), eps);

    // ---- collect the candidate band ----
    for (int w = wave; w < p.G; w += NWV) {
        const uint32_t n = p.counts[(size_t)w * 64 + q];
        const uint2* L = p.lists + ((size_t)w * 64 + q) * p.cap;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint2 e = L[i];
            if (bits_f32(e.x) > theta) {
                const uint32_t s = atomicAdd(&misc[2], 1u);
                if (s < MERGE_SMAX) { s_row[s] = e.y; s_app[s] = bits_f32(e.x); }
            }
        }
    }
    __syncthreads();
    const uint32_t nsurv = misc[2];
    if (nsurv > MERGE_SMAX) {
        if (tid == 0) {
            *o_qst = ATLAS_Q_FALLBACK;
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_FALLBACK);
            atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_FALLBACK], 1u);
        }
        return;
    }

    // ---- exact rescoring in the canonical order, canonical keys ----
    for (uint32_t i = tid; i < nsurv; i += NT) {
        const uint32_t row = s_row[i];
        const double s = exact_dot_dev(qs, p.slab + (size_t)row * p.d, p.d);
        const uint16_t h = f64_to_f16_bits(s);
        s_key[i] = local_key(h, row);
        // a-posteriori check of the error model on every rescored row
        const float err = fabsf((float)((double)s_app[i] - s));
        const float ratio = eps > 0.f ? err / eps : (err > 0.f ? 2.0f : 0.0f);
        atomicMax(&misc[4], f32_bits(ratio));
    }
    __syncthreads();
    // rank by counting (keys are unique: the row is part of the key)
    for (uint32_t i = tid; i < nsurv; i += NT) {
        const uint64_t ki = s_key[i];
        uint32_t pos = 0;
        for (uint32_t j = 0; j < nsurv; ++j) pos += (s_key[j] > ki) ? 1u : 0u;
        if (pos < (uint32_t)k) {
            o_score[pos] = f16_from_order_key((uint16_t)(ki >> 32));
            o_idx[pos] = (int64_t)(0xffffffffu - (uint32_t)ki);
        }
    }
    for (uint32_t i = nsurv + tid; i < (uint32_t)k; i += NT) { o_score[i] = 0xfc00; o_idx[i] = -1; }
    if (tid == 0) {
        *o_qst = ATLAS_Q_OK;
        atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_CANDIDATES], total);
        atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_RESCORED], nsurv);
        atomicMax((uint32_t*)&p.out_status[ATLAS_ST_MAXERR_BITS], misc[4]);
        if (bits_f32(misc[4]) > 1.0f)
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_EPS_VIOLATION);
        // scan-level flags / pmax (idempotent across blocks)
        atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], p.gstat[1]);
        atomicMax((uint32_t*)&p.out_status[ATLAS_ST_PMAX_BITS], f32_bits(sqrtf(bits_f32(p.gstat[0])) * 1.000001f));
    }
}

// ------------------------------------------------------------------------------------------
// exact path (no MFMA): canonical keys for every row, then an exact radix select
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
exact_keys_kernel(const uint16_t* __restrict__ slab, int64_t N, int d, const uint16_t* __restrict__ qrow,
                  uint64_t* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* qs = (uint16_t*)smem;
    for (int i = threadIdx.x; i < d; i += 256) qs[i] = qrow[i];
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (int64_t)gridDim.x * 256) {
        const double s = exact_dot_dev(qs, slab + (size_t)r * d, d);
        keys[r] = local_key(f64_to_f16_bits(s), (uint32_t)r);
    }
}

// state derived from the histograms of the passes already done: (prefix, remaining rank)
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
        prefix |= (uint64_t)b << (56 - 8 * ps);
    }
}

__global__ void __launch_bounds__(256)
exact_hist_kernel(const uint64_t* __restrict__ keys, int64_t N, int k, int pass, uint32_t* __restrict__ hists) {
    __shared__ uint32_t lh[256];
    __shared__ uint64_t s_prefix;
    lh[threadIdx.x] = 0;
    if (threadIdx.x == 0) { uint64_t pf; uint32_t rem; radix_state(hists, pass, k, pf, rem); s_prefix = pf; }
    __syncthreads();
    const uint64_t prefix = s_prefix;
    const int shift = 56 - 8 * pass;
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
exact_collect_kernel(const uint64_t* __restrict__ keys, int64_t N, int k, const uint32_t* __restrict__ hists,
                     uint64_t* __restrict__ sel /*[k]*/, uint32_t* __restrict__ nsel) {
    __shared__ uint64_t s_kth;
    if (threadIdx.x == 0) {
        uint64_t pf; uint32_t rem;
        radix_state(hists, 8, k, pf, rem);
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

__global__ void __launch_bounds__(256)
exact_emit_kernel(const uint64_t* __restrict__ sel, const uint32_t* __restrict__ nsel, int k,
                  uint16_t* __restrict__ o_score, int64_t* __restrict__ o_idx) {
    uint32_t n = *nsel;
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
    if (lane == 0 && best > 0.f) atomicMax(out_bits, f32_bits(sqrtf(best) * 1.00001f));
}

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {

constexpr int SCAN_NW = 8, SCAN_PF = 4, SCAN_RING = 4;
constexpr int SCAN_TILE = SCAN_NW * SCAN_PF * 16;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct ScanPlan {
    int G;               // workgroups
    int64_t rows_per_wg;
    int keep_max, cap;
    size_t off_qfrag, off_qrow, off_qeps, off_counts, off_gstat, off_qflag, off_lists, total;
    size_t scan_lds, merge_lds;
};

int device_cus() {
    static int cus = 0;     // immutable after first query; benign cache
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

ScanPlan make_plan(int64_t N, int d, int k, int cus) {
    ScanPlan pl{};
    // one workgroup per CU (the 96 KB query image allows exactly one resident workgroup);
    // every workgroup gets a contiguous, 16-row aligned range of equal size
    int64_t frags = (N + 15) / 16;
    int64_t G = cus;
    if (G > (frags + SCAN_NW - 1) / SCAN_NW) G = (frags + SCAN_NW - 1) / SCAN_NW;
    if (G < 1) G = 1;
    pl.G = (int)G;
    pl.rows_per_wg = ((frags + G - 1) / G) * 16;
    pl.keep_max = (2 * k > k + 64) ? 2 * k : k + 64;
    pl.cap = pl.keep_max + SCAN_TILE;
    size_t o = 0;
    pl.off_qfrag = o;  o += (size_t)QFRAG_U4 * 16;
    pl.off_qrow = o;   o += align_up((size_t)QCHUNK * d * 2, 256);
    pl.off_qeps = o;   o += 256;
    pl.off_gstat = o;  o += 256;
    pl.off_qflag = o;  o += 256;
    pl.off_counts = o; o += align_up((size_t)pl.G * 64 * 4, 256);
    pl.off_lists = o;  o += (size_t)pl.G * 64 * pl.cap * 8;
    pl.total = align_up(o, 256);
    pl.scan_lds = (size_t)ScanSmem::keys_off + (size_t)SCAN_NW * pl.cap * 4;
    pl.merge_lds = align_up((size_t)d * 2, 16) + 256 * 4 + 8 * 4 + (size_t)MERGE_SMAX * (4 + 4 + 8);
    return pl;
}

struct ExactPlan { size_t off_qrow, off_qfrag_dummy, off_qeps, off_keys, off_hists, off_sel, off_nsel, total; };
ExactPlan make_exact_plan(int64_t N, int d, int k) {
    ExactPlan e{}; size_t o = 0;
    e.off_qrow = o;  o += align_up((size_t)QCHUNK * d * 2, 256);
    e.off_qfrag_dummy = o; o += 256;
    e.off_qeps = o;  o += 256;
    e.off_hists = o; o += 8 * 256 * 4;
    e.off_nsel = o;  o += 256;
    e.off_sel = o;   o += align_up((size_t)k * 8, 256);
    e.off_keys = o;  o += align_up((size_t)(N > 0 ? N : 1) * 8, 256);
    e.total = o;
    return e;
}

template <typename KernelT>
void allow_lds(KernelT kern) {   // opt in to the full 160 KiB of LDS (idempotent)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

extern "C" {

int atlas_abi_version(void) { return ATLAS_ABI_VERSION; }
const char* atlas_build_info(void) { return "atlas_hip gfx950 scan(NW=8,PF=4,RING=4) " __DATE__ " " __TIME__; }

size_t atlas_scan_topk_workspace_bytes(int64_t N, int B, int d, int k) {
    (void)B;
    if (N < 0 || d <= 0 || k <= 0) return 0;
    return make_plan(N, d, k, device_cus()).total;
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
    if (!q || (!slab_f16 && N > 0) || !out_score_f16 || !out_idx || !out_status || !ws) return ATLAS_E_BADARG;
    if (B <= 0 || k <= 0 || N < 0 || q_dtype < 0 || q_dtype > 2 || !(pmax_hint >= 0.f)) return ATLAS_E_BADARG;
    if (d != D_FAST || k > K_FAST_MAX || N >= (int64_t)0xffffffffll) return ATLAS_E_UNSUPPORTED;
    const ScanPlan pl = make_plan(N, d, k, device_cus());
    // per-lane byte offsets inside one workgroup's range are 32-bit (buffer voffset)
    if ((pl.rows_per_wg + 2 * SCAN_TILE) * (int64_t)(D_FAST * 2) >= (int64_t)0xfff00000ll) return ATLAS_E_UNSUPPORTED;
    if (ws_bytes < pl.total) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* w = (unsigned char*)ws;

    auto scan = scan_kernel<SCAN_NW, SCAN_PF, SCAN_RING>;
    auto merge = merge_rescore_kernel<512>;
    allow_lds(scan);
    allow_lds(merge);

    hipError_t e = hipMemsetAsync(out_status, 0, sizeof(int32_t) * (ATLAS_STATUS_HEADER + (size_t)B), stream);
    if (e != hipSuccess) return (int)e;

    for (int q0 = 0; q0 < B; q0 += QCHUNK) {
        const int nq = (B - q0 < QCHUNK) ? (B - q0) : QCHUNK;
        e = hipMemsetAsync(w + pl.off_gstat, 0, 512, stream);          // gstat + qflag
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(prep_queries_kernel, dim3(QCHUNK), dim3(256), 0, stream, q, q_dtype, q0, nq, d,
                           pmax_hint, (uint16_t*)(w + pl.off_qrow), (uint16_t*)(w + pl.off_qfrag),
                           (float*)(w + pl.off_qeps));
        ScanParams sp{};
        sp.slab = (const uint16_t*)slab_f16; sp.N = N;
        sp.qfrag = (const uint4*)(w + pl.off_qfrag); sp.qeps = (const float*)(w + pl.off_qeps);
        sp.lists = (uint2*)(w + pl.off_lists); sp.counts = (uint32_t*)(w + pl.off_counts);
        sp.gstat = (uint32_t*)(w + pl.off_gstat); sp.qflag = (uint32_t*)(w + pl.off_qflag);
        sp.rows_per_wg = pl.rows_per_wg; sp.nq = nq; sp.k = k; sp.cap = pl.cap; sp.keep_max = pl.keep_max;
        sp.pmax2_hint = pmax_hint * pmax_hint;
        if (q0 == 0 && ev_scan_begin) (void)hipEventRecord((hipEvent_t)ev_scan_begin, stream);
        hipLaunchKernelGGL(scan, dim3(pl.G), dim3(SCAN_NW * 64), pl.scan_lds, stream, sp);
        if (q0 == 0 && ev_scan_end) (void)hipEventRecord((hipEvent_t)ev_scan_end, stream);
        MergeParams mp{};
        mp.slab = (const uint16_t*)slab_f16; mp.N = N; mp.d = d;
        mp.qrow = (const uint16_t*)(w + pl.off_qrow); mp.qeps = sp.qeps;
        mp.lists = sp.lists; mp.counts = sp.counts; mp.G = pl.G; mp.cap = pl.cap;
        mp.gstat = sp.gstat; mp.qflag = sp.qflag; mp.k = k; mp.q0 = q0;
        mp.out_score = (uint16_t*)out_score_f16; mp.out_idx = out_idx; mp.out_status = out_status;
        hipLaunchKernelGGL(merge, dim3(nq), dim3(512), pl.merge_lds, stream, mp);
    }
    return (int)hipGetLastError();
}

size_t atlas_exact_topk_workspace_bytes(int64_t N, int B, int d, int k) {
    (void)B;
    if (N < 0 || d <= 0 || k <= 0) return 0;
    return make_exact_plan(N, d, k).total;
}

int atlas_exact_topk(const void* q, int q_dtype, const void* slab_f16, int64_t N, int B, int d, int k,
                     void* out_score_f16, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream_) {
    if (!q || (!slab_f16 && N > 0) || !out_score_f16 || !out_idx || !ws) return ATLAS_E_BADARG;
    if (B <= 0 || k <= 0 || N < 0 || d <= 0 || q_dtype < 0 || q_dtype > 2) return ATLAS_E_BADARG;
    if (k > K_EXACT_MAX || d > 16384 || N >= (int64_t)0xffffffffll) return ATLAS_E_UNSUPPORTED;
    const ExactPlan pl = make_exact_plan(N, d, k);
    if (ws_bytes < pl.total) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* w = (unsigned char*)ws;
    const int cus = device_cus();
    int grid = (int)((N + 255) / 256);
    if (grid > cus * 8) grid = cus * 8;
    if (grid < 1) grid = 1;
    for (int q0 = 0; q0 < B; q0 += QCHUNK) {
        const int nq = (B - q0 < QCHUNK) ? (B - q0) : QCHUNK;
        // prep reuses the fast path's converter (no fragment image: qfrag == nullptr)
        hipLaunchKernelGGL(prep_queries_kernel, dim3(QCHUNK), dim3(256), 0, stream, q, q_dtype, q0, nq, d, 0.f,
                           (uint16_t*)(w + pl.off_qrow), (uint16_t*)nullptr, (float*)(w + pl.off_qeps));
        for (int j = 0; j < nq; ++j) {
            hipError_t e = hipMemsetAsync(w + pl.off_hists, 0, 8 * 256 * 4 + 256, stream);   // hists + nsel
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(exact_keys_kernel, dim3(grid), dim3(256), (size_t)align_up((size_t)d * 2, 16), stream,
                               (const uint16_t*)slab_f16, N, d, (const uint16_t*)(w + pl.off_qrow) + (size_t)j * d,
                               (uint64_t*)(w + pl.off_keys));
            for (int pass = 0; pass < 8; ++pass)
                hipLaunchKernelGGL(exact_hist_kernel, dim3(grid), dim3(256), 0, stream,
                                   (const uint64_t*)(w + pl.off_keys), N, k, pass, (uint32_t*)(w + pl.off_hists));
            hipLaunchKernelGGL(exact_collect_kernel, dim3(grid), dim3(256), 0, stream,
                               (const uint64_t*)(w + pl.off_keys), N, k, (const uint32_t*)(w + pl.off_hists),
                               (uint64_t*)(w + pl.off_sel), (uint32_t*)(w + pl.off_nsel));
            hipLaunchKernelGGL(exact_emit_kernel, dim3(1), dim3(256), 0, stream, (const uint64_t*)(w + pl.off_sel),
                               (const uint32_t*)(w + pl.off_nsel), k,
                               (uint16_t*)out_score_f16 + (size_t)(q0 + j) * k, out_idx + (size_t)(q0 + j) * k);
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
    int grid = (int)((N + 3) / 4);
    if (grid > device_cus() * 16) grid = device_cus() * 16;
    hipLaunchKernelGGL(slab_pmax_kernel, dim3(grid), dim3(256), 0, stream, (const uint16_t*)slab_f16, N, d,
                       (uint32_t*)out_pmax);
    return (int)hipGetLastError();
}

}  // extern "C"
