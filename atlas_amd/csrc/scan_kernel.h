// scan_kernel.h — the fused MFMA scan + candidate-list kernel, as a template so that the
// product (atlas_hip.hip) and the tuning harness (microbench.hip) instantiate the same source.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/atlas_hip.h"

#ifndef ATLAS_TUNING
#define ATLAS_TUNING 0
#endif
// wall-clock stamps exist in the tuning build only: the product kernel carries no trace of them
#if ATLAS_TUNING
#define ATLAS_SCAN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATLAS_SCAN_STAMP(i) do { } while (0)
#endif

namespace atlas {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define D_FAST 768               // EMBEDDINGS_DIM, src/retrievers.py:13
#define KSTEPS (D_FAST / 32)     // 24 MFMA k-steps of 32
#define QCHUNK 64                // queries per slab pass (4 MFMA column groups of 16)
#define QFRAG_U4 (KSTEPS * 4 * 64)   // uint4 elements of the fragment-ordered query image

static __device__ __forceinline__ float neg_inf() { return bits_f32(0xff800000u); }
static __device__ __forceinline__ float pos_inf() { return bits_f32(0x7f800000u); }

// raw workgroup barrier that orders LDS only: prefetched global loads stay in flight
// (a __syncthreads() here would drain vmcnt once per tile; cdna guide §5 "Pipelining across barriers")
static __device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// global -> LDS copy of the 96 KiB query image with all loads of a thread in flight at once
// (the naive loop compiles to load / s_waitcnt vmcnt(0) / ds_write per element: one L2 latency each)
template <int NT>
static __device__ __forceinline__ void copy_qfrag_to_lds(uint4* __restrict__ s_q, const uint4* __restrict__ g, const int tid) {
    constexpr int PER = (QFRAG_U4 + NT - 1) / NT;      // 6 for 1024 threads, 24 for 256
    constexpr int BATCH = PER < 12 ? PER : 12;
#pragma unroll 1
    for (int b0 = 0; b0 < PER; b0 += BATCH) {
        uint4 tmp[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = tid + (b0 + u) * NT;
            tmp[u] = (i < QFRAG_U4) ? g[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = tid + (b0 + u) * NT;
            if (i < QFRAG_U4) s_q[i] = tmp[u];
        }
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
    const float* theta0;      // [64] initial per-query thresholds from the sample pre-pass (or -inf)
    uint2* lists;             // [64][G][cap]  {f32 bits of approx score, row}: the candidates of workgroup g for query q
    uint32_t* list_cnt;       // [64][G] entries of lists[q][g] at the end of the scan (plain stores: every workgroup writes its 64)
    uint32_t* wg_stat;        // [G][2]  per workgroup: largest row sum of squares seen (float bits) | ATLAS_F_* flags
    uint32_t* qflag;          // [64] per-query fallback flag (band overflow; plain idempotent stores)
    int64_t rows_per_wg;
    int nq, k, cap, keep_max;
    int buf_cap;              // entries of the LDS candidate buffer
    int flush_at;             // buffer fill at which a flush into the global lists is requested
    float pmax2_hint;
    unsigned long long* dbg;  // tuning build only (atlas_tune_set_scan_stamps): 8 wall-clock stamps (100 MHz, common to all XCDs) per workgroup; null in production
};

struct ScanSmem {   // byte offsets into dynamic LDS
    static constexpr int q_off = 0;                       // 98304 B
    static constexpr int theta_off = QFRAG_U4 * 16;       // 64 f32
    static constexpr int cnt_off = theta_off + 256;       // 64 u32
    static constexpr int flag_off = cnt_off + 256;        // 64 B: [0],[1] flush/compaction request by tile parity, [2] buffer fill
    static constexpr int aux_off = flag_off + 64;         // 3 x 64 u32: per-query base / running position / buffered count (final hand-over)
    static constexpr int buf_off = aux_off + 768;         // buf_cap x {u32 score bits, u32 (query<<26)|row}
};

// AUX = cache-policy bits of the slab loads (0 = default, 2 = nt: rows are read once by one CU)
template <int NW, int PF, int RING, int AUX = 0>
__global__ void __launch_bounds__(NW * 64)
scan_kernel(const ScanParams p) {
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)p.slab + (span > 0 ? wrow0 : 0) * (int64_t)ROWB), 0, (int)span, 0x00020000);

    // fill cursor: (rows of the tile being fetched -> vo[], k-step -> fill_step); it runs RING-1
    // steps ahead of the consumer. Past the last tile it keeps walking forward: those loads fall
    // out of the descriptor's bounds (return 0, no memory traffic).
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
            abuf[s][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, AUX);
        fill_advance();
        // keep issue order == ring order: hipcc's waitcnt for slot 0 is the minimum over the loop
        // entry and the back edge, so a shuffled prologue would cost ring depth on every revolution
        __builtin_amdgcn_sched_barrier(0);
    }

    // the query image is copied into LDS AFTER the first ring loads are in flight (their HBM latency
    // overlaps the 96 KiB copy from L2)
    ATLAS_SCAN_STAMP(1);        // [1] ring prologue issued
    copy_qfrag_to_lds<NW * 64>(s_q, p.qfrag, tid);
    if (tid < 64) {
        s_theta[tid] = (tid < p.nq) ? p.theta0[tid] : pos_inf();
        s_cnt[tid] = 0;
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; s_flag[2] = 0; }
    __syncthreads();
    ATLAS_SCAN_STAMP(2);        // [2] query image in LDS

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

    // Flush the LDS buffer into the per-query global lists, then compact every list that crossed
    // keep_max (k-th largest -> certified threshold -> in-place prune). Called by ALL waves between
    // two barriers, rarely: buffer half full, direct-store overflow, or end of the scan.
    auto flush_and_compact = [&](const int parity, const bool final_flush) {
        wg_barrier_lds();                       // everybody has read the request words
        const uint32_t nbuf = s_flag[2] < (uint32_t)p.buf_cap ? s_flag[2] : (uint32_t)p.buf_cap;
        for (uint32_t i = tid; i < nbuf; i += NW * 64) {
            const uint2 e = s_buf[i];
            const uint32_t qq = e.y >> 26;
            const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
            if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(e.x, gbase + (e.y & 0x03ffffffu));
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
            const float theta = prune_threshold(f32_from_order_key(prefix), p.qeps[qq]);
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
                abuf[fill][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], fill_step * 64, AUX);
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
                // Candidates go to a workgroup-wide LDS buffer: LDS traffic is counted in lgkmcnt, so the
                // ring of HBM loads (vmcnt) is not disturbed. (gfx9 counts loads and stores in one vmcnt
                // and retires them out of order with respect to each other: a single pending global store
                // would force vmcnt(0) in front of every ring slot.) Only when the buffer is full do entries
                // go straight to the global lists, and that branch drains itself.
                const uint32_t rrel = (uint32_t)row0 + (uint32_t)lgrp * 4u;
                bool spilled = false;
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                    for (int qf = 0; qf < 4; ++qf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[pf][qf][r];
                            if (v > th[qf]) {
                                const uint32_t qq = (uint32_t)(qf * 16 + lrow);
                                const uint32_t slot = atomicAdd(&s_flag[2], 1u);
                                if (slot < (uint32_t)p.buf_cap) {
                                    s_buf[slot] = make_uint2(f32_bits(v), (qq << 26) | (rrel + (uint32_t)(pf * 16 + r)));
                                    if (slot >= (uint32_t)p.flush_at) s_flag[par] = 1u;   // request a flush
                                } else {
                                    const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
                                    if (gs < (uint32_t)p.cap)
                                        my_lists[qq * qstride + gs] =
                                            make_uint2(f32_bits(v), gbase + rrel + (uint32_t)(pf * 16 + r));
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
            nrm[pf] = 0.f;
#pragma unroll
            for (int qf = 0; qf < 4; ++qf) acc[pf][qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        wg_barrier_lds();
        if (row0 == 0) ATLAS_SCAN_STAMP(3);   // [3] first tile done
#if ATLAS_TUNING
        // per-tile end stamps of this workgroup, collected in LDS (no global store in the loop) and dumped after the hand-over
        if (p.dbg && tid == 0 && row0 / TILE < 120) ((unsigned long long*)(smem + ScanSmem::buf_off + (size_t)p.buf_cap * 8))[row0 / TILE] = wall_clock64();
#endif
        // the request word of this tile's parity cannot change until every wave has passed the next
        // barrier, so all waves take the same branch
        if (s_flag[par] != 0u) flush_and_compact(par, false);
        row0 += TILE;
        par ^= 1;
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
            const uint32_t qq = e.y >> 26;
            const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
            if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(e.x, gbase + (e.y & 0x03ffffffu));
        }
        // largest row norm^2 of the workgroup (x1.001: v_dot2 accumulates in fp32): waves -> LDS -> one word
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
        float* s_pm = (float*)(smem + ScanSmem::aux_off);
        if (lane == 0) s_pm[wave] = pm;
        wg_barrier_lds();
        if (tid < 64) {
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
    if (p.dbg && tid < 120) p.dbg[2048 + blockIdx.x * 120 + tid] = (tid < ntiles) ? ((unsigned long long*)(smem + ScanSmem::buf_off + (size_t)p.buf_cap * 8))[tid] : 0ull;
#endif
}


// ------------------------------------------------------------------------------------------
// sample pre-pass: approximate scores of S evenly spread rows -> sample_scores,
// from which sample_theta_kernel derives a certified initial threshold per query. Without it
// every workgroup starts at -inf and pays a "cold start" (its first tile passes entirely).
//   grid = S/64 blocks of 256 threads; wave w of block j scores rows row(j) + 16w .. +16
// ------------------------------------------------------------------------------------------
struct SampleParams {
    const uint16_t* slab; int64_t N;
    const uint4* qfrag;
    float* top2;              // [64][S/64][2]: the two best approximate scores of each 64-row sample tile
    int S;                    // multiple of 64
    int64_t stride_rows;      // first row of sample tile j = j * stride_rows (multiple of 16, >= 64)
};

__global__ void __launch_bounds__(256)
sample_scores_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_q = (uint4*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * p.stride_rows + wave * 16;
    int64_t r = row0 + lrow;
    if (r >= p.N) r = p.N - 1;
    const uint4* src = (const uint4*)((const unsigned char*)p.slab + r * (int64_t)(D_FAST * 2) + lgrp * 16);
    // all 24 fragments of this wave's 16 rows go out first (HBM latency overlaps the query copy)
    uint4 a[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) a[s] = src[s * 4];
    copy_qfrag_to_lds<256>(s_q, p.qfrag, tid);
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) acc[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const f16x8 av = __builtin_bit_cast(f16x8, a[s]);
#pragma unroll
        for (int qf = 0; qf < 4; ++qf)
            acc[qf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                av, __builtin_bit_cast(f16x8, s_q[(s * 4 + qf) * 64 + lane]), acc[qf], 0, 0, 0);
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
    if (lgrp == 0) {
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

// One wave per query (4 queries per 256-thread block): k-th largest of the 2*nblk tile maxima written by
// sample_scores_kernel (nblk <= 256 -> 8 keys per lane), entirely in registers. All of them are scores of
// distinct slab rows, so prune_threshold(k-th) is a certified initial threshold for the scan.
__global__ void __launch_bounds__(256)
sample_theta_kernel(const float* __restrict__ top2, int nblk, int k, const float* __restrict__ qeps, int nq,
                    float* __restrict__ theta0) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= 64) return;
    if (q >= nq) { if (lane == 0) theta0[q] = pos_inf(); return; }
    const int n = nblk * 2;
    uint32_t key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = lane + u * 64;
        const float v = top2[(size_t)q * n + (i < n ? i : n - 1)];      // unconditional (clamped) loads: all in flight
        key[u] = (i < n && v > neg_inf()) ? f32_order_key(v) : 0u;       // 0 = padding, below every real key
    }
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) valid += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(key[u] != 0u));
    const uint32_t kth = wave_kth_key<8>(key, (uint32_t)k, 22);
    if (lane == 0) theta0[q] = (valid >= (uint32_t)k) ? prune_threshold(f32_from_order_key(kth), qeps[q]) : neg_inf();
}

}  // namespace atlas
