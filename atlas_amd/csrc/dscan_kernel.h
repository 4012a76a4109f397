// dscan_kernel.h -- the slab pass of up to 64 queries with the slab staged through LDS-DMA and the queries in REGISTERS (round 6).
//
// Why a third scan kernel. scan_kernel.h takes the slab HBM -> VGPR in the MFMA A-operand shape: a lane is one of 16 rows, so one wave
// instruction is 16 rows x 64 B -- HALF of sixteen 128-byte lines -- and the second half follows in the next instruction. Measured on this pool
// (tools/read_ceiling.hip, profiles/r06/read_ceiling_32m.txt): a read-only stream in that shape gives 6.0-6.5 TB/s with the default cache
// policy and LESS with `nt` (the two halves of a line go to the L2 as two requests); a stream whose every wave instruction covers FULL lines
// gives 6.3-6.5 with the default policy and 6.6-6.9 TB/s with `nt`. The operand layout cannot take full lines from a register load (four lanes
// per row), a transposition in registers costs the VALU slots the stream does not have -- but LDS-DMA can: `buffer_load_dwordx4 ... lds`
// writes 8 rows x 128 B per wave instruction, swizzled on the source address (as gscan_kernel.h), and the fragments come back with one
// conflict-free ds_read_b128 each. What made room for 128 KiB of stages: the 98 KiB query image is gone. Wave w owns the 16 queries of fragment
// w & 3 for the WHOLE k = 768 in 96 VGPRs (the MFMA B operand of every k-step) and the slab rows 128 (w >> 2) .. + 128 of the workgroup's
// 256-row tile: per 32 KiB stage (one 64-wide k-tile of 256 rows) a wave issues 4 DMA pieces, 16 ds_read_b128 and 16 MFMAs, and not one VALU
// instruction. tools/dscan_proto.hip (the bare k-loop + filter): 6.88 ms at 32M rows against 8.08 ms of scan_kernel<16,1,8,64> on the same box.
//
// The static part of the slab is DEALT to the workgroups tile by tile (workgroup g: tiles g, g + G, ...): all of them stream one ~100 MB window that
// moves through the slab -- on boxes whose large allocations stream slower past their first ~24 GB this is worth 3-7 % at 32M rows (see `static_end` below).
//
// Everything around the k-loop is scan_kernel.h's, restated for 8 waves: ScanParams, the candidate entries and per-(query, workgroup) lists
// the merge gathers, the coop threshold exchange on the first tiles, flush + compaction, the run-time tile pool at the end of the slab, the
// hand-over without global atomics, the certifying twin (Gram MFMAs, two of a tile half's eight fragments per wave). The host runs it for the
// coop, unpaired, 64-query passes (every search of up to 64 queries on a shard of >= 65 536 rows); 96-query and paired passes and shards too
// small for the coop exchange stay on scan_kernel.h.
//
// Counted waits. A wave's vector-memory stream in the k-loop is its DMA pieces, four per stage, and they land in order: `s_waitcnt vmcnt(8)` in
// front of a stage's barrier = this wave's pieces of that stage have landed, the two younger stages may fly. Stores (candidate flushes, granules)
// and the one returning atomic of the tile pool only ever make such a wait stricter. Every LDS access of the steady state is inline asm: behind an
// LDS-DMA it cannot prove disjoint, hipcc's wait insertion drains vmcnt(0) in front of any LDS instruction it sees.
#pragma once
#include "scan_kernel.h"

namespace atlas {

#define DS_TILE 256               // slab rows of a workgroup tile
#define DS_NKT (D_FAST / 64)      // 12 k-tiles (stages) per tile
#define DS_STG (256 * 128)        // bytes of a stage: 256 rows x 128 B
#define DS_NSTAGE 4               // stages in LDS: one being read, three in flight (96 KiB per CU)
#define DS_NW 8
#define DS_BUF_CAP 3968           // candidate buffer entries (31 KiB)

struct DScanSmem {                // byte offsets into dynamic LDS (the stages first: 128-byte aligned, `^ 64` addresses k-step 1)
    static constexpr int theta_off = DS_NSTAGE * DS_STG;          // 64 f32
    static constexpr int cnt_off = theta_off + 256;               // 64 u32
    static constexpr int flag_off = cnt_off + 256;                // [0], [1] flush request by tile parity, [2] buffer fill, [4] posted pool ticket
    static constexpr int aux_off = flag_off + 64;                 // [0, 64) per-wave norm maxima, [64, 320) per-query eps
    static constexpr int buf_off = aux_off + 320;                 // DS_BUF_CAP x {u32 score bits, u32 (query << 26) | virtual row}
    static constexpr int total = buf_off + DS_BUF_CAP * 8;
};
static_assert(DScanSmem::total <= 160 * 1024, "one workgroup per CU: all of its LDS");

typedef __attribute__((address_space(3))) void* ds_lds_ptr;
// LDS accesses the wait-insertion pass does not see (see the header): addresses are 32-bit LDS byte addresses
static __device__ __forceinline__ uint32_t ds_ld32(const uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
static __device__ __forceinline__ void ds_st32(const uint32_t addr, const uint32_t v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
static __device__ __forceinline__ void ds_st64(const uint32_t addr, const uint32_t lo, const uint32_t hi) {
    const unsigned long long e = (unsigned long long)lo | ((unsigned long long)hi << 32);
    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(e) : "memory");
}
static __device__ __forceinline__ uint32_t ds_add_rtn(const uint32_t addr, const uint32_t v) {
    uint32_t r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
template <int OFF>
static __device__ __forceinline__ void ds_rd128(u32x4& dst, const uint32_t addr) {       // issued, NOT waited for
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// Certified first threshold of one query (one wave): the k-th largest of n <= 64 * KPL sample scores of DISTINCT rows, pushed through prune_threshold
// (scan_kernel.h: initial_theta, which takes one maximum per workgroup; here every workgroup brings TWO scores per query, see first_tile_exchange)
template <int KPL>
static __device__ __forceinline__ float sample_theta(const uint32_t (&key)[KPL], const int k, const float eps) {      // key = f32_order_key(score); 0 = no score (padding, below every real key)
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < KPL; ++u) valid += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(key[u] != 0u));
    const uint32_t kth = wave_kth_key<KPL>(key, (uint32_t)k, 14);
    return (valid >= (uint32_t)k) ? prune_threshold(f32_from_order_key(kth), eps) : neg_inf();
}

// AUX & 31 = cache-policy bits of the slab DMA (2 = nt: every line is read once, by one CU); AUX & 64 = the caller's pmax is certified
// (ATLAS_SCAN_TRUST_PMAX): no row norms are measured
template <int AUX>
__global__ void __launch_bounds__(DS_NW * 64)
dscan_kernel(const ScanParams p) {
    constexpr int NW = DS_NW, NQ = QCHUNK, QSHIFT = 26;
    constexpr uint32_t ROWMASK = (1u << QSHIFT) - 1u;
    constexpr bool CERT = !(AUX & 64);
    constexpr int POLICY = AUX & 31;
    constexpr int ROWB = D_FAST * 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    float* s_theta = (float*)(smem + DScanSmem::theta_off);
    uint32_t* s_cnt = (uint32_t*)(smem + DScanSmem::cnt_off);
    uint32_t* s_flag = (uint32_t*)(smem + DScanSmem::flag_off);
    float* s_eps = (float*)(smem + DScanSmem::aux_off + 64);
    uint2* s_buf = (uint2*)(smem + DScanSmem::buf_off);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qf = wave & 3, half = wave >> 2;
    const int lr = lane & 15, lg = lane >> 4;
    const uint32_t G = gridDim.x;

    // The static part of the slab, [0, pool_begin) (pool_begin == N without a pool), is DEALT to the workgroups tile by tile: workgroup g takes the tiles
    // g, g + G, g + 2 G, ... -- at any moment the G workgroups stream ONE window of G x 384 KiB = ~100 MB that moves through the slab, instead of G
    // streams ~190 MB apart (scan_kernel.h's contiguous ranges). Measured on the bare k-loop (tools/dscan_proto.hip -DDEAL=1, profiles/r06/
    // dscan_proto_dealt_tiles.txt): 32M rows 7.40 -> 6.87, 7.63 -> 7.07, 7.48 -> 7.23 ms on a box whose large allocations stream slower past their first
    // ~24 GB (scan_halves.txt: the translation reach, not the DRAM); nothing lost at 4M and 1M rows. The sample of the coop exchange -- every workgroup's
    // first tile -- is therefore the slab's first G tiles, not G tiles spread over it: any k real scores give a valid threshold.
    // (p.deal == 0, the tuning build's A/B: scan_kernel.h's split -- workgroup g owns rows [g * rows_per_wg, + rows_per_wg) of the static part)
    const bool deal = p.deal != 0;
    const int64_t r_begin = deal ? 0 : (int64_t)blockIdx.x * p.rows_per_wg;
    const int64_t static_end = deal ? p.pool_begin : (r_begin + p.rows_per_wg < p.pool_begin ? r_begin + p.rows_per_wg : p.pool_begin);
    const int n_static = static_end > r_begin ? (int)((static_end - r_begin + DS_TILE - 1) / DS_TILE) : 0;
    const int ntiles = !deal ? n_static : n_static > (int)blockIdx.x ? (n_static - (int)blockIdx.x + (int)G - 1) / (int)G : 0;       // workgroup-uniform
    const size_t qstride = (size_t)gridDim.x * p.cap;
    uint2* my_lists = p.lists + (size_t)blockIdx.x * p.cap;
    // candidate entries carry a 26-bit VIRTUAL row: 256 c + r = row r of the workgroup's c-th static tile (below rows_per_wg: with a pool every workgroup
    // has exactly rows_per_wg / 256 of them), from vpool on = pool rows (scan_kernel.h); without a pool nothing reaches vpool
    const uint32_t vpool = p.pool_tiles > 0 ? (uint32_t)p.rows_per_wg : (1u << QSHIFT);
    const uint32_t pbase = (uint32_t)p.pool_begin - vpool;
    const uint32_t gbase = (uint32_t)r_begin;
    auto global_row = [&](const uint32_t v) -> uint32_t {
        return v >= vpool ? v + pbase : deal ? ((((v >> 8) * G + blockIdx.x) << 8) | (v & 255u)) : v + gbase;
    };

    // ---- the tile sequence: the workgroup's dealt static tiles 0 .. ntiles - 1, pool tile blockIdx.x, then pool tiles G + ticket (one returning atomic per tile) ----
    struct Tile { int64_t row0; int rem; uint32_t vrow0; };          // first slab row, rows (0: no tile), first virtual row
    auto static_tile = [&](const int c) -> Tile {
        const int64_t r0 = deal ? ((int64_t)c * G + blockIdx.x) * DS_TILE : r_begin + (int64_t)c * DS_TILE;
        int64_t rem = static_end - r0;
        if (rem > DS_TILE) rem = DS_TILE;
        return Tile{r0, (int)(rem > 0 ? rem : 0), (uint32_t)c * DS_TILE};
    };
    // (pool tiles are SHORTER than static ones -- p.pool_tile_rows of the 256 rows: the rows past a tile's end are not fetched, so it costs its share
    //  of a tile's HBM time, and the workgroups' finishing times differ by a short tile, not a whole one)
    auto pool_tile = [&](const uint32_t pt) -> Tile {
        if (pt >= (uint32_t)p.pool_tiles) return Tile{0, 0, 0u};
        const uint32_t ptr = (uint32_t)p.pool_tile_rows;
        int rem = p.pool_rows - (int)(pt * ptr);
        if (rem > (int)ptr) rem = (int)ptr;
        return Tile{p.pool_begin + (int64_t)pt * ptr, rem > 0 ? rem : 0, vpool + pt * ptr};
    };
    auto seq_tile = [&](const int c) -> Tile { return c < ntiles ? static_tile(c) : pool_tile(blockIdx.x); };     // (c <= ntiles)

    // ---- LDS-DMA: this wave's 4 pieces of k-tile kt of tile t into stage `buf`: LDS rows 32 wave + 8 i + (lane >> 3), 128 B each; LDS[row][16-byte
    // position pos] holds source chunk pos ^ (row & 7). Rows at or past the end of the tile are not fetched (zeros land in LDS) ----
    const uint32_t vbase = (uint32_t)(lane >> 3) * ROWB + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    auto issue = [&](const Tile& t, const int kt, const int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slab + (size_t)(t.rem > 0 ? t.row0 : 0) * D_FAST), 0, t.rem * ROWB, 0x00020000);
        unsigned char* const ls = smem + buf * DS_STG + wave * 4096;
        uint32_t vs = vbase + (uint32_t)(wave * 32 * ROWB);
        asm volatile("" : "+v"(vs));                    // (formed here: hoisted, it lives across the k-loop)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ds_lds_ptr)(ls + i * 1024), 16, (int)(vs + (uint32_t)(i * 8 * ROWB)), kt * 128, 0, POLICY);
    };

    // per-call LDS state FIRST, with plain stores: behind the first LDS-DMA every LDS access hipcc can see costs a vmcnt(0) (header). What reads
    // it lies a dozen stage barriers on (or, for a workgroup without rows, behind first_tile_exchange's own barrier)
    if (tid < NQ) { s_cnt[tid] = 0; s_theta[tid] = tid < p.nq ? neg_inf() : pos_inf(); }
    if (tid < 16) s_flag[tid] = 0;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* gran = (gu64*)p.theta_gran;
    const uint32_t tag = *p.epoch + 1u;                               // (uniform: a scalar load)
    if (blockIdx.x == 0 && p.q0 == 0 && tid < ATLAS_STATUS_HEADER) p.out_status[tid] = 0;   // the merge accumulates into it

    Tile cur = seq_tile(0), nxt = seq_tile(1);
    if (ntiles == 0) { cur.rem = 0; nxt.rem = 0; }
    // the first three stages go out before the queries are asked for: 96 KiB per CU land while those are loaded and converted (a wave's loads
    // return in order: the first batch of query loads waits for the pieces, ~4 us of cold HBM, the later ones are trips to the L2)
    if (ntiles > 0) {
#pragma unroll
        for (int s = 0; s < DS_NSTAGE - 1; ++s) issue(cur, s, s);
    }

    // ---- the wave's 16 queries -> registers (MFMA B layout: lane = (query 16 qf + lr, k-group lg), k-step s = elements 32 s + 8 lg .. + 8),
    // converted RNE to fp16 = `.half()` (src/index.py:117); rows >= nq are zero ----
    u32x4 bq[KSTEPS];
    {
        const int qi = 16 * qf + lr;
        const size_t base = (size_t)(p.q0 + (qi < p.nq ? qi : 0)) * D_FAST + (size_t)lg * 8;
        constexpr int BATCH = 6;
#pragma unroll
        for (int b0 = 0; b0 < KSTEPS; b0 += BATCH) {
            QRaw raw[BATCH];
            if (p.q_dtype == ATLAS_DT_F32) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) { const uint4* s = (const uint4*)((const float*)p.q + base + (size_t)(b0 + u) * 32); raw[u].a = s[0]; raw[u].b = s[1]; }
            } else {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) { raw[u].a = *(const uint4*)((const uint16_t*)p.q + base + (size_t)(b0 + u) * 32); raw[u].b = make_uint4(0, 0, 0, 0); }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const uint4 h = p.q_dtype == ATLAS_DT_F32 ? q8_to_f16(raw[u], ATLAS_DT_F32) : p.q_dtype == ATLAS_DT_F16 ? raw[u].a : q8_to_f16(raw[u], ATLAS_DT_BF16);
                bq[b0 + u] = qi < p.nq ? (u32x4){h.x, h.y, h.z, h.w} : (u32x4){0u, 0u, 0u, 0u};
            }
        }
        // eps of the lane's query: fp32 sum of squares of the fp16 values (relative error <= 768 * 2^-24: inside query_eps' margin)
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const f16x2 h = __builtin_bit_cast(f16x2, (uint32_t)bq[s][i]); ss = __builtin_amdgcn_fdot2(h, h, ss, false); }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (half == 0 && lg == 0) ds_st32(lds0 + DScanSmem::aux_off + 64 + (uint32_t)qi * 4u, f32_bits(qi < p.nq ? query_eps(ss, p.pmax) : 0.f));
    }

    float pm = 0.0f;                                                  // CERT: largest row sum of squares this lane has seen
    int par = 0;

    // ---- flush the LDS buffer into the per-query global lists, then compact every list that crossed keep_max (scan_kernel.h: flush_and_compact).
    // Called by ALL waves, rarely; the global stores and the read-back drain this wave's DMA pieces as well (vmcnt(0)): every later counted
    // wait only gets stricter ----
    auto flush_and_compact = [&](const int parity, const bool final_flush) {
        wg_barrier_lds();                       // everybody has read the request words
        const uint32_t nbuf = s_flag[2] < (uint32_t)DS_BUF_CAP ? s_flag[2] : (uint32_t)DS_BUF_CAP;
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
            if (n <= (uint32_t)(final_flush ? p.keep_max : p.k)) continue;
            uint2* L = my_lists + (size_t)qq * qstride;
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
                    p.qflag[qq] = 1u;           // candidate band wider than the list (mass ties): exact path
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

    // ---- coop: the scan's own first tiles are the sample (scan_kernel.h: first_tile_exchange) -- with TWO scores per (workgroup, query) instead of one: the
    // best and the runner-up of the tile (t1 >= t2 come in per wave: the two best of the lane's query among the wave's 128 rows). scan_kernel.h publishes the
    // tile's maximum only, and the k-th largest of G = 256 maxima is a weak bound when k is large -- for k = 256 it is the SMALLEST tile maximum: 4.4M
    // candidates per search at 4M rows (1.7 % of all scores), 1.49 ms instead of 0.96; the k-th largest of 2 G = 512 scores of distinct rows sits near the
    // sample's own k-th best up to k = 256 (profiles/r06/dscan_k_sweep.txt). Granules: gran_max[q][2 G] (inside the [96][1024] the workspace holds) ----
    auto first_tile_exchange = [&](const float t1, const float t2) {
        float* s_tmax = (float*)s_buf;                     // [2 halves][2][64]: the candidate buffer is still empty
        gu64* gmax = (gu64*)p.gran_max;
        const uint32_t G2 = 2u * G;
        if (lg == 0) { s_tmax[(half * 2 + 0) * NQ + 16 * qf + lr] = t1; s_tmax[(half * 2 + 1) * NQ + 16 * qf + lr] = t2; }
        wg_barrier_lds();
        if (tid < 2 * p.nq) {
            const int qq = tid >> 1, j = tid & 1;
            const float a1 = s_tmax[0 * NQ + qq], a2 = s_tmax[1 * NQ + qq], b1 = s_tmax[2 * NQ + qq], b2 = s_tmax[3 * NQ + qq];
            const float m = j == 0 ? fmaxf(a1, b1) : fmaxf(fminf(a1, b1), fmaxf(a2, b2));      // the tile's best | its runner-up
            __hip_atomic_store(gmax + (size_t)qq * G2 + 2u * blockIdx.x + (uint32_t)j, ((unsigned long long)tag << 32) | (unsigned long long)f32_bits(m),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == NW - 1) {
            if ((int)blockIdx.x < p.nq) {                  // this workgroup derives the threshold of query blockIdx.x
                const int qq = blockIdx.x;
                constexpr int KPL = 8;                     // 2 x 256 granules over 64 lanes; a granule becomes its 32-bit order key as it arrives (the
                uint32_t key[KPL];                         //  wave holds its tile's 32 accumulators and 96 query registers meanwhile)
                uint32_t have = 0u;                        // bit u: granule u has arrived, or lies past the 2 G of this launch
#pragma unroll
                for (int u = 0; u < KPL; ++u) { key[u] = 0u; if ((uint32_t)lane + 64u * u >= G2) have |= 1u << u; }
                for (const unsigned long long spin_end = wall_clock64() + ATLAS_SPIN_TICKS; ; ) {
                    bool missing = false;
#pragma unroll
                    for (int u = 0; u < KPL; ++u) {
                        if ((have >> u) & 1u) continue;
                        const unsigned long long g = __hip_atomic_load(gmax + (size_t)qq * G2 + ((uint32_t)lane + 64u * u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(g >> 32) == tag) {
                            have |= 1u << u;
                            const float val = bits_f32((uint32_t)g);
                            key[u] = val > neg_inf() ? f32_order_key(val) : 0u;
                        } else missing = true;
                    }
                    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                    if (wall_clock64() >= spin_end) break;          // (what has not arrived in time stays 0: a looser bound, never a wrong one)
                    __builtin_amdgcn_s_sleep(2);
                }
                const float th = sample_theta<KPL>(key, p.k, s_eps[qq]);
                if (lane == 0) __hip_atomic_store(gran + qq, ((unsigned long long)tag << 32) | (unsigned long long)f32_bits(th), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long g = 0ull;
            for (const unsigned long long spin_end = wall_clock64() + ATLAS_SPIN_TICKS; ; ) {
                bool missing = false;
                if (lane < p.nq && (uint32_t)(g >> 32) != tag) {
                    g = __hip_atomic_load(gran + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    missing = (uint32_t)(g >> 32) != tag;
                }
                if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                if (wall_clock64() >= spin_end) break;
                __builtin_amdgcn_s_sleep(2);
            }
            s_theta[lane] = lane < p.nq ? ((uint32_t)(g >> 32) == tag ? bits_f32((uint32_t)g) : neg_inf()) : pos_inf();
        }
        wg_barrier_lds();
    };
    if (ntiles == 0) first_tile_exchange(neg_inf(), neg_inf());       // a workgroup without rows still owes the others its (empty) scores

    // the lane's fragment address in stage 0: slab row half * 128 + 16 a + lr, chunk lg of k-step 0 (k-step 1: ^ 64)
    const uint32_t as0 = lds0 + (uint32_t)((half * 128 + lr) * 128 + ((lg ^ (lr & 7)) * 16));
    const uint32_t ag0 = as0 + (uint32_t)(qf * 2048);      // CERT: the two fragments whose row norms this wave measures, qf and qf + 4 of its half
    const uint32_t a_theta = lds0 + DScanSmem::theta_off + (uint32_t)(16 * qf + lr) * 4u;
    const uint32_t a_flag = lds0 + DScanSmem::flag_off;
    const uint32_t a_buf = lds0 + DScanSmem::buf_off;

    f32x4 acc[8];
    f32x4 gram[2];
    uint32_t tk = 0;                                       // wave 0, lane 0: the pool ticket drawn at kt == 0
    int c_seq = 0;                                         // tile the workgroup is in
    bool tk_drawn = false;                                 // tile c_seq + 1 comes from a ticket (workgroup-uniform)

    if (ntiles > 0)
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int kt = 0; kt < DS_NKT; ++kt) {
            const int buf = kt % DS_NSTAGE;                // (DS_NKT % DS_NSTAGE == 0: a k-tile's stage is a compile-time constant)
            // this wave's pieces of the stage about to be read have landed (the two younger stages may fly) ...
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            // ... wave 0 posts the ticket it drew two stages ago (a plain returning atomic: hipcc waits for it right here, by count -- it is
            // older than the eight pieces issued since)
            if (kt == 2 && tk_drawn && wave == 0 && lane == 0) ds_st32(a_flag + 16, tk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the stage about to be overwritten have returned
            __builtin_amdgcn_s_barrier();
            if (kt == 0) {
                // the candidate buffer crossed its flush mark during the previous tile (the request word of that tile's parity; every wave
                // reads the same value: it cannot change before all waves have passed this barrier again)
                if (c_seq > 0 && ds_ld32(a_flag + 4u * (uint32_t)(par ^ 1)) != 0u) flush_and_compact(par ^ 1, false);
                // tile c_seq + 1: static, the pre-assigned pool tile, or -- beyond those -- a ticket, drawn now and read at kt == 2
                tk_drawn = c_seq + 1 > ntiles;
                if (!tk_drawn) nxt = seq_tile(c_seq + 1);
                else if (wave == 0 && lane == 0) tk = atomicAdd(p.ticket, 1u);
            }
            if (kt == 2 && tk_drawn) nxt = pool_tile((uint32_t)__builtin_amdgcn_readfirstlane((int)(G + ds_ld32(a_flag + 16))));
            {   // refill the stage read in the previous iteration with k-tile kt + 3 (of the next tile from kt == 9 on)
                const int k2 = kt + DS_NSTAGE - 1;
                if (k2 < DS_NKT) issue(cur, k2, k2 % DS_NSTAGE); else issue(nxt, k2 - DS_NKT, k2 % DS_NSTAGE);
            }
            const uint32_t s0 = as0 + buf * DS_STG, s1 = s0 ^ 64u;
            // k-step 0, then k-step 1 into the registers the first eight MFMAs have released (the matrix pipe is ~20 % busy in this kernel: a
            // stage is 32 KiB of HBM time, ~2 500 cycles, against ~1 000 of reads and MFMAs -- registers, not overlap, are what is short: 96 hold
            // the queries). CERT: the wave reads ITS two norm fragments once more (which two is a run-time address, not a register select)
            {
                u32x4 f0[8];
                ds_rd128<0 * 2048>(f0[0], s0); ds_rd128<1 * 2048>(f0[1], s0); ds_rd128<2 * 2048>(f0[2], s0); ds_rd128<3 * 2048>(f0[3], s0);
                ds_rd128<4 * 2048>(f0[4], s0); ds_rd128<5 * 2048>(f0[5], s0); ds_rd128<6 * 2048>(f0[6], s0); ds_rd128<7 * 2048>(f0[7], s0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f0[4]), "+v"(f0[5]), "+v"(f0[6]), "+v"(f0[7]) :: "memory");
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f0[a]), __builtin_bit_cast(f16x8, bq[2 * kt]),
                                                                   kt == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[a], 0, 0, 0);
            }
            {
                u32x4 f1[8];
                ds_rd128<0 * 2048>(f1[0], s1); ds_rd128<1 * 2048>(f1[1], s1); ds_rd128<2 * 2048>(f1[2], s1); ds_rd128<3 * 2048>(f1[3], s1);
                ds_rd128<4 * 2048>(f1[4], s1); ds_rd128<5 * 2048>(f1[5], s1); ds_rd128<6 * 2048>(f1[6], s1); ds_rd128<7 * 2048>(f1[7], s1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]), "+v"(f1[4]), "+v"(f1[5]), "+v"(f1[6]), "+v"(f1[7]) :: "memory");
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f1[a]), __builtin_bit_cast(f16x8, bq[2 * kt + 1]), acc[a], 0, 0, 0);
            }
            if constexpr (CERT) {
                // row sums of squares: the diagonal of the fragment's Gram matrix (A and B of v_mfma_16x16x32 share a register layout: D = A A^T)
                const uint32_t sg0 = ag0 + buf * DS_STG, sg1 = sg0 ^ 64u;
                u32x4 g0[2], g1[2];
                ds_rd128<0>(g0[0], sg0); ds_rd128<4 * 2048>(g0[1], sg0); ds_rd128<0>(g1[0], sg1); ds_rd128<4 * 2048>(g1[1], sg1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0[0]), "+v"(g0[1]), "+v"(g1[0]), "+v"(g1[1]) :: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gram[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, g0[j]), __builtin_bit_cast(f16x8, g0[j]),
                                                                    kt == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gram[j], 0, 0, 0);
                    gram[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, g1[j]), __builtin_bit_cast(f16x8, g1[j]), gram[j], 0, 0, 0);
                }
            }
        }

        // ------------------------- end of a tile --------------------------------
        // accumulator element (a, r) of the lane = slab row half * 128 + 16 a + 4 lg + r of the tile, query 16 qf + lr
        int rbase = half * 128 + 4 * lg;
        asm volatile("" : "+v"(rbase));                 // (formed HERE: hipcc otherwise hoists the 32 row-validity compares of a partial tile into the k-loop -- 64 SGPRs)
        const bool full = cur.rem == DS_TILE;              // workgroup-uniform: every row of the tile exists
        if constexpr (CERT) {
            // lane (column lr, rows 4 lg + r) of a Gram matrix holds the diagonal element of row lr iff lg == lr >> 2 (rows past the end are zeros)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 gm = gram[j];
                const int r = lr & 3;
                const float x = r == 0 ? gm[0] : r == 1 ? gm[1] : r == 2 ? gm[2] : gm[3];
                pm = fmaxf(pm, lg == (lr >> 2) ? x : 0.f);
            }
        }
        // the best score of the lane's query among its 32 rows of the tile (a maximum first, as gscan_kernel.h: one compare per lane decides whether
        // the 32 elements are looked at one by one)
        float m = neg_inf();
        if (full) {
#pragma unroll
            for (int a = 0; a < 8; ++a) m = fmaxf(m, fmaxf(fmaxf(acc[a][0], acc[a][1]), fmaxf(acc[a][2], acc[a][3])));
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, rbase + 16 * a + r < cur.rem ? acc[a][r] : neg_inf());
        }
        if (c_seq == 0) {                                  // workgroup-uniform: the first tile is the sample
            // the best and the runner-up of the lane's query among the wave's 128 rows: over the lane's 32 elements, then over the four lanes of the query
            float t1 = neg_inf(), t2 = neg_inf();
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = rbase + 16 * a + r < cur.rem ? acc[a][r] : neg_inf();
                    t2 = fmaxf(t2, fminf(t1, x));
                    t1 = fmaxf(t1, x);
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float b1 = __shfl_xor(t1, o), b2 = __shfl_xor(t2, o);
                t2 = fmaxf(fminf(t1, b1), fmaxf(t2, b2));
                t1 = fmaxf(t1, b1);
            }
            first_tile_exchange(t1, t2);
        }
        {
            const float th = bits_f32(ds_ld32(a_theta));   // the lane OWNS its query: one threshold
            if (__builtin_amdgcn_ballot_w64(m > th) != 0ull) {
                // candidates go to the workgroup-wide LDS buffer (slots from one LDS counter); a full buffer sends them straight to the global lists
                uint32_t lr_o = (uint32_t)lr;
                asm volatile("" : "+v"(lr_o));             // (what depends on the lane's query is formed here, not hoisted over the k-loop)
                const uint32_t qq = (uint32_t)(16 * qf) + lr_o;
                const uint32_t vr0 = cur.vrow0 + (uint32_t)rbase;
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[a][r];
                        if ((rbase + 16 * a + r < cur.rem) && v > th) {
                            const uint32_t vrow = vr0 + (uint32_t)(16 * a + r);
                            const uint32_t slot = ds_add_rtn(a_flag + 8, 1u);
                            if (slot < (uint32_t)DS_BUF_CAP) {
                                ds_st64(a_buf + slot * 8u, f32_bits(v), (qq << QSHIFT) | vrow);
                                if (slot >= (uint32_t)p.flush_at) ds_st32(a_flag + 4u * (uint32_t)par, 1u);      // request a flush
                            } else {
                                const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
                                if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(f32_bits(v), global_row(vrow));
                                ds_st32(a_flag + 4u * (uint32_t)par, 1u);
                            }
                        }
                    }
            }
        }
        par ^= 1;
        ++c_seq;
        cur = nxt;
        if (cur.rem == 0) break;                           // workgroup-uniform: every wave formed the same next tile
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (pieces past the last tile: out of bounds, zeros, no traffic -- but counted)

    // ---- final hand-over, without global atomics (scan_kernel.h): buffered candidates -> this workgroup's own lists, the 64 list lengths and the
    // norm / flag word as plain stores; the merge kernel gathers the G segments of its query ----
    {
        wg_barrier_lds();
        const uint32_t nbuf = s_flag[2] < (uint32_t)DS_BUF_CAP ? s_flag[2] : (uint32_t)DS_BUF_CAP;
        for (uint32_t i = tid; i < nbuf; i += NW * 64) {
            const uint2 e = s_buf[i];
            const uint32_t qq = e.y >> QSHIFT;
            const uint32_t gs = atomicAdd(&s_cnt[qq], 1u);
            if (gs < (uint32_t)p.cap) my_lists[qq * qstride + gs] = make_uint2(e.x, global_row(e.y & ROWMASK));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
        float* s_pm = (float*)(smem + DScanSmem::aux_off);
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
            m *= 1.001f;                                   // (the Gram MFMA accumulates in fp32)
            p.wg_stat[(size_t)blockIdx.x * 2 + 0] = f32_bits(m);
            p.wg_stat[(size_t)blockIdx.x * 2 + 1] = (m > p.pmax2_hint) ? (uint32_t)ATLAS_F_PMAX_VIOLATION : 0u;
        }
    }
}

}  // namespace atlas
