// gscan_kernel.h -- the slab scan for BIG query batches (a rank of an N-GPU search scores ALL gathered queries, src/index.py:127-131:
// 512 per call at W = 8 x 64), shaped like a GEMM: scores[256 slab rows x 256 queries] per workgroup tile, both operands staged through LDS
// by LDS-DMA, accumulators in registers for the whole k = 768, and the threshold filter of scan_kernel.h as the epilogue. Scores never
// reach HBM here either.
//
// Why a second kernel: scan_kernel.h streams -- one 16-row fragment per wave, the query operand re-read from an LDS image for every MFMA
// (one ds_read_b128 per MFMA) -- which is right while the slab bytes are the bound (<= 96 queries per pass: 64-96 flop per byte against a
// ridge of ~312), and wrong above it: 512 queries on a 4M-row shard ran at 0.216 of the f16 MFMA peak. Here a wave owns a
// 128-row x 64-query block: 24 ds_read_b128 feed 64 MFMAs per 64-wide k-tile.
//
// Decomposition. One workgroup per CU (all 160 KiB of LDS), 8 waves as 2 (slab rows) x 4 (queries); the two waves of a SIMD run in
// opposite phases (one reads its fragments and issues LDS-DMA while the other multiplies: the ping-pong schedule of encoder.hip's bulk
// GEMM). Workgroup g: XCD x = g & 7, slot s = g >> 3; query column tile c = s % ncol, row range (x, s / ncol): a contiguous, tile-aligned
// share of the slab. The ncol workgroups that score the same rows against different query tiles sit on the SAME XCD in neighbouring slots
// and walk their range in step, so every slab tile comes from HBM once and from that XCD's L2 (or the Infinity Cache) the other ncol - 1
// times: HBM traffic is one slab read per launch whatever the batch (PMC: 1.12 x the slab, profiles/r04/gscan_fetch_size.txt).
// A column tile is 256, 192 or 128 queries wide (FB = 4 | 3 | 2 query fragments per wave): a pass takes the narrowest of 128, 192, 256,
// 2 x 192, 2 x 256, 4 x 256 that holds its queries (atlas_hip.hip: GS_WIDTH, with the measured cost of each).
//
// The k-loop (round 4's second version; `stage` and the loop have the details). Steady state: ~2 150 cycles per k-tile against the 2 048 the
// 2 x 64 MFMAs of a SIMD's two waves take (shader-cycle stamps, tools/gscan_phases.py); what got it there from ~3 400:
//   * EVERY LDS-DMA piece is issued from a READ phase, between the phase's fragment reads (single ds_read_b128 asm statements): a piece holds
//     the wave's instruction issue until the CU's vector-memory front end takes it, which is free beside the partner group's MFMAs and
//     beside reads that are in flight anyway -- and ~500 cycles of an idle matrix pipe in front of a group's own MFMAs (the first version:
//     group B's query pieces). For that, each group refills the slab half the OTHER group reads, and each wave its own SIMD's query rows,
//     so that no piece waits for a barrier; landing is checked with counted s_waitcnt vmcnt;
//   * the filter epilogue looks at COLUMN maxima first (16 v_max3 per query column of the wave), fragments only inside a hit column, with
//     the fragment bodies out of line: ~750 cycles for a tile without a candidate instead of ~2 700;
//   * group A reads the next tile's first k-tile BEFORE its epilogue: the phases stay aligned across tile boundaries (no extra barrier, no
//     deferred pieces); a tile boundary costs ~2 200 cycles (of ~28 000 per tile) instead of ~6 000.
// The shader clock under this kernel is 1.69 GHz (both clocks stamped at the ends of workgroup 0), not the 2.4 GHz the 2.5 PFLOP/s peak is
// quoted at: 1 024 queries x 4M rows run at 0.48 of that peak = 0.68 of what the matrix pipe can do at the clock it is given.
//
// Thresholds. A first launch of the same kernel in SAMPLE mode scores s_tiles evenly spread tiles (about 1/64 of the slab) and leaves, per
// query, the maximum of every 16-row fragment; gtheta_kernel takes a lower bound of the k-th largest of those maxima (scores of DISTINCT rows ->
// prune_threshold gives a certified threshold). The scan then runs as TWO launches: the first eighth of every row range with
// those thresholds, then -- gtheta_kernel again, over the candidates the first launch left in the lists -- the rest with the k-th best of
// N / 8 rows: ~650 candidates per query reach the merge at k = 40 whatever the shard's size. A wave appends what passes to its own LDS buffer
// (slots from ballot prefix counts: no atomics, no barrier) and empties it into the per-query global lists when it is half full (one returning
// atomic per entry on gcnt[query], all lanes at once) -- a few times per millisecond.
// The merge is merge_rescore_kernel in FLAT mode (merge_kernel.h): one contiguous list per query, cut into 1024 virtual segments.
//
// Measured and NOT adopted (profiles/r04/): a THIRD slab stage (the wave candidate buffers moved to global memory to make room: 5 x 32 KiB
// of LDS), slab pieces issued two k-tiles ahead behind a counted s_waitcnt vmcnt(8) -- 3-4 % SLOWER (three_slab_stages.patch,
// batch_gemm_pass_ab_*_three_slab_stages.txt); the slab through a register ring instead of LDS (gscan2_kernel.h, tuning build: 3-4 % slower);
// more of the pieces on group A, 13 / 14 / 16 of 16 per wave pair (slower step by step); one instruction stream for both groups with the
// piece parameters SELECTED per piece (~200 cycles of scalar code per piece: 16 % slower).
#pragma once
#include "scan_kernel.h"

namespace atlas {

#define GS_TILE 256               // slab rows per workgroup tile, and queries per column tile
#define GS_NK (D_FAST / 64)       // 12 k-tiles of 64 halfs (128 bytes of every row)
#define GS_STG (256 * 128)        // bytes of one operand stage
#define GS_WBUF_ENTRIES 448       // entries of one wave's LDS buffer (8 x 3.5 KiB behind the four stages): GS_WBUF_REAL candidates + one dummy slot per lane
#define GS_WBUF_REAL (GS_WBUF_ENTRIES - 64)
#define GS_LDS_BYTES (4 * GS_STG + 8 * GS_WBUF_ENTRIES * 8)     // 159 744
#define GTHETA_LDS(nmax) (64 + 4096 + (size_t)(nmax) * 4)
#define GTHETA_MAXKEYS 32768
#define GS_FRAG_PER_TILE 16       // 16-row fragments of a tile: the sample keeps one maximum per fragment and query

struct GScanParams {
    const uint16_t* slab;     // [N][768] fp16
    int64_t N;
    const uint16_t* q16;      // [nq][768] the queries of this pass as fp16 rows (gprep_kernel)
    int nq, ncol;             // queries of the pass, column tiles (of 64 FB queries; ncol divides gridDim.x / 8)
    int64_t rows_per_range;   // SCAN: rows of every row range (a multiple of 256)
    int tile_begin, tile_end; // SCAN: this launch takes tiles [tile_begin, tile_end) of every range (two launches per pass: the second one runs with
                              // thresholds tightened by what the first one found, gtheta_kernel)
    int s_tiles;              // SAMPLE: tiles of the sample; tile t = rows [t * s_stride, + 256), all inside the slab
    int64_t s_stride;
    const float* theta;       // SCAN: [ncol * 64 FB] pruning thresholds (+inf for the padding queries)
    float* smax;              // SAMPLE: [s_tiles * 16][ncol * 64 FB] fragment maxima
    uint2* lists;             // SCAN: [nq][gcap] {f32 bits of the approximate score, shard-local row}
    uint32_t* gcnt;           // SCAN: [nq] entries appended to lists[q] (may exceed gcap: the query is then flagged)
    uint32_t* qflag;          // SCAN: [nq] fallback flags (plain idempotent stores)
    int gcap;
    uint32_t* wg_stat;        // SCAN: [gridDim.x][2] {largest row norm^2 seen (MODE 2; 0 otherwise), ATLAS_F_* flags} of THIS launch
    float pmax2_hint;         // MODE 2: the square of the caller's pmax hint
    unsigned long long* dbg;  // tuning build only (atlas_tune_set_scan_stamps): shader-clock stamps of workgroup 0, [8 waves][GS_STAMP_ITERS][8]; null in production
};
#define GS_STAMP_FIRST 24         // the stamped iterations: k-tiles 24 .. 55 of the workgroup (its third to fifth tile)
#define GS_STAMP_ITERS 32
#if ATLAS_TUNING
#define GS_STAMP(i) do { if (MODE != 1 && p.dbg != nullptr && blockIdx.x == 0 && it >= GS_STAMP_FIRST && it < GS_STAMP_FIRST + GS_STAMP_ITERS && lane_now() == 0) \
        p.dbg[((size_t)wave * GS_STAMP_ITERS + (it - GS_STAMP_FIRST)) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
// epilogue stamps: [8 waves][8 tiles][4] behind the k-loop stamps
#define GS_ESTAMP(i) do { if (MODE != 1 && p.dbg != nullptr && blockIdx.x == 0 && ti < 8 && lane_now() == 0) \
        p.dbg[(size_t)8 * GS_STAMP_ITERS * 8 + ((size_t)wave * 8 + ti) * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define GS_STAMP(i) do { } while (0)
#define GS_ESTAMP(i) do { } while (0)
#endif

typedef unsigned int gs_u4 __attribute__((ext_vector_type(4)));
template <int B, int E, class F>
static __device__ __forceinline__ void gs_static_for(F&& f) {           // f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>)
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); gs_static_for<B + 1, E>(f); }
}
template <int OFF>
static __device__ __forceinline__ void gs_ds_read(gs_u4& dst, const uint32_t addr) {      // issued, NOT waited for
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// MODE 0 = scan (filter epilogue), 1 = sample (fragment maxima), 2 = scan that also MEASURES every row's norm (the certifying twin: the caller's
// pmax is a hint, atlas_scan_topk's default contract). The four waves that share a slab fragment row split its eight fragments, two each: 16
// v_dot2 per k-tile and wave beside its 64 MFMAs. Round 5: WHICH two is decided by the wave's LDS read addresses, not by register selects --
// wave wj reads slab fragment a ^ 2 wj into register slot a (an XOR permutation: slot a's address is one of FOUR per-wave base addresses +
// the compile-time offset (a & 1) * 2048), so every wave squares its slots 0 and 1 and they are fragments 2 wj, 2 wj + 1. Round 4 picked the two
// fragments out of the eight slots with 3 v_cndmask per v_dot2 (64 VALU per k-tile: +10-12 % on the pass, profiles/r04/gscan_certifying_twin.txt);
// the filter epilogue undoes the permutation in the row tag (one XOR).
// FB = 16-query fragments per wave: 4 -> the column tile is 256 queries wide (a wave owns 128 rows x 64 queries); 3 -> 192 wide; 2 -> 128 wide
// (128 x 32: half the MFMAs per k-tile and 48 instead of 64 LDS-DMA pieces). The narrower tiles serve the pass widths a 256-wide tile would
// leave part empty at the full cost: 97..128 and 129..192 queries, and -- two or four column tiles of 192 -- 257..384 and 513..768.
// NT = 1 (round 6): the slab pieces carry the `nt` cache policy -- for passes of ONE column tile (<= 256 queries: every slab line is read once, by one
// CU; tools/read_ceiling.hip: a full-line LDS-DMA stream gains 4-7 % with it). With two or four column tiles the neighbouring workgroups of an XCD
// re-read the tile from its L2, and nt lines are the first to leave it: those passes keep the default policy. Query pieces: always default.
template <int MODE, int FB = 4, int NT = 0>
__global__ void __launch_bounds__(512)
gscan_kernel(const GScanParams p) {
    constexpr int QW = 16 * FB;                        // queries per wave
    constexpr int CW = 4 * QW;                         // queries per column tile
    static_assert(FB >= 2 && FB <= 4, "column tiles of 128, 192 or 256 queries");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // S0 | S1 | Q0 | Q1 (32 KiB each) | 8 wave buffers
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;
    const bool grpB = wave >= 4;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int ROWB = D_FAST * 2;
    constexpr bool SCAN = MODE != 1, CERT = MODE == 2;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int col = slot % p.ncol;
    const int per_xcd = nslots / p.ncol;
    const int range = xcd * per_xcd + slot / p.ncol, nranges = 8 * per_xcd;
    int64_t begin = 0, end = 0;
    int ntl;
    if (SCAN) {
        begin = (int64_t)range * p.rows_per_range;
        end = begin + p.rows_per_range;
        if (end > p.N) end = p.N;
        ntl = end > begin ? (int)((end - begin + GS_TILE - 1) / GS_TILE) : 0;
        ntl = (ntl < p.tile_end ? ntl : p.tile_end) - p.tile_begin;
        if (ntl < 0) ntl = 0;
    } else {
        ntl = range < p.s_tiles ? (p.s_tiles - range + nranges - 1) / nranges : 0;
    }
    auto tile_row0 = [&](const int ti) -> int64_t {
        return SCAN ? begin + (int64_t)(p.tile_begin + ti) * GS_TILE : (int64_t)(range + ti * nranges) * p.s_stride;
    };
    if (SCAN && tid == 0) { p.wg_stat[(size_t)blockIdx.x * 2] = 0u; p.wg_stat[(size_t)blockIdx.x * 2 + 1] = 0u; }
    if (ntl == 0) return;
#if ATLAS_TUNING
    // (both clocks at the two ends of workgroup 0: what a 'cycle' of the stamps is in wall time)
    if (MODE != 1 && p.dbg != nullptr && blockIdx.x == 0 && tid == 0) { p.dbg[8 * GS_STAMP_ITERS * 8 + 8 * 8 * 4 + 0] = wall_clock64(); p.dbg[8 * GS_STAMP_ITERS * 8 + 8 * 8 * 4 + 1] = __builtin_readcyclecounter(); }
#endif
    const int total_it = ntl * GS_NK;

    // LDS-DMA: one wave instruction writes 8 LDS rows x 128 B, lane-linear; wave w stages LDS rows 32 w + 8 i + (lane >> 3), i = 0..3, of
    // both operands. LDS[row][16-byte position p] holds source chunk p ^ (row & 7) (the swizzle sits on the SOURCE address), which makes the
    // fragment reads below conflict-free. Everything that selects a ROW is in the bounds-checked voffset: rows past the end of the range /
    // of the queries are not fetched (zeros land in LDS); the k-tile rides in the scalar offset.
    const uint32_t chb = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    const uint32_t vbase = (uint32_t)(lane >> 3) * ROWB + chb;
    int qrows = p.nq - col * CW;
    qrows = __builtin_amdgcn_readfirstlane(qrows < 0 ? 0 : (qrows > CW ? CW : qrows));     // (hipcc clamps with v_med3: back to an SGPR, or the descriptor lives in VGPRs)
    // A k-tile is 64 pieces of 8 rows x 128 B (32 of the slab, 32 of the queries), one wave instruction each, and EVERY piece is issued from a
    // READ phase, between the phase's fragment reads: a piece holds the wave's instruction issue until the CU's one vector-memory front end
    // takes it (~95 cycles when 4 waves issue together), which costs nothing beside the partner group's MFMAs and beside ds_reads that are in
    // flight anyway -- but ~500 cycles of an idle matrix pipe in front of a group's own MFMAs, however few pieces (round 4's first version: group
    // B's 5 query pieces, queued behind group A's 44). What lets every piece go out a phase or more before its buffer is read:
    //   * slab rows 0..127 of a stage are read by group A only, rows 128..255 by group B only: each half is refilled by the OTHER group in
    //     its next read phase (A: half 1 of k-tile it + 1 while it reads k-tile it; B: half 0 of k-tile it + 2 while it reads k-tile it);
    //   * query rows wj * QW .. + QW are read by the two waves (wj, wj + 4) of one SIMD only: wave wj + 4 refills one half of them for k-tile
    //     it + 2 right behind its own reads of k-tile it (its twin read them a phase earlier), wave wj the other half a phase later.
    // NPW = 4 + QW / 16 pieces per wave and phase, 32 KiB per phase for the CU; landing time: a k-tile period or more (the slab: HBM), a
    // phase and a half for group A's query pieces (L2), all behind COUNTED s_waitcnt vmcnt (a wave's LDS-DMA loads land in order).
    constexpr int QP = QW / 16;                        // query pieces per wave and phase
    constexpr int NPW = 4 + QP;                        // pieces per wave and phase
    // (per piece: m0, one v_add, the DMA -- everything else is formed once per phase; the group is a compile-time parameter of the phase's code:
    //  a first version that SELECTED descriptor, rows and destination per piece by group spent ~200 cycles of scalar code on every piece)
    auto stage = [&](auto grp_tag, const int buf, const int it, auto&& between) __attribute__((always_inline)) {     // this wave's pieces of k-tile `it` into stage `buf`
        constexpr bool GB = decltype(grp_tag)::value;
        // (both descriptors are formed HERE, SGPR arithmetic: a descriptor carried across the k-loop ends up in VGPRs and every DMA in a
        //  readfirstlane loop)
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q16 + (size_t)col * CW * D_FAST), 0, qrows * ROWB, 0x00020000);
        const int ti = it / GS_NK, kt = it - ti * GS_NK;
        const int64_t r0 = tile_row0(ti);
        int64_t rem = (SCAN ? end : p.N) - r0;
        if (rem > GS_TILE) rem = GS_TILE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slab + (size_t)r0 * D_FAST), 0, (int)rem * ROWB, 0x00020000);
        const int kb = kt * 128;
        const int sp0 = (GB ? 0 : 16) + wj;                        // slab pieces sp0 + 4 i: the OTHER group's half, spread over the four waves
        const int qp0 = wj * (QW / 8) + (GB ? QP : 0);             // query pieces qp0 + i: this wave's half of the SIMD's QW rows
        unsigned char* const ls = smem + buf * GS_STG + sp0 * 1024;
        unsigned char* const lq = smem + 2 * GS_STG + buf * GS_STG + qp0 * 1024;
        uint32_t vs = vbase, vq = vbase;               // (from copies hipcc cannot hoist: hoisted, they live across the k-loop)
        asm volatile("" : "+v"(vs));
        vq = vs + (uint32_t)(qp0 * 8 * ROWB);
        vs += (uint32_t)(sp0 * 8 * ROWB);
        // group A: query pieces first (they are needed a phase and a half on: `vmcnt(4)` at the end of the multiply phase covers them), group B:
        // slab pieces first (its query pieces overwrite rows it reads in this very phase: they go out behind those reads)
        gs_static_for<0, NPW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr bool isq = GB ? (i >= 4) : (i < QP);
            constexpr int j = GB ? (i >= 4 ? i - 4 : i) : (i < QP ? i : i - QP);          // the j-th query / slab piece of the phase
            if constexpr (isq) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr)(lq + j * 1024), 16, (int)(vq + (uint32_t)(j * 8 * ROWB)), kb, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ls + j * 4096), 16, (int)(vs + (uint32_t)(j * 32 * ROWB)), kb, 0, NT ? 2 : 0);
            between(ic);
        });
    };

    // the lane's fragment chunks: slab rows wi * 128 + 16 a + lr (MFMA A operand), query rows wj * QW + 16 b + lr (B operand); k-step 0 of a
    // k-tile = chunks 0..3 (chunk lg of the lane), k-step 1 = chunks 4..7
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t as0 = lds0 + (wi * 128 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    // (ONE lane-dependent address lives across the k-loop: k-step 1 = chunk (4 + lg) ^ (lr & 7) = k-step 0's with bit 2 flipped, byte address
    //  ^ 64 -- the dynamic LDS segment starts at a multiple of 128 --, and the query rows sit a wave-uniform distance behind the slab rows)

    // C layout of v_mfma_f32_16x16x32_f16: lane l holds query column l & 15 and slab rows 4 (l >> 4) + r of the 16 x 16 block: the lane
    // OWNS its queries, the thresholds are four per-lane scalars for the whole kernel
    float th[FB];
    if (SCAN) {
#pragma unroll
        for (int b = 0; b < FB; ++b) th[b] = p.theta[col * CW + wj * QW + b * 16 + lr];
    }
    uint2* wbuf = (uint2*)(smem + 4 * GS_STG) + wave * GS_WBUF_ENTRIES;
    uint32_t cnt = 0;                                   // entries in this wave's buffer (wave-uniform)
    auto flush = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the epilogue's ds_write_b64 are inline asm
        const uint32_t nrows = (uint32_t)(end - begin);
        for (uint32_t i = (uint32_t)lane_now(); i < cnt; i += 64) {
            const uint2 e = wbuf[i];
            if ((e.y & 0xffffffu) >= nrows) continue;           // a row past the end of the range (zeros of a partial last tile)
            const uint32_t qq = (uint32_t)(col * CW) + (e.y >> 24);
            const uint32_t gs = atomicAdd(&p.gcnt[qq], 1u);
            if (gs < (uint32_t)p.gcap) p.lists[(size_t)qq * p.gcap + gs] = make_uint2(e.x, (uint32_t)begin + (e.y & 0xffffffu));
            else p.qflag[qq] = 1u;                      // the list is full (mass ties, no usable threshold): exact path
        }
        cnt = 0;
    };

    f32x4 acc[8][FB];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto epilogue = [&](const int ti) __attribute__((always_inline)) {
        if (SCAN) {
            // (1) which of the wave's query COLUMNS hold a passing score: the maximum of a column's 32 scores (8 fragments x 4 rows) is a chain
            // of 16 v_max3, one compare per column -- ~70 VALU instructions, no branch. (Round 4's first version tested the 32 fragments one
            // by one: ~230 instructions, 1 200 cycles per tile and wave.) A lane OWNS its query columns: th[b] is per-lane.
            uint32_t colhit = 0;
            GS_ESTAMP(0);
            // (round 5: the FB column chains step by step, BREADTH first -- written column by column hipcc emitted each column's 16 v_max3 as one
            //  dependent chain, an instruction-level parallelism of one; scheduling barriers keep the step order)
            float m[FB];
            auto el = [&](const int b, const int i) __attribute__((always_inline)) -> float { return acc[i >> 2][b][i & 3]; };     // element i = 0..31 of column b
#pragma unroll
            for (int b = 0; b < FB; ++b) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m[b]) : "v"(el(b, 0)), "v"(el(b, 1)), "v"(el(b, 2)));      // (fmaxf: two canonicalising v_max more)
#pragma unroll
            for (int s = 1; s < 15; ++s) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < FB; ++b) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m[b]) : "v"(m[b]), "v"(el(b, 2 * s + 1)), "v"(el(b, 2 * s + 2)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                m[b] = fmaxf(m[b], el(b, 31));
                colhit |= (__builtin_amdgcn_ballot_w64(m[b] > th[b]) != 0ull ? 1u : 0u) << b;
            }
            GS_ESTAMP(1);
            if (colhit != 0u) {
                // (everything lane-derived is formed HERE from a lane id hipcc cannot trace back: hoisted over the k-loop -- which runs at the
                //  register cap: 128 accumulators + 96 fragment registers -- it would push fragments into scratch)
                const int ln = lane_now(), lr_e = ln & 15, lg_e = ln >> 4;
                const uint32_t tag0 = ((uint32_t)(wj * QW + lr_e) << 24) | (uint32_t)((p.tile_begin + ti) * GS_TILE + wi * 128 + lg_e * 4);   // (query << 24) | row relative to `begin` (< 2^24) of acc[0][0][0]
                const uint32_t wb = lds0 + 4 * GS_STG + (uint32_t)wave * (GS_WBUF_ENTRIES * 8);
                if (cnt > GS_WBUF_REAL / 2) flush();
                // (2) passing scores take the buffer slots cnt, cnt + 1, ... in walk order (ballot prefix counts: no atomics, no barrier). A
                // handful pass per tile and wave and the buffer has room for at least GS_WBUF_REAL / 2 more; slots are CLAMPED to the buffer,
                // and what a tile brings beyond that (no usable threshold, mass ties: adversarial data) sends ITS queries to the
                // exact path, as a full list does. Rows past the end of the range (zeros in a partial last tile) are dropped by the flush.
                // The fragment bodies are kept SMALL and OUT OF LINE (the 32 copies are ~10 KB of code that stays in the instruction cache).
                // TAKEN BRANCHES are what this walk costs (~80 cycles each beside a partner wave that streams MFMAs): one test per column
                // (a column holds a passing score in ~30 % of the tiles), inside a hit column one test per fragment whose COMMON outcome --
                // nothing here -- falls through, and NO branch inside a hit fragment: every lane stores, the lanes without a passing score
                // into a dummy slot of their own behind the real ones.
                const uint32_t dummy = wb + (uint32_t)(GS_WBUF_REAL + ln) * 8u;
                const uint32_t rot16 = CERT ? (uint32_t)(wj * 32) : 0u;     // MODE 2: register slot a holds slab fragment a ^ 2 wj (rows (a * 16) ^ (2 wj * 16))
                uint32_t lost = 0;                      // bit b: a passing score of the lane's query column b found no slot
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    if ((colhit & (1u << b)) == 0u) continue;
                    // which of the column's 8 fragments: eight independent chains that end in scalar bit arithmetic (a chain that ends in a
                    // branch costs its whole latency, ~100 cycles per fragment), then one test per fragment that falls through
                    uint32_t fhit = 0;
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const f32x4 v = acc[a][b];
                        float m;
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
                        const uint64_t any = __builtin_amdgcn_ballot_w64(m > th[b]) | __builtin_amdgcn_ballot_w64(v[3] > th[b]);
                        fhit |= (any != 0ull ? 1u : 0u) << a;
                    }
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        if (__builtin_expect((fhit & (1u << a)) == 0u, 1)) continue;
                        const f32x4 v = acc[a][b];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool pass = v[r] > th[b];
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                            uint32_t idx = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            // (the buffer's last slot is a trash slot: what lands there or beyond it is LOST, and the query it belongs to is noted)
                            if (pass && idx >= GS_WBUF_REAL - 1) lost |= 1u << b;
                            idx = idx < GS_WBUF_REAL - 1 ? idx : GS_WBUF_REAL - 1;
                            const unsigned long long e = (unsigned long long)f32_bits(v[r]) |
                                                         ((unsigned long long)(tag0 + (((uint32_t)(b * 16) << 24) | (((uint32_t)(a * 16) ^ rot16) + (uint32_t)r))) << 32);
                            // (asm: behind an LDS-DMA it cannot prove disjoint hipcc puts s_waitcnt vmcnt(0) in front of every LDS store)
                            asm volatile("ds_write_b64 %0, %1" :: "v"(pass ? wb + idx * 8u : dummy), "v"(e) : "memory");
                            cnt += (uint32_t)__popcll(mask);
                        }
                    }
                }
                if (cnt > GS_WBUF_REAL - 1) {           // entries were lost: exactly the queries that lost one take the exact path
                    cnt = GS_WBUF_REAL - 1;
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        if (lost & (1u << b)) p.qflag[col * CW + wj * QW + b * 16 + lr_e] = 1u;
                }
            }
            GS_ESTAMP(2);
#if ATLAS_TUNING
            if (p.dbg != nullptr && blockIdx.x == 0 && ti < 8 && lane_now() == 0)
                p.dbg[(size_t)8 * GS_STAMP_ITERS * 8 + ((size_t)wave * 8 + ti) * 4 + 3] = ((unsigned long long)__popc(colhit) << 32) | cnt;
#endif
        } else {
            const int ts = range + ti * nranges;
            const size_t ldq = (size_t)p.ncol * CW;
            const int ln = lane_now(), lr_e = ln & 15, lg_e = ln >> 4;
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const f32x4 v = acc[a][b];
                    float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    if (lg_e == 0) p.smax[((size_t)ts * GS_FRAG_PER_TILE + wi * 8 + a) * ldq + (size_t)(col * CW + wj * QW + b * 16 + lr_e)] = m;
                }
        }
    };

    // Phases are separated by s_barrier (all 8 waves); group A = waves 0-3, group B = waves 4-7 (wave w and w + 4 share a SIMD):
    //     phase 2i     : A reads k-tile i, issues its pieces of k-tile i + 1       | B multiplies k-tile i - 1
    //     phase 2i + 1 : A multiplies k-tile i                                     | B reads k-tile i, issues its pieces of k-tile i + 2
    // One flat loop over the k-tiles of ALL tiles of the workgroup: the staging runs straight through the tile boundaries. The epilogue of a
    // tile needs no barrier and no stage buffer, and the phases stay aligned across it: group B runs its epilogue where its tile ends (behind
    // the barrier of its last multiply phase, in front of its next reads); group A, whose tile ends a phase earlier, first READS the next
    // tile's first k-tile (beside group B's last MFMAs, as in every other phase; the fragments wait in registers) and runs its epilogue
    // behind that phase's barrier, in front of its own MFMAs -- side by side with group B's (two waves per SIMD fill each other's gaps).
    auto nothing = [](auto) __attribute__((always_inline)) {};
    if (!grpB) stage(std::false_type{}, 0, 0, nothing); else stage(std::true_type{}, 0, 0, nothing);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's pieces of k-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grpB) {                                        // B's phase 0: nothing to multiply yet
        if (total_it > 1) stage(std::true_type{}, 1, 1, nothing);
        __builtin_amdgcn_s_barrier();
    }
    int kt = 0, ti = 0, a_due = 0;
    // MODE 2: the four per-wave displacements of the slab slot PAIRS (see `bs` in the loop), wave-uniform: slots 2 j, 2 j + 1 hold fragments
    // (2 j) ^ 2 wj and + 1
    const int cperm[4] = {((0 ^ (2 * wj)) & 7) * 2048, ((2 ^ (2 * wj)) & 7) * 2048, ((4 ^ (2 * wj)) & 7) * 2048, ((6 ^ (2 * wj)) & 7) * 2048};
    float nrm0 = 0.f, nrm1 = 0.f, pm = 0.f;          // MODE 2: running sums of squares of two slab rows' elements, largest row sum seen
#pragma unroll 1
    for (int it = 0; it < total_it; ++it) {
        const int buf = it & 1;
        GS_STAMP(0);
        gs_u4 fs0[8], fq0[FB], fs1[8], fq1[FB];
        const uint32_t s0 = as0 + buf * GS_STG, s1 = s0 ^ 64u, q0 = s0 + (uint32_t)(2 * GS_STG + (wj * QW - wi * 128) * 128), q1 = q0 ^ 64u;
        // MODE 2: slot a holds slab fragment a ^ 2 wj, at byte offset (a ^ 2 wj) * 2048 = ((a & 6) ^ 2 wj) * 2048 + (a & 1) * 2048: a wave-uniform
        // displacement per slot PAIR (four bases, slot a takes bs[a >> 1]) plus the compile-time offset 0 | 2048
        uint32_t bs[4] = {s0, s0, s0, s0};
        if constexpr (CERT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[j] = s0 + (uint32_t)cperm[j];
        }
        // slab fragment read of register slot k (k-step 0 | 1)
        auto rd_s0 = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (CERT) gs_ds_read<(k & 1) * 2048>(fs0[k], bs[k >> 1]); else gs_ds_read<k * 2048>(fs0[k], s0);
        };
        auto rd_s1 = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (CERT) {
                if constexpr (k == 0) {                // (the step-0 slab reads have all been issued: the bases move on to k-step 1 in place)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bs[j] ^= 64u;
                }
                gs_ds_read<(k & 1) * 2048>(fs1[k], bs[k >> 1]);
            } else gs_ds_read<k * 2048>(fs1[k], s1);
        };
        constexpr int NR = 16 + 2 * FB;                // fragment reads of a phase
        // (every fragment register is an in/out operand of the phase's last wait: nothing reads one in front of it)
#define GS_FS_OPS "+v"(fs0[0]), "+v"(fs0[1]), "+v"(fs0[2]), "+v"(fs0[3]), "+v"(fs0[4]), "+v"(fs0[5]), "+v"(fs0[6]), "+v"(fs0[7]), \
                  "+v"(fs1[0]), "+v"(fs1[1]), "+v"(fs1[2]), "+v"(fs1[3]), "+v"(fs1[4]), "+v"(fs1[5]), "+v"(fs1[6]), "+v"(fs1[7]), \
                  "+v"(fq0[0]), "+v"(fq0[1]), "+v"(fq1[0]), "+v"(fq1[1])
#define GS_READS_DONE() do { \
            if constexpr (FB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : GS_FS_OPS, "+v"(fq0[2]), "+v"(fq1[2]), "+v"(fq0[3]), "+v"(fq1[3]) :: "memory"); \
            else if constexpr (FB == 3) asm volatile("s_waitcnt lgkmcnt(0)" : GS_FS_OPS, "+v"(fq0[2]), "+v"(fq1[2]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : GS_FS_OPS :: "memory"); } while (0)
        const bool stages = grpB ? (it + 2 < total_it) : (it + 1 < total_it);
        if (stages && !grpB) {
            // group A: its pieces of k-tile it + 1 (the other stage), every piece followed by its share of the reads
            __builtin_amdgcn_sched_barrier(0);
            stage(std::false_type{}, buf ^ 1, it + 1, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                gs_static_for<i * NR / NPW, (i + 1) * NR / NPW>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < 8) rd_s0(std::integral_constant<int, k>{});
                    else if constexpr (k < 8 + FB) gs_ds_read<(k - 8) * 2048>(fq0[k - 8], q0);
                    else if constexpr (k < 16 + FB) rd_s1(std::integral_constant<int, k - 8 - FB>{});
                    else gs_ds_read<(k - 16 - FB) * 2048>(fq1[k - 16 - FB], q1);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            GS_READS_DONE();
        } else if (stages) {
            // group B: its pieces of k-tile it + 2 into THIS stage. The query fragments first; the four slab pieces (rows only group A reads,
            // and read a phase ago), each followed by its share of the slab reads; then -- the query reads have returned: lgkmcnt = the slab
            // reads issued since -- the query pieces, which overwrite the rows just read
            __builtin_amdgcn_sched_barrier(0);
            gs_static_for<0, FB>([&](auto bc) __attribute__((always_inline)) {
                constexpr int b = decltype(bc)::value;
                gs_ds_read<b * 2048>(fq0[b], q0);
                gs_ds_read<b * 2048>(fq1[b], q1);
            });
            stage(std::true_type{}, buf, it + 2, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                gs_static_for<i * 16 / NPW, (i + 1) * 16 / NPW>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < 8) rd_s0(std::integral_constant<int, k>{}); else rd_s1(std::integral_constant<int, k - 8>{});
                });
                if constexpr (i == 3) {
                    constexpr int since = 4 * 16 / NPW;
                    if constexpr (since == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    else if constexpr (since == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
                    else { static_assert(since == 10, "NPW 6 | 7 | 8"); asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            GS_READS_DONE();
        } else {
            __builtin_amdgcn_sched_barrier(0);
            // (inline asm: hipcc's wait insertion would drain vmcnt(0) in front of any ds_read it sees behind an LDS-DMA it cannot prove disjoint)
            if constexpr (CERT) {                      // (the permuted slots: single reads, as in the staging phases)
                gs_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { rd_s0(kc); });
                gs_static_for<0, FB>([&](auto bc) __attribute__((always_inline)) { constexpr int b = decltype(bc)::value; gs_ds_read<b * 2048>(fq0[b], q0); });
                gs_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { rd_s1(kc); });
                gs_static_for<0, FB>([&](auto bc) __attribute__((always_inline)) { constexpr int b = decltype(bc)::value; gs_ds_read<b * 2048>(fq1[b], q1); });
                GS_READS_DONE();
            } else if constexpr (FB == 4) {
                asm volatile(
                    "ds_read_b128 %0, %24\n ds_read_b128 %1, %24 offset:2048\n ds_read_b128 %2, %24 offset:4096\n ds_read_b128 %3, %24 offset:6144\n"
                    "ds_read_b128 %4, %24 offset:8192\n ds_read_b128 %5, %24 offset:10240\n ds_read_b128 %6, %24 offset:12288\n ds_read_b128 %7, %24 offset:14336\n"
                    "ds_read_b128 %8, %25\n ds_read_b128 %9, %25 offset:2048\n ds_read_b128 %10, %25 offset:4096\n ds_read_b128 %11, %25 offset:6144\n"
                    "ds_read_b128 %12, %26\n ds_read_b128 %13, %26 offset:2048\n ds_read_b128 %14, %26 offset:4096\n ds_read_b128 %15, %26 offset:6144\n"
                    "ds_read_b128 %16, %26 offset:8192\n ds_read_b128 %17, %26 offset:10240\n ds_read_b128 %18, %26 offset:12288\n ds_read_b128 %19, %26 offset:14336\n"
                    "ds_read_b128 %20, %27\n ds_read_b128 %21, %27 offset:2048\n ds_read_b128 %22, %27 offset:4096\n ds_read_b128 %23, %27 offset:6144\n"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(fs0[0]), "=&v"(fs0[1]), "=&v"(fs0[2]), "=&v"(fs0[3]), "=&v"(fs0[4]), "=&v"(fs0[5]), "=&v"(fs0[6]), "=&v"(fs0[7]),
                      "=&v"(fq0[0]), "=&v"(fq0[1]), "=&v"(fq0[2]), "=&v"(fq0[3]),
                      "=&v"(fs1[0]), "=&v"(fs1[1]), "=&v"(fs1[2]), "=&v"(fs1[3]), "=&v"(fs1[4]), "=&v"(fs1[5]), "=&v"(fs1[6]), "=&v"(fs1[7]),
                      "=&v"(fq1[0]), "=&v"(fq1[1]), "=&v"(fq1[2]), "=&v"(fq1[3])
                    : "v"(s0), "v"(q0), "v"(s1), "v"(q1)
                    : "memory");
            } else if constexpr (FB == 3) {
                asm volatile(
                    "ds_read_b128 %0, %22\n ds_read_b128 %1, %22 offset:2048\n ds_read_b128 %2, %22 offset:4096\n ds_read_b128 %3, %22 offset:6144\n"
                    "ds_read_b128 %4, %22 offset:8192\n ds_read_b128 %5, %22 offset:10240\n ds_read_b128 %6, %22 offset:12288\n ds_read_b128 %7, %22 offset:14336\n"
                    "ds_read_b128 %8, %23\n ds_read_b128 %9, %23 offset:2048\n ds_read_b128 %10, %23 offset:4096\n"
                    "ds_read_b128 %11, %24\n ds_read_b128 %12, %24 offset:2048\n ds_read_b128 %13, %24 offset:4096\n ds_read_b128 %14, %24 offset:6144\n"
                    "ds_read_b128 %15, %24 offset:8192\n ds_read_b128 %16, %24 offset:10240\n ds_read_b128 %17, %24 offset:12288\n ds_read_b128 %18, %24 offset:14336\n"
                    "ds_read_b128 %19, %25\n ds_read_b128 %20, %25 offset:2048\n ds_read_b128 %21, %25 offset:4096\n"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(fs0[0]), "=&v"(fs0[1]), "=&v"(fs0[2]), "=&v"(fs0[3]), "=&v"(fs0[4]), "=&v"(fs0[5]), "=&v"(fs0[6]), "=&v"(fs0[7]),
                      "=&v"(fq0[0]), "=&v"(fq0[1]), "=&v"(fq0[2]),
                      "=&v"(fs1[0]), "=&v"(fs1[1]), "=&v"(fs1[2]), "=&v"(fs1[3]), "=&v"(fs1[4]), "=&v"(fs1[5]), "=&v"(fs1[6]), "=&v"(fs1[7]),
                      "=&v"(fq1[0]), "=&v"(fq1[1]), "=&v"(fq1[2])
                    : "v"(s0), "v"(q0), "v"(s1), "v"(q1)
                    : "memory");
            } else {
                asm volatile(
                    "ds_read_b128 %0, %20\n ds_read_b128 %1, %20 offset:2048\n ds_read_b128 %2, %20 offset:4096\n ds_read_b128 %3, %20 offset:6144\n"
                    "ds_read_b128 %4, %20 offset:8192\n ds_read_b128 %5, %20 offset:10240\n ds_read_b128 %6, %20 offset:12288\n ds_read_b128 %7, %20 offset:14336\n"
                    "ds_read_b128 %8, %21\n ds_read_b128 %9, %21 offset:2048\n"
                    "ds_read_b128 %10, %22\n ds_read_b128 %11, %22 offset:2048\n ds_read_b128 %12, %22 offset:4096\n ds_read_b128 %13, %22 offset:6144\n"
                    "ds_read_b128 %14, %22 offset:8192\n ds_read_b128 %15, %22 offset:10240\n ds_read_b128 %16, %22 offset:12288\n ds_read_b128 %17, %22 offset:14336\n"
                    "ds_read_b128 %18, %23\n ds_read_b128 %19, %23 offset:2048\n"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(fs0[0]), "=&v"(fs0[1]), "=&v"(fs0[2]), "=&v"(fs0[3]), "=&v"(fs0[4]), "=&v"(fs0[5]), "=&v"(fs0[6]), "=&v"(fs0[7]),
                      "=&v"(fq0[0]), "=&v"(fq0[1]),
                      "=&v"(fs1[0]), "=&v"(fs1[1]), "=&v"(fs1[2]), "=&v"(fs1[3]), "=&v"(fs1[4]), "=&v"(fs1[5]), "=&v"(fs1[6]), "=&v"(fs1[7]),
                      "=&v"(fq1[0]), "=&v"(fq1[1])
                    : "v"(s0), "v"(q0), "v"(s1), "v"(q1)
                    : "memory");
            }
        }
#undef GS_READS_DONE
#undef GS_FS_OPS
        GS_STAMP(1);
        // what must have LANDED before the barrier: pieces issued before this phase (the slab rows the other group reads next, group B's
        // query rows) -- all but this phase's NPW
        if (stages) __builtin_amdgcn_s_waitcnt(0x0F70 | NPW); else __builtin_amdgcn_s_waitcnt(0x0F70);
        GS_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GS_STAMP(3);
        GS_STAMP(4);
        // group A: the epilogue of the tile that ended a phase ago (see above). (A flag of its own, opaque to hipcc: derived from `kt == 0` --
        // the condition of the C = 0 form of the MFMAs below -- the whole phase is threaded into two copies, and the second copy's registers
        // go to scratch)
        asm volatile("" : "+s"(a_due));
        if (a_due != 0) { epilogue(ti - 1); a_due = 0; }
        // MODE 2: row sums of squares of this wave's two of the eight fragments of its slab rows, 2 wj and 2 wj + 1 = its register slots 0 and 1
        // (the lane holds 8 + 8 of a row's 64 elements of this k-tile), in SIXTEEN v_dot2 per k-tile: units 0..7 of each k-step, one behind
        // every group of FB MFMAs. (Round 4: the wave picked the two fragments out of eight identical slots with 3 v_cndmask per v_dot2 -- 64 VALU
        // instructions per k-tile that did NOT vanish in the MFMAs' shadow: multiply phase + 100-170 cycles, the partner's read phase as
        // much, 3.43 ms against the trusting twin's 2.77 at 4M rows x 512 queries, profiles/r04/gscan_certifying_twin.txt.)
        auto cert_unit = [&](const gs_u4 (&f)[8], auto uc) __attribute__((always_inline)) {        // unit u of a k-step: dword u & 3 of slot u >> 2 (= fragment 2 wj + (u >> 2))
            if constexpr (CERT) {
                constexpr int u = decltype(uc)::value, e = (u >> 2) & 1, d = u & 3;
                // (a BUILTIN, and left where hipcc puts it: as pure arithmetic the sixteen v_dot2 of a k-tile end up in front of and behind the MFMA
                //  block, and that is the cheapest place -- pinned one behind every group of FB MFMAs (volatile asm between scheduling barriers)
                //  they cost the wave's own dependency-paced MFMA stream as much as round 4's selects did: same-process A/B of three builds,
                //  4M rows x 128 / 256 / 512 queries, certifying over trusting: round 4's selects + 8.0 / + 9.1 / + 9.7 %, these + 3.4 / + 5.8 /
                //  + 6.9 %, pinned + 4.1 / + 9.5 / + 10.2 % (profiles/r05/gscan_cert_ab.txt). As inline asm they also lost a dot: hipcc's hazard
                //  recognizer does not look inside asm, and a DOT's result read by a different VALU opcode needs 3 wait states --
                //  tests/test_gpu_search.py::test_gemm_shaped_certifying_twin_measures_every_fragment_of_a_tile found it.)
                __builtin_amdgcn_sched_barrier(0);         // (kept: this is the build that was measured; it does not pin the v_dot2, see above)
                const f16x2 h = __builtin_bit_cast(f16x2, (uint32_t)f[e][d]);
                if (e == 0) nrm0 = __builtin_amdgcn_fdot2(h, h, nrm0, false); else nrm1 = __builtin_amdgcn_fdot2(h, h, nrm1, false);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (kt == 0) {                                 // a tile's first k-tile starts from C = 0 (an inline constant: no 128 v_mov per tile)
            gs_static_for<0, 8>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
#pragma unroll
                for (int b = 0; b < FB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fs0[a]), __builtin_bit_cast(f16x8, fq0[b]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cert_unit(fs0, ac);
            });
        } else {
            gs_static_for<0, 8>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
#pragma unroll
                for (int b = 0; b < FB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fs0[a]), __builtin_bit_cast(f16x8, fq0[b]), acc[a][b], 0, 0, 0);
                cert_unit(fs0, ac);
            });
        }
        gs_static_for<0, 8>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
#pragma unroll
            for (int b = 0; b < FB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fs1[a]), __builtin_bit_cast(f16x8, fq1[b]), acc[a][b], 0, 0, 0);
            cert_unit(fs1, ac);
        });
        __builtin_amdgcn_sched_barrier(0);
        GS_STAMP(5);
        // A: its QUERY pieces of k-tile it + 1 (issued first; read in the next phase) have landed, the four slab pieces may still fly
        if (!grpB) { if (stages) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        GS_STAMP(6);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GS_STAMP(7);
        if (++kt == GS_NK) {
            if (CERT) {                                // full row norms: the 4 lanes {l, l + 16, l + 32, l + 48} hold the 4 chunks of row l & 15
                float x = nrm0, y = nrm1;
                x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
                y += __shfl_xor(y, 16); y += __shfl_xor(y, 32);
                pm = fmaxf(pm, fmaxf(x, y));
                nrm0 = 0.f; nrm1 = 0.f;
            }
            if (grpB) epilogue(ti); else a_due = 1;
            kt = 0;
            ++ti;
        }
    }
    if (!grpB) { __builtin_amdgcn_s_barrier(); epilogue(ti - 1); }      // (the barrier group B's phase 0 is owed; group A's last tile)
    if (SCAN) flush();
#if ATLAS_TUNING
    if (MODE != 1 && p.dbg != nullptr && blockIdx.x == 0 && tid == 0) { p.dbg[8 * GS_STAMP_ITERS * 8 + 8 * 8 * 4 + 2] = wall_clock64(); p.dbg[8 * GS_STAMP_ITERS * 8 + 8 * 8 * 4 + 3] = __builtin_readcyclecounter(); }
#endif
    if (CERT) {
        // largest row norm^2 of the workgroup (x 1.001: v_dot2 accumulates in fp32): waves -> LDS -> one word pair, as scan_kernel.h leaves it
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
        float* s_pm = (float*)(smem + 4 * GS_STG);      // (the wave buffers are empty now)
        __syncthreads();
        if (lane == 0) s_pm[wave] = pm;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < 8; ++w) m = fmaxf(m, s_pm[w]);
            m *= 1.001f;
            p.wg_stat[(size_t)blockIdx.x * 2 + 0] = f32_bits(m);
            p.wg_stat[(size_t)blockIdx.x * 2 + 1] = (m > p.pmax2_hint) ? (uint32_t)ATLAS_F_PMAX_VIOLATION : 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------
// gprep_kernel: queries of a pass -> fp16 rows (RNE = `.half()`, src/index.py:117) in the workspace (the LDS-DMA of gscan_kernel copies
// bytes), and the per-pass state: list lengths to zero, status header (first pass of a call). One block of 96 threads per query.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(96)
gprep_kernel(const void* __restrict__ q, const int q_dtype, const int q0, uint4* __restrict__ q16, uint32_t* __restrict__ gcnt,
             int32_t* __restrict__ out_status) {
    const int j = blockIdx.x, t = threadIdx.x;
    q16[(size_t)j * (D_FAST / 8) + t] = q8_to_f16(load_q8(q, q_dtype, (size_t)(q0 + j) * D_FAST + (size_t)t * 8), q_dtype);
    if (t == 0) gcnt[j] = 0u;
    if (q0 == 0 && j == 0 && t < ATLAS_STATUS_HEADER) out_status[t] = 0;
}

// ------------------------------------------------------------------------------------------
// gtheta_kernel: the certified initial threshold of every query of a pass from the sample's fragment maxima. One block per query slot
// (ldq = ncol * 256 of them; slots >= nq get +inf: they never collect anything). A lower bound T of the k-th largest of the nmax maxima
// from one 1024-bin histogram over [min, max] (any T with count(maxima >= T) >= k is valid: k DISTINCT rows score >= T - eps), then
// prune_threshold(T, eps) (common.h). Fewer than k maxima: -inf.
// ------------------------------------------------------------------------------------------
// LIST mode (lists != nullptr, between the two scan launches of a pass): the same from the approximate scores of the candidates the first
// launch left in lists[q] (all written: a kernel boundary lies in between) -- the k-th best of the first eighth of the slab prunes the rest
// ~8 x harder than the sample's threshold; theta[q] = max(old, new), both are certified.
__global__ void __launch_bounds__(256)
gtheta_kernel(const float* __restrict__ smax, int nmax, const int ldq, const uint16_t* __restrict__ q16, const int nq, const float pmax,
              const int k, float* __restrict__ theta, const uint2* __restrict__ lists, const uint32_t* __restrict__ gcnt, const int gcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (all LDS is dynamic, GTHETA_LDS(nmax) bytes: the launch may ask for up to 133 KiB)
    double* s_ss = (double*)smem;                      // [4]
    uint32_t* s_mm = (uint32_t*)(smem + 32);           // [2]; [2] = the bin edge
    uint32_t& s_bin = s_mm[2];
    uint32_t* hist = (uint32_t*)(smem + 64);           // [1024]
    uint32_t* keys = hist + 1024;                      // [nmax]
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (q >= nq) { if (tid == 0) theta[q] = pos_inf(); return; }
    const bool from_list = lists != nullptr;
    if (from_list) {
        const uint32_t n = gcnt[q];
        nmax = (int)(n < (uint32_t)gcap ? n : (uint32_t)gcap);
        if (nmax > GTHETA_MAXKEYS) nmax = GTHETA_MAXKEYS;       // (any subset of the candidates gives a valid threshold)
    }
    double ss = 0.0;
    for (int i = tid; i < D_FAST; i += 256) { const double v = (double)(float)__builtin_bit_cast(_Float16, q16[(size_t)q * D_FAST + i]); ss += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) s_ss[wave] = ss;
    if (tid == 0) { s_mm[0] = 0xffffffffu; s_mm[1] = 0u; s_bin = 0u; }
    for (int i = tid; i < 1024; i += 256) hist[i] = 0u;
    __syncthreads();
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (int i = tid; i < nmax; i += 256) {
        const uint32_t key = f32_order_key(from_list ? bits_f32(lists[(size_t)q * gcap + i].x) : smax[(size_t)i * ldq + q]);
        keys[i] = key;
        kmin = key < kmin ? key : kmin;
        kmax = key > kmax ? key : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { atomicMin(&s_mm[0], kmin); atomicMax(&s_mm[1], kmax); }
    __syncthreads();
    kmin = s_mm[0]; kmax = s_mm[1];
    const float eps = query_eps((float)(s_ss[0] + s_ss[1] + s_ss[2] + s_ss[3]) * 1.000001f, pmax);
    if (nmax < k) { if (tid == 0 && !from_list) theta[q] = neg_inf(); return; }
    const uint32_t span = kmax - kmin;
    const int shift = span >= 1024u ? (32 - __builtin_clz(span)) - 10 : 0;      // (key - kmin) >> shift < 1024
    for (int i = tid; i < nmax; i += 256) atomicAdd(&hist[(keys[i] - kmin) >> shift], 1u);
    __syncthreads();
    if (tid < 64) {                                    // lane l owns bins [16 l, 16 l + 16): suffix sums find the lane, then the bin
        uint32_t own = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) own += hist[tid * 16 + b];
        uint32_t suf = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_down(suf, o);
            if (tid + o < 64) suf += y;
        }
        const uint32_t above = suf - own;
        if (above < (uint32_t)k && suf >= (uint32_t)k) {
            uint32_t acc = above; int bin = tid * 16;
            for (int b = 15; b >= 0; --b) {
                acc += hist[tid * 16 + b];
                if (acc >= (uint32_t)k) { bin = tid * 16 + b; break; }
            }
            s_bin = kmin + ((uint32_t)bin << shift);  // lower edge of the bin that holds rank k
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float th = prune_threshold(f32_from_order_key(s_bin), eps);
        theta[q] = from_list ? fmaxf(theta[q], th) : th;
    }
}

}  // namespace atlas
