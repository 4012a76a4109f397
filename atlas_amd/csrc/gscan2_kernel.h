// gscan2_kernel.h -- EXPERIMENT (tuning build; atlas_tune_set_scan_gemm(2)): the GEMM-shaped pass of gscan_kernel.h with the SLAB operand
// streamed HBM -> VGPR through a register ring, as scan_kernel.h streams it, instead of through LDS. Only the queries are staged in LDS.
//
// Why: in gscan_kernel a k-tile takes ~3.4k cycles where its 128 MFMAs per SIMD need 2.05k; LDS-DMA is 64 pieces of 1 KiB per k-tile and CU
// (half slab, half queries) through a path that sustains ~45 B/clk, and neither hiding the exposed landing time (a third slab stage) nor moving
// the issue between the groups changed that. Here a wave owns 32 slab rows x ALL 256 queries of the column tile: its two 16-row fragments per
// k-step come straight from the row-major slab (the lane's 16 bytes of row l & 15 are exactly the MFMA A operand: scan_kernel's load), kept
// RING k-steps ahead in registers -- no LDS-DMA, no LDS space and no barrier for the slab. The queries (32 KiB per 64-wide k-tile) go through
// FOUR LDS stages two k-tiles ahead, one barrier per k-tile.
//   per k-step (32 of k) and wave: 2 slab loads, 16 ds_read_b128 (in two halves of 8 query fragments), 32 MFMAs
//   registers: 128 accumulators + 48 ring + 32 query fragments
// MEASURED (profiles/r04/gscan2_register_ring_experiment.txt, first version, bit-identical results): 3-4 % SLOWER than gscan_kernel (4M rows:
// 256 queries 1.795 vs 1.733 ms, 512 3.267 vs 3.151, 1024 6.328 vs 6.089). Two different structures at the same ~4.1k cycles per k-tile all-in
// point at what they share: 8 waves of 256 registers, i.e. wave tiles of 128 x 64 / 32 x 256, 0.375 / 0.5 ds_read_b128 per MFMA and every
// LDS / vector-memory instruction issued from a SIMD whose other wave wants to issue MFMAs. The GEMMs that reach 70 % of the matrix pipe on
// this chip run ONE wave per SIMD with 128 x 128 wave tiles (accumulators in AGPRs, 0.25 reads per MFMA): the next step for this pass, not
// built this round. NOT part of the product library.
#pragma once
#include "gscan_kernel.h"

namespace atlas {

#define GS2_RING 6                // k-steps of slab fragments in registers (5 in flight); divides the 24 k-steps of a tile
#define GS2_QSTAGES 4
#define GS2_LDS_BYTES (GS2_QSTAGES * GS_STG + 1024 + 8 * GS_WBUF_ENTRIES * 8)      // 4 query stages | 256 thresholds | 8 wave buffers = 160 768

template <int MODE>
__global__ void __launch_bounds__(512)
gscan2_kernel(const GScanParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // Q0..Q3 (32 KiB each) | theta[256] | 8 wave buffers
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpB = wave >= 4;                                               // (wave w and w + 4 share a SIMD)
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int ROWB = D_FAST * 2;
    constexpr bool SCAN = MODE != 1, CERT = MODE == 2;
    constexpr int KS_TILE = D_FAST / 32;                                       // 24 k-steps per tile

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
    const int total_kt = ntl * GS_NK;                                          // k-tiles of 64 (= 2 k-steps)

    // ---- queries: LDS-DMA, 4 pieces per wave and k-tile (rows 32 w + 8 i + (lane >> 3)), source-side swizzle as in gscan_kernel.h
    const uint32_t chb = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    const uint32_t vq = (uint32_t)(wave * 32 + (lane >> 3)) * ROWB + chb;
    int qrows = p.nq - col * GS_TILE;
    qrows = __builtin_amdgcn_readfirstlane(qrows < 0 ? 0 : (qrows > GS_TILE ? GS_TILE : qrows));
    auto stage_q = [&](const int ktg) __attribute__((always_inline)) {       // k-tile ktg of the workgroup -> stage ktg & 3 (past the end: nothing is fetched)
        const int kt = ktg % GS_NK;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q16 + (size_t)col * GS_TILE * D_FAST), 0, ktg < total_kt ? qrows * ROWB : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t vo = vq;
            asm volatile("" : "+v"(vo));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr)(smem + (ktg & 3) * GS_STG + (wave * 32 + i * 8) * 128), 16, (int)(vo + (uint32_t)(i * 8 * ROWB)), kt * 128, 0, 0);
        }
    };
    // ---- slab: the wave's rows 32 w .. 32 w + 31 of a tile; lane (lr, lg) loads 16 bytes of row 16 a + lr at k-offset 64 ks + 16 lg
    const uint32_t vs = (uint32_t)(wave * 32 + lr) * ROWB + (uint32_t)lg * 16;
    gs_u4 ring[GS2_RING][2];
    auto load_slab = [&](const int slot_, const int ksg) __attribute__((always_inline)) {      // k-step ksg of the workgroup -> ring slot
        const int ti = ksg / KS_TILE, ks = ksg - ti * KS_TILE;
        const int64_t r0 = tile_row0(ti < ntl ? ti : 0);
        int64_t rem = ti < ntl ? (SCAN ? end : p.N) - r0 : 0;               // past the last tile: size 0, zeros, no traffic
        if (rem > GS_TILE) rem = GS_TILE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slab + (size_t)r0 * D_FAST), 0, (int)rem * ROWB, 0x00020000);
        uint32_t vo = vs;
        asm volatile("" : "+v"(vo));
        ring[slot_][0] = __builtin_bit_cast(gs_u4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, ks * 64, 0));
        ring[slot_][1] = __builtin_bit_cast(gs_u4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16u * ROWB), ks * 64, 0));
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // query fragment b (rows 16 b + lr) of k-step ks (chunks 4 (ks & 1) + lg of the 128-byte row): + b * 2048
    const uint32_t aq0 = lds0 + lr * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t aq1 = lds0 + lr * 128 + ((4 + lg) ^ (lr & 7)) * 16;
    float* s_theta = (float*)(smem + GS2_QSTAGES * GS_STG);                  // [256], as [lr][16 b]: a lane's 16 thresholds are 64 contiguous bytes
    uint2* wbuf = (uint2*)(smem + GS2_QSTAGES * GS_STG + 1024) + wave * GS_WBUF_ENTRIES;
    if (SCAN && tid < 256) s_theta[(tid & 15) * 16 + (tid >> 4)] = p.theta[col * GS_TILE + tid];

    uint32_t cnt = 0;
    auto flush = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t nrows = (uint32_t)(end - begin);
        for (uint32_t i = (uint32_t)lane_now(); i < cnt; i += 64) {
            const uint2 e = wbuf[i];
            if ((e.y & 0xffffffu) >= nrows) continue;
            const uint32_t qq = (uint32_t)(col * GS_TILE) + (e.y >> 24);
            const uint32_t gs = atomicAdd(&p.gcnt[qq], 1u);
            if (gs < (uint32_t)p.gcap) p.lists[(size_t)qq * p.gcap + gs] = make_uint2(e.x, (uint32_t)begin + (e.y & 0xffffffu));
            else p.qflag[qq] = 1u;
        }
        cnt = 0;
    };

    f32x4 acc[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float nrm0 = 0.f, nrm1 = 0.f, pm = 0.f;

    auto epilogue = [&](const int ti) __attribute__((always_inline)) {
        if (SCAN) {
            // the lane's 16 thresholds (queries 16 b + lr): four ds_read_b128 (asm: no vmcnt(0) in front of an LDS read behind an LDS-DMA)
            gs_u4 t4[4];
            {
                const uint32_t ta = lds0 + GS2_QSTAGES * GS_STG + (uint32_t)(lane_now() & 15) * 64;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(t4[0]), "=&v"(t4[1]), "=&v"(t4[2]), "=&v"(t4[3]) : "v"(ta) : "memory");
            }
            auto th = [&](const int b) { return bits_f32(t4[b >> 2][b & 3]); };
            uint32_t hit = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const f32x4 v = acc[a][b];
                    float m;
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
                    const uint64_t any = __builtin_amdgcn_ballot_w64(m > th(b)) | __builtin_amdgcn_ballot_w64(v[3] > th(b));
                    hit |= (any != 0ull ? 1u : 0u) << (a * 16 + b);
                }
            if (hit != 0u) {
                const int ln = lane_now(), lr_e = ln & 15, lg_e = ln >> 4;
                const uint32_t tag0 = ((uint32_t)lr_e << 24) | (uint32_t)((p.tile_begin + ti) * GS_TILE + wave * 32 + lg_e * 4);
                const uint32_t wb = lds0 + GS2_QSTAGES * GS_STG + 1024 + (uint32_t)wave * (GS_WBUF_ENTRIES * 8);
                if (cnt > GS_WBUF_REAL / 2) flush();
                const uint32_t dummy = wb + (uint32_t)(GS_WBUF_REAL + ln) * 8u;
#pragma unroll
                for (int g4 = 0; g4 < 8; ++g4) {
                    if ((hit & (0xfu << (g4 * 4))) == 0u) continue;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int a = (g4 * 4 + f) >> 4, b = (g4 * 4 + f) & 15;
                        if ((hit & (1u << (g4 * 4 + f))) == 0u) continue;
                        const f32x4 v = acc[a][b];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool pass = v[r] > th(b);
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                            uint32_t idx = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            idx = idx < GS_WBUF_REAL - 1 ? idx : GS_WBUF_REAL - 1;
                            const unsigned long long e = (unsigned long long)f32_bits(v[r]) |
                                                         ((unsigned long long)(tag0 + (((uint32_t)(b * 16) << 24) | (uint32_t)(a * 16 + r))) << 32);
                            asm volatile("ds_write_b64 %0, %1" :: "v"(pass ? wb + idx * 8u : dummy), "v"(e) : "memory");
                            cnt += (uint32_t)__popcll(mask);
                        }
                    }
                }
                if (cnt > GS_WBUF_REAL) {
                    cnt = GS_WBUF_REAL;
#pragma unroll
                    for (int b = 0; b < 16; ++b) p.qflag[col * GS_TILE + b * 16 + lr_e] = 1u;
                }
            }
        } else {
            const int ts = range + ti * nranges;
            const size_t ldq = (size_t)p.ncol * GS_TILE;
            const int ln = lane_now(), lr_e = ln & 15, lg_e = ln >> 4;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const f32x4 v = acc[a][b];
                    float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    if (lg_e == 0) p.smax[((size_t)ts * GS_FRAG_PER_TILE + wave * 2 + a) * ldq + (size_t)(col * GS_TILE + b * 16 + lr_e)] = m;
                }
        }
    };

    // ---- prologue: queries of k-tiles 0 and 1, the ring's first GS2_RING - 1 k-steps
    stage_q(0);
    stage_q(1);
#pragma unroll
    for (int s = 0; s < GS2_RING - 1; ++s) load_slab(s, s);
    int ksg = 0, ti = 0, kst = 0;                       // k-step of the workgroup, tile, k-step inside the tile
    // One loop iteration = one ring revolution = GS2_RING k-steps = 3 k-tiles (static ring indices: no register shuffling)
#pragma unroll 1
    for (int ktg = 0; ktg < total_kt; ktg += GS2_RING / 2) {
#pragma unroll
        for (int u = 0; u < GS2_RING / 2; ++u) {
            // start of k-tile ktg + u: this wave's query pieces of it (issued two k-tiles ago; since then 12 younger vector-memory operations
            // went out -- 4 + 4 slab loads and the 4 pieces of the next k-tile -- and loads complete in order) have landed; the barrier makes that
            // true for everybody's pieces and says that every wave is done with the stage the next pieces go to
            if (grpB) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);     // (group B's pieces went out half a k-tile later: 8 younger operations)
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!grpB) stage_q(ktg + u + 2);            // (group B issues its pieces half a k-tile later: the two waves of a SIMD do not stall together)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int s = 2 * u + k2;               // ring slot of this k-step (compile time)
                load_slab((s + GS2_RING - 1) % GS2_RING, ksg + GS2_RING - 1);
                __builtin_amdgcn_sched_barrier(0);
                if (grpB && k2 == 1) stage_q(ktg + u + 2);
                const uint32_t qa = (k2 ? aq1 : aq0) + (uint32_t)((ktg + u) & 3) * GS_STG;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    gs_u4 fq[8];
                    asm volatile(
                        "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:2048\n ds_read_b128 %2, %8 offset:4096\n ds_read_b128 %3, %8 offset:6144\n"
                        "ds_read_b128 %4, %8 offset:8192\n ds_read_b128 %5, %8 offset:10240\n ds_read_b128 %6, %8 offset:12288\n ds_read_b128 %7, %8 offset:14336\n"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(fq[0]), "=&v"(fq[1]), "=&v"(fq[2]), "=&v"(fq[3]), "=&v"(fq[4]), "=&v"(fq[5]), "=&v"(fq[6]), "=&v"(fq[7])
                        : "v"(qa + (uint32_t)(h * 16384))
                        : "memory");
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            acc[a][h * 8 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[s][a]), __builtin_bit_cast(f16x8, fq[b]),
                                                                                       acc[a][h * 8 + b], 0, 0, 0);
                }
                if (CERT) {
                    auto sq = [](const gs_u4& f, float acc2) {
                        const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const f16x2 hh = __builtin_bit_cast(f16x2, w[i]); acc2 = __builtin_amdgcn_fdot2(hh, hh, acc2, false); }
                        return acc2;
                    };
                    nrm0 = sq(ring[s][0], nrm0);
                    nrm1 = sq(ring[s][1], nrm1);
                }
                __builtin_amdgcn_sched_barrier(0);
                ++ksg;
            }
        }
        kst += GS2_RING;
        if (kst == KS_TILE) {                           // a tile is four revolutions: its end is the end of a revolution (ONE copy of the epilogue)
            if (CERT) {
                float x = nrm0, y = nrm1;
                x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
                y += __shfl_xor(y, 16); y += __shfl_xor(y, 32);
                pm = fmaxf(pm, fmaxf(x, y));
                nrm0 = 0.f; nrm1 = 0.f;
            }
            epilogue(ti);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 16; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kst = 0;
            ++ti;
        }
    }
    if (SCAN) flush();
    if (CERT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
        float* s_pm = (float*)smem;
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

}  // namespace atlas
