// dscan_proto.hip -- feasibility prototype (dev tool, round 6): can a 64-query slab pass run at the rate of a FULL-LINE, non-temporal stream?
// tools/read_ceiling.hip: a read-only stream in the scan's fragment shape (16 rows x 64 B per wave instruction, default policy: what scan_kernel.h
// does) gives 6.0-6.5 TB/s on this pool; full 128-B lines per instruction with `nt` give 6.6-6.9 (registers or LDS-DMA). The MFMA operand layout
// (a lane = one of 16 rows) cannot take full lines from a register load, so this prototype takes the slab through LDS-DMA instead:
//   * one 8-wave workgroup per CU; the slab tile (256 rows) is staged k-tile by k-tile (256 rows x 128 B = 32 KiB, 4 stages) by
//     `buffer_load_dwordx4 ... lds nt`, one wave instruction = 8 rows x 128 B, swizzled on the source address (as gscan_kernel.h);
//   * the QUERIES live in registers: wave w owns query fragment w & 3 (16 queries x 768 dims = 96 VGPRs) and the slab rows 128 (w >> 2) .. + 128;
//     per stage a wave issues 4 DMA pieces, 16 ds_read_b128 (slab fragments) and 16 MFMAs; no query image in LDS, no VALU in the k-loop;
//   * end of tile: threshold compare of the 32 accumulators (the lane owns its query column), passing scores counted (no lists here).
// Output: ms per pass and TB/s at the given shard size, and a check of the per-query maximum score against a plain fp32 dot-product kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dscan_proto.hip -o /tmp/dscan_proto && /tmp/dscan_proto [rows] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
#define D 768
#define ROWB 1536
#define TILE 256
#define NKT 12
#define STG (256 * 128)
#ifndef NSTAGE
#define NSTAGE 4
#endif
#ifndef DMA_AUX
#define DMA_AUX 2
#endif
#ifndef ISSUE4
#define ISSUE4 0
#endif
#ifndef DEAL
#define DEAL 0        // 1: the workgroup's tiles are dealt round-robin (tile = blockIdx + i * gridDim) instead of one contiguous range
#endif
#ifndef Q128
#define Q128 0        // 1: 128 queries per pass -- a wave = one of EIGHT query fragments x all 256 rows of the tile (16 accumulator fragments)
#endif
#define NQ (Q128 ? 128 : 64)
#define NA (Q128 ? 16 : 8)

struct Params { const uint16_t* slab; int64_t N; const uint16_t* q16; float theta; unsigned* qmax; unsigned long long* npass; int64_t rows_per_wg; };

__device__ __forceinline__ uint32_t okey(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

template <int OFF>
__device__ __forceinline__ void ds_read(u32x4& dst, const uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory"); }

__global__ void __launch_bounds__(512) dscan_proto(const Params p) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qf = Q128 ? wave : (wave & 3), half = Q128 ? 0 : (wave >> 2);
    const int lr = lane & 15, lg = lane >> 4;
#if DEAL
    const int64_t tiles_all = (p.N + TILE - 1) / TILE;
    const int ntl = (int)((tiles_all - (int64_t)blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int64_t begin = 0, end = p.N;
    if (ntl <= 0) return;
#if DEAL == 2     // ... and within the window every XCD (workgroup g runs on XCD g & 7) takes one contiguous eighth: 32 adjacent tiles = 12.5 MB per XCD
#define TILE_ROW0(ti) (((int64_t)(ti) * gridDim.x + (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * TILE)
#else
#define TILE_ROW0(ti) (((int64_t)(ti) * gridDim.x + blockIdx.x) * TILE)
#endif
#else
    const int64_t begin = (int64_t)blockIdx.x * p.rows_per_wg;
    int64_t end = begin + p.rows_per_wg; if (end > p.N) end = p.N;
    const int ntl = end > begin ? (int)((end - begin + TILE - 1) / TILE) : 0;
    if (ntl == 0) return;
#define TILE_ROW0(ti) (begin + (int64_t)(ti) * TILE)
#endif

    // the wave's 16 queries, MFMA B layout: lane (query 16 qf + lr, k-group lg), k-step s = elements 32 s + 8 lg .. + 8
    u32x4 bq[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) bq[s] = *(const u32x4*)(p.q16 + (size_t)(16 * qf + lr) * D + 32 * s + 8 * lg);

    const uint32_t chb = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    const uint32_t vbase = (uint32_t)(lane >> 3) * ROWB + chb;
    // this wave's 4 pieces of stage `it` (k-tile kt of tile ti) -> buffer buf: LDS rows 32 wave + 8 i + (lane >> 3)
    auto issue = [&](const int ti, const int kt, const int buf) __attribute__((always_inline)) {
        const int64_t r0 = TILE_ROW0(ti);
        int64_t rem = end - r0; if (rem > TILE) rem = TILE; if (rem < 0 || ti >= ntl) rem = 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slab + (size_t)(rem > 0 ? r0 : 0) * D), 0, (int)rem * ROWB, 0x00020000);
#if ISSUE4
        // variant: only waves 4..7 issue, 8 pieces each (tools/read_ceiling.hip: fewer issuing waves stream faster)
        if (wave < 4) return;
        unsigned char* const ls = smem + buf * STG + (wave - 4) * 8192;
        uint32_t vs = vbase + (uint32_t)((wave - 4) * 64 * ROWB);
        asm volatile("" : "+v"(vs));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ls + i * 1024), 16, (int)(vs + (uint32_t)(i * 8 * ROWB)), kt * 128, 0, DMA_AUX);
#else
        unsigned char* const ls = smem + buf * STG + wave * 4096;
        uint32_t vs = vbase + (uint32_t)(wave * 32 * ROWB);
        asm volatile("" : "+v"(vs));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ls + i * 1024), 16, (int)(vs + (uint32_t)(i * 8 * ROWB)), kt * 128, 0, DMA_AUX);
#endif
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t as0 = lds0 + (half * 128 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;

    f32x4 acc[NA];
    const float th = p.theta;
    float qm = -INFINITY;
    unsigned npass = 0;

    // prologue: stages 0 .. NSTAGE - 2 in flight
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s / NKT, s % NKT, s % NSTAGE);

#pragma unroll 1
    for (int ti = 0; ti < ntl; ++ti) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            constexpr int dummy = 0; (void)dummy;
            const int buf = kt % NSTAGE;                              // (NKT % NSTAGE == 0: the stage of a k-tile is a compile-time constant)
            // this wave's pieces of the stage about to be read have landed: everything but the (NSTAGE - 2) x 4 younger pieces
            if constexpr (NSTAGE == 4 && ISSUE4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and its reads of the stage about to be overwritten have returned
            __builtin_amdgcn_s_barrier();
            {   // refill the stage read in the previous iteration with k-tile it + NSTAGE - 1
                const int k2 = kt + NSTAGE - 1;
                issue(ti + (k2 >= NKT ? 1 : 0), k2 % NKT, k2 % NSTAGE);
            }
            const uint32_t s0 = as0 + buf * STG, s1 = s0 ^ 64u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int g8 = 0; g8 < NA / 8; ++g8) {
                    const uint32_t sa = (h ? s1 : s0) + g8 * 16384;
                    u32x4 f[8];
                    ds_read<0 * 2048>(f[0], sa); ds_read<1 * 2048>(f[1], sa); ds_read<2 * 2048>(f[2], sa); ds_read<3 * 2048>(f[3], sa);
                    ds_read<4 * 2048>(f[4], sa); ds_read<5 * 2048>(f[5], sa); ds_read<6 * 2048>(f[6], sa); ds_read<7 * 2048>(f[7], sa);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory");
#pragma unroll
                    for (int a = 0; a < 8; ++a)
                        acc[g8 * 8 + a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f[a]), __builtin_bit_cast(f16x8, bq[2 * kt + h]),
                                                                               (kt == 0 && h == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[g8 * 8 + a], 0, 0, 0);
                }
            }
        }
        // end of the tile: the lane owns query 16 qf + lr; rows half * 128 + 16 a + 4 lg + r
        const int nrows = (int)(end - TILE_ROW0(ti));
        bool any = false;
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = half * 128 + 16 * a + 4 * lg + r < nrows;
                const float v = valid ? acc[a][r] : -INFINITY;
                m = fmaxf(m, v);
                any |= v > th;
            }
        qm = fmaxf(qm, m);
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) npass += (half * 128 + 16 * a + 4 * lg + r < nrows && acc[a][r] > th) ? 1u : 0u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicMax(&p.qmax[16 * qf + lr], okey(qm));
    if (npass) atomicAdd(p.npass, (unsigned long long)npass);
}

// reference: per-query maximum of plain fp32 dot products (one wave per row)
__global__ void ref_kernel(const uint16_t* slab, int64_t N, const uint16_t* q16, unsigned* qmax, float theta, unsigned long long* npass) {
    __shared__ _Float16 sq[64 * D];
    for (int i = threadIdx.x; i < 64 * D; i += blockDim.x) sq[i] = ((const _Float16*)q16)[i];   // (q16 points at the 64 queries of this launch)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float best = -INFINITY; unsigned np = 0;           // lane = query
    for (int64_t r = (int64_t)blockIdx.x * nw + wave; r < N; r += (int64_t)gridDim.x * nw) {
        const _Float16* row = (const _Float16*)slab + r * D;
        float s = 0.f;
        for (int e = 0; e < D; ++e) s += (float)row[e] * (float)sq[lane * D + e];
        best = fmaxf(best, s); np += s > theta;
    }
    atomicMax(&qmax[lane], okey(best));
    if (np) atomicAdd(npass, (unsigned long long)np);
}

__global__ void fill_kernel(uint16_t* p, int64_t n, uint64_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = ((uint64_t)i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        // a value in (-1/16, 1/16) with a random mantissa: sum of 12 bits - offset, crude bell shape; row norm ~ 1
        const float u = ((float)(x & 0xffff) + (float)((x >> 16) & 0xffff) + (float)((x >> 32) & 0xffff) - 98302.5f) * (1.0f / 65536.f) * 0.0722f;
        p[i] = __builtin_bit_cast(uint16_t, (_Float16)u);
    }
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 32000000ll;
    const int iters = argc > 2 ? atoi(argv[2]) : 8;
    const int64_t check_rows = argc > 3 ? atoll(argv[3]) : 1000000ll;
    uint16_t *slab, *q16; unsigned *qmax, *qmax_ref; unsigned long long *np, *np_ref;
    if (hipMalloc(&slab, (size_t)N * ROWB) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&q16, NQ * ROWB); hipMalloc(&qmax, NQ * 4); hipMalloc(&qmax_ref, NQ * 4); hipMalloc(&np, 8); hipMalloc(&np_ref, 8);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, slab, N * D, 1ull);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, q16, (int64_t)NQ * D, 0x1234567ull);
    hipDeviceSynchronize();
    const int G = 256;
    const int64_t tiles = (N + TILE - 1) / TILE, rows_per_wg = ((tiles + G - 1) / G) * TILE;
    const int lds = NSTAGE * STG;
    hipFuncSetAttribute((const void*)dscan_proto, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const float theta = 0.16f;
    // correctness on the first `check_rows` rows
    {
        const int64_t n = check_rows < N ? check_rows : N;
        const int64_t t = (n + TILE - 1) / TILE, rpw = ((t + G - 1) / G) * TILE;
        hipMemset(qmax, 0, NQ * 4); hipMemset(qmax_ref, 0, NQ * 4); hipMemset(np, 0, 8); hipMemset(np_ref, 0, 8);
        Params p = {slab, n, q16, theta, qmax, np, rpw};
        hipLaunchKernelGGL(dscan_proto, dim3(G), dim3(512), lds, 0, p);
        for (int c = 0; c < NQ / 64; ++c) hipLaunchKernelGGL(ref_kernel, dim3(1024), dim3(256), 0, 0, slab, n, q16 + (size_t)c * 64 * D, qmax_ref + c * 64, theta, np_ref);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        unsigned a[NQ], b[NQ]; unsigned long long na, nb;
        hipMemcpy(a, qmax, NQ * 4, hipMemcpyDeviceToHost); hipMemcpy(b, qmax_ref, NQ * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&na, np, 8, hipMemcpyDeviceToHost); hipMemcpy(&nb, np_ref, 8, hipMemcpyDeviceToHost);
        auto unkey = [](unsigned k) { unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; union { unsigned u; float f; } x; x.u = u; return x.f; };
        double worst = 0; for (int i = 0; i < NQ; ++i) worst = fmax(worst, fabs((double)unkey(a[i]) - (double)unkey(b[i])));
        printf("# check on %lld rows: per-query maxima agree within %.3g (q0: %.6f vs %.6f), scores above %.2f: %llu vs %llu (fp32 reference)\n",
               (long long)n, worst, unkey(a[0]), unkey(b[0]), theta, na, nb);
    }
    Params p = {slab, N, q16, theta, qmax, np, rows_per_wg};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(dscan_proto, dim3(G), dim3(512), lds, 0, p); hipDeviceSynchronize();
    float sum = 0, best = 1e30f;
    for (int i = 0; i < iters; ++i) {
        hipEventRecord(e0, 0); hipLaunchKernelGGL(dscan_proto, dim3(G), dim3(512), lds, 0, p); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
    }
    const double bytes = (double)N * ROWB;
    printf("dscan_proto NSTAGE=%d aux=%d issue4=%d deal=%d  %lld rows x %d queries: mean %.4f ms (min %.4f)  %.3f TB/s = %.3f of 8 TB/s\n", NSTAGE, DMA_AUX, ISSUE4, DEAL, (long long)N, NQ,
           sum / iters, best, bytes / (sum / iters * 1e-3) / 1e12, bytes / (sum / iters * 1e-3) / 8e12);
    return 0;
}
