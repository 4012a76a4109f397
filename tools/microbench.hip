// microbench.hip — tuning harness (NOT part of the product library): streams the slab with different
// access patterns and runs scan_kernel variants, timing each with hipEvents.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../atlas_amd/csrc/scan_kernel.h"
using namespace atlas;

// ---- pure streaming kernels: what can the load path deliver, without MFMA/LDS/filter? ----
// pattern 0: fully coalesced (wave instruction = 1 KiB contiguous)
// pattern 1: MFMA-A fragment shape of the scan (16 rows x 64 B per instruction), ring of R steps
// pattern 2: fragment shape, both 64-B halves of every 128-B line issued back to back
template <int PATTERN, int PF, int R>
__global__ void __launch_bounds__(512) stream_kernel(const unsigned char* slab, int64_t N, int64_t rows_per_wg, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
    int64_t r_end = r_begin + rows_per_wg; if (r_end > N) r_end = N;
    constexpr int ROWB = 1536, TILE = 8 * PF * 16;
    const int64_t wrow0 = r_begin + wave * PF * 16;
    int64_t span = (wrow0 < N) ? (N - wrow0) * (int64_t)ROWB : 0;
    if (span > 0xfffffff0ll) span = 0xfffffff0ll;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(slab + (span > 0 ? wrow0 : 0) * (int64_t)ROWB), 0, (int)span, 0x00020000);
    const int ntiles = (r_end > r_begin) ? (int)((r_end - r_begin + TILE - 1) / TILE) : 0;
    u32x4 acc = {0, 0, 0, 0};
    if (PATTERN == 0) {
        // each wave-tile = PF*16 rows = PF*16*1536 B contiguous; lane reads 16 B, instruction = 1 KiB
        unsigned vo = lane * 16;
        constexpr int NLD = PF * 16 * ROWB / 1024;   // loads per wave-tile
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll 8
            for (int i = 0; i < NLD; ++i) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo, i * 1024, 0);
                acc ^= v;
            }
            vo += TILE * ROWB;
        }
    } else {
        unsigned vo[PF];
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) vo[pf] = (unsigned)((pf * 16 + (lane & 15)) * ROWB + (lane >> 4) * 16);
        for (int t = 0; t < ntiles; ++t) {
            if (PATTERN == 1) {
#pragma unroll R
                for (int s = 0; s < 24; ++s) {
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], s * 64, 0);
                }
            } else {
#pragma unroll R
                for (int s = 0; s < 24; s += 2) {
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], s * 64, 0);
                        acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[pf], s * 64 + 64, 0);
                    }
                }
            }
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) vo[pf] += TILE * ROWB;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}

template <typename F>
static float time_ms(F launch, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / iters;
}

extern "C" float mb_stream(int pattern, int ring, const void* slab, int64_t N, unsigned* out, int iters) {
    int64_t frags = (N + 15) / 16, G = 256;
    int64_t rows_per_wg = ((frags + G - 1) / G) * 16;
    auto go = [&](auto kern) { return time_ms([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(512), 0, 0, (const unsigned char*)slab, N, rows_per_wg, out); }, iters); };
    if (pattern == 0) return go(stream_kernel<0, 4, 8>);
    if (pattern == 1 && ring == 4) return go(stream_kernel<1, 4, 4>);
    if (pattern == 1 && ring == 8) return go(stream_kernel<1, 4, 8>);
    if (pattern == 1 && ring == 24) return go(stream_kernel<1, 4, 24>);
    if (pattern == 2 && ring == 4) return go(stream_kernel<2, 4, 4>);
    if (pattern == 2 && ring == 12) return go(stream_kernel<2, 4, 12>);
    return -1.f;
}

// ---- scan_kernel variants ----
struct Plan { int G; int64_t rows_per_wg; int keep_max, cap, buf_cap; size_t lds; };
template <int NW, int PF>
static Plan plan(int64_t N, int k) {
    Plan p; int64_t frags = (N + 15) / 16; int64_t G = 256;
    if (G > (frags + NW - 1) / NW) G = (frags + NW - 1) / NW; if (G < 1) G = 1;
    p.G = (int)G; p.rows_per_wg = ((frags + G - 1) / G) * 16;
    p.keep_max = (2 * k > k + 64) ? 2 * k : k + 64; p.buf_cap = 4096; p.cap = p.keep_max + p.buf_cap + NW * PF * 16;
    p.lds = (size_t)ScanSmem::buf_off + (size_t)p.buf_cap * 8;
    return p;
}

template <int NW, int PF, int RING>
static float run_scan(const void* slab, int64_t N, const void* qfrag, const void* qeps, const void* theta0, void* ws, int nq, int k, int iters) {
    Plan pl = plan<NW, PF>(N, k);
    unsigned char* w = (unsigned char*)ws;
    ScanParams sp{};
    sp.slab = (const uint16_t*)slab; sp.N = N; sp.qfrag = (const uint4*)qfrag; sp.qeps = (const float*)qeps; sp.theta0 = (const float*)theta0;
    sp.gstat = (uint32_t*)w; sp.qflag = (uint32_t*)(w + 256); sp.dense_cnt = (uint32_t*)(w + 512);
    sp.dense = (uint2*)(w + 1024); sp.dense_cap = 32768;
    sp.lists = (uint2*)(w + 1024 + 64 * 32768 * 8);
    sp.rows_per_wg = pl.rows_per_wg; sp.nq = nq; sp.k = k; sp.cap = pl.cap; sp.keep_max = pl.keep_max; sp.buf_cap = pl.buf_cap; sp.pmax2_hint = 4.0f;
    auto kern = scan_kernel<NW, PF, RING>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return time_ms([&] { hipMemsetAsync(w, 0, 1024, 0); hipLaunchKernelGGL(kern, dim3(pl.G), dim3(NW * 64), pl.lds, 0, sp); }, iters);
}

extern "C" float mb_scan(int variant, const void* slab, int64_t N, const void* qfrag, const void* qeps, const void* theta0, void* ws, int nq, int k, int iters) {
    switch (variant) {
        case 0: return run_scan<8, 4, 4>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 1: return run_scan<8, 4, 3>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 2: return run_scan<8, 4, 6>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 3: return run_scan<8, 2, 4>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 4: return run_scan<8, 2, 8>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 5: return run_scan<12, 2, 4>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 6: return run_scan<16, 2, 4>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 7: return run_scan<16, 1, 8>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
        case 8: return run_scan<4, 4, 6>(slab, N, qfrag, qeps, theta0, ws, nq, k, iters);
    }
    return -1.f;
}
