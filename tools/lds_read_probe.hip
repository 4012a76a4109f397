// lds_read_probe.hip -- what a READ PHASE of the refresh GEMM / the GEMM-shaped scan costs by itself (round 5).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_read_probe.hip -o tools/lds_read_probe && tools/lds_read_probe
// tools/pt_cycles.py: with MFMAs and LDS-DMA pieces switched off, an iteration of gemm_pt_kernel (group A: 24 ds_read_b128 per wave, barrier,
// group B: the same, barrier) still takes ~2 000 shader cycles, i.e. ~1 000 per phase for 4 waves x 24 KiB = 96 B/clk/CU, against the 256 B/clk
// the LDS delivers for ds_read_b128. This probe runs exactly that phase structure on all 256 CUs (8 waves, the kernel's swizzled fragment
// addresses, 160 KiB allocated) and varies one thing at a time:
//   mode 0  ping-pong as in the kernel: waves 0-3 read 24 fragments, s_barrier, waves 4-7 read, s_barrier
//   mode 1  all 8 waves read their 24 fragments, one s_barrier per iteration
//   mode 2  ping-pong, 12 fragments per phase                         (does the phase scale with the reads?)
//   mode 3  ping-pong, lane-linear addresses (1 KiB contiguous per instruction) instead of the swizzled fragment pattern
//   mode 4  ping-pong, no barrier at all (each group free-running)    (what the barriers cost)
//   mode 5  ping-pong, the 24 reads as 3 blocks of 8 with s_waitcnt lgkmcnt(0) after each   (is the 4-bit lgkmcnt the limit?)
//   mode 6  ping-pong, 48 x ds_read_b64 of the same bytes
// and the LDS allocation (160 KiB / 64 KiB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

#define RD8(dst, base, off0)                                                                                                        \
    asm volatile("ds_read_b128 %0, %8 offset:%c9\n ds_read_b128 %1, %8 offset:%c9+2048\n ds_read_b128 %2, %8 offset:%c9+4096\n"          \
                 "ds_read_b128 %3, %8 offset:%c9+6144\n ds_read_b128 %4, %8 offset:%c9+8192\n ds_read_b128 %5, %8 offset:%c9+10240\n"    \
                 "ds_read_b128 %6, %8 offset:%c9+12288\n ds_read_b128 %7, %8 offset:%c9+14336"                                           \
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]) \
                 : "v"(base), "n"(off0) : "memory")
#define RD4(dst, base)                                                                                                              \
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4 offset:4096\n ds_read_b128 %3, %4 offset:6144" \
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(base) : "memory")
#define WAIT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int MODE>
__global__ void __launch_bounds__(512) probe(int iters, unsigned long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;
    for (int i = tid; i < 16384; i += 512) ((uint32_t*)smem)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    const int lr = lane & 15, lg = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // the kernel's addresses inside one 64 KiB window (W stage 0 at 0, activation stage at 32 KiB): fragment a adds a * 2048
    uint32_t w0 = lds0 + (wi * 128 + lr) * 128 + ((lg) ^ (lr & 7)) * 16, a0 = lds0 + 32768 + (wj * 64 + lr) * 128 + ((lg) ^ (lr & 7)) * 16;
    if (MODE == 3) { w0 = lds0 + wi * 16384 + lane * 16; a0 = lds0 + 32768 + wj * 8192 + lane * 16; }      // (fragment a: + 2048 = the next two KiB)
    const uint32_t w1 = w0 ^ 64u, a1 = a0 ^ 64u;
    u4 fw0[8], fw1[8], fa0[4], fa1[4];
    uint32_t acc = 0;
    const bool grpB = wave >= 4;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        for (int ph = 0; ph < 2; ++ph) {
            const bool mine = (MODE == 1) ? (ph == 0) : ((ph == 1) == grpB);
            if (mine) {
                if (MODE == 6) {
                    u2 t[48];
#pragma unroll
                    for (int k = 0; k < 48; ++k) {
                        const uint32_t b = (k < 16 ? w0 : k < 24 ? a0 : k < 40 ? w1 : a1);
                        const int kk = (k < 16 ? k : k < 24 ? k - 16 : k < 40 ? k - 24 : k - 40);
                        asm volatile("ds_read_b64 %0, %1 offset:%c2" : "=&v"(t[k]) : "v"(b), "n"((kk >> 1) * 2048 + (kk & 1) * 8) : "memory");
                    }
                    WAIT0();
#pragma unroll
                    for (int k = 0; k < 48; ++k) acc ^= t[k].x ^ t[k].y;
                } else {
                    RD8(fw0, w0, 0);
                    if (MODE == 5) WAIT0();
                    RD4(fa0, a0);
                    if (MODE != 2) {
                        RD8(fw1, w1, 0);
                        if (MODE == 5) WAIT0();
                        RD4(fa1, a1);
                    }
                    WAIT0();
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc ^= fw0[k].x ^ (MODE != 2 ? fw1[k].w : 0u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc ^= fa0[k].y ^ (MODE != 2 ? fa1[k].z : 0u);
                }
            }
            if (MODE != 4 && !(MODE == 1 && ph == 1)) __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) sink[0] = acc;
    if (tid == 0) out[blockIdx.x] = c1 - c0;
}

template <int MODE>
static void run(const char* what, int lds_bytes, int bytes_per_iter) {
    unsigned long long* d; uint32_t* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 64);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), lds_bytes, 0, iters, d, sink);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double cyc = s / 256 / iters;
    printf("%-86s LDS %3d KiB  %7.0f cycles per iteration  %6.1f B/clk/CU\n", what, lds_bytes >> 10, cyc, bytes_per_iter / cyc);
    hipFree(d); hipFree(sink);
}

int main() {
    const int full = 8 * 24 * 1024;
    run<0>("0 ping-pong: A 24 x ds_read_b128, barrier, B 24 x ds_read_b128, barrier (the kernel's phases)", 160 * 1024, full);
    run<0>("0 the same", 64 * 1024, full);
    run<1>("1 all 8 waves read 24 fragments, one barrier", 160 * 1024, full);
    run<2>("2 ping-pong, 12 fragments per phase", 160 * 1024, full / 2);
    run<3>("3 ping-pong, lane-linear addresses", 160 * 1024, full);
    run<4>("4 ping-pong order, no barrier", 160 * 1024, full);
    run<5>("5 ping-pong, lgkmcnt(0) after every 8 reads", 160 * 1024, full);
    run<6>("6 ping-pong, 48 x ds_read_b64 of the same bytes", 160 * 1024, full);
    return 0;
}
