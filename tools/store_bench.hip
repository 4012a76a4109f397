// What does a GEMM epilogue's store burst cost, and what bounds it? (dev tool, round 3)
// Every workgroup (512 threads = 8 waves, one per CU) writes `iters` tiles of 256 x 256 fp16 (128 KiB) of a [M][768] tensor with 16-byte
// stores, then idles `gap_us` (the k-loop of the next tile), all workgroups in step -- as the refresh GEMMs do. Reported per burst: the time
// until the last store is ISSUED (what the wave is held up for) and until all are complete (vmcnt(0)), for
//   pattern 0: a wave instruction = 16 rows x 64 B   (the register epilogue of gemm_pt_kernel)
//   pattern 1: a wave instruction =  2 rows x 512 B  (the LDS-transposed epilogue of gemm_pp_kernel)
//   G = 1, 8, 32, 256 active workgroups (is it the CU's own store path or the chip's write bandwidth?)
//   policy 0 / nt / sc1
//   hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o tools/store_bench && tools/store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int AUX>
__global__ void __launch_bounds__(512) burst(uint16_t* C, long long rows, int iters, int gap_ticks, unsigned long long* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3, lr = lane & 15, lg = lane >> 4;
    unsigned long long t_issue = 0, t_done = 0;
    u4 v = {(unsigned)tid, 0x3c003c00u, (unsigned)blockIdx.x, 0x38003800u};
    for (int it = 0; it < iters; ++it) {
        const long long m0 = (((long long)blockIdx.x * iters + it) * 256) % (rows - 256);
        const int n0 = (it % 3) * 256;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * 768), 0, 256 * 768 * 2, 0x00020000);
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned vo;
                if (PATTERN == 0) vo = (unsigned)(((wj * 64 + b * 16 + lr) * 768 + n0 + wi * 128 + 32 * j + 8 * lg) * 2);
                else vo = (unsigned)(((wave * 32 + (b * 4 + j) * 2 + (lane >> 5)) * 768 + n0 + 8 * (lane & 31)) * 2);
                v.x += 1;
                __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)vo, 0, AUX);
            }
        const unsigned long long t1 = wall_clock64();
        __builtin_amdgcn_s_waitcnt(0x0F70);
        const unsigned long long t2 = wall_clock64();
        t_issue += t1 - t0; t_done += t2 - t0;
        const unsigned long long until = t2 + gap_ticks;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    // slowest wave of the workgroup
    __shared__ unsigned long long s[2];
    if (tid == 0) { s[0] = 0; s[1] = 0; }
    __syncthreads();
    if (lane == 0) { atomicMax(&s[0], t_issue); atomicMax(&s[1], t_done); }
    __syncthreads();
    if (tid == 0) { out[blockIdx.x * 2] = s[0]; out[blockIdx.x * 2 + 1] = s[1]; }
}

template <int PATTERN, int AUX> void run(uint16_t* C, long long rows, unsigned long long* out, int G, int gap_us) {
    const int iters = 64;
    std::vector<unsigned long long> h(512);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((burst<PATTERN, AUX>), dim3(G), dim3(512), 0, 0, C, rows, iters, gap_us * 100, out);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), out, G * 16, hipMemcpyDeviceToHost);
    double is = 0, dn = 0, ismax = 0, dnmax = 0;
    for (int g = 0; g < G; ++g) { is += h[2 * g]; dn += h[2 * g + 1]; ismax = std::max(ismax, (double)h[2 * g]); dnmax = std::max(dnmax, (double)h[2 * g + 1]); }
    printf("pattern %d  policy %2d  G %3d  gap %2d us:  issued %6.2f us (max %6.2f)   complete %6.2f us (max %6.2f)  per 128 KiB burst and workgroup -> %6.1f GB/s per CU, %5.2f TB/s chip\n",
           PATTERN, AUX, G, gap_us, is / G / iters / 100.0, ismax / iters / 100.0, dn / G / iters / 100.0, dnmax / iters / 100.0,
           131072.0 / (dn / G / iters / 100.0 * 1e-6) / 1e9, G * 131072.0 / (dn / G / iters / 100.0 * 1e-6) / 1e12);
}

int main() {
    const long long rows = 8ll << 20;                      // [8M][768] fp16 = 12.9 GB: bursts never meet a line they wrote before in cache
    uint16_t* C; unsigned long long* out;
    (void)hipMalloc(&C, rows * 768 * 2); (void)hipMalloc(&out, 8192);
    (void)hipMemset(C, 0, rows * 768 * 2);
    for (int gap : {20, 0}) {
        for (int G : {1, 8, 32, 256}) {
            run<0, 0>(C, rows, out, G, gap);
            run<1, 0>(C, rows, out, G, gap);
        }
        run<0, 2>(C, rows, out, 256, gap);
        run<1, 2>(C, rows, out, 256, gap);
        run<0, 16>(C, rows, out, 256, gap);
        run<1, 16>(C, rows, out, 256, gap);
    }
    return 0;
}
