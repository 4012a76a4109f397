// Is "buffer_store_dwordx4 with an SGPR soffset, then a VALU write of one of its data VGPRs in the very next instruction" safe on gfx950?
// hipcc's hazard recognizer inserts the wait state only for an immediate / absent soffset. The tuning build of gemm_pt_kernel had that
// sequence in two of its three epilogue variants, and exactly those two returned NaNs (round 3, session b).
// Modes: 0 = SGPR soffset, data VGPRs overwritten at once | 1 = SGPR soffset, one s_nop between | 2 = immediate soffset, overwritten at once
//        3 = immediate soffset, s_nop 0 between | 4 = immediate soffset, s_nop 1 between (what hipcc emits: two wait states)
// Prints the number of 16-byte stores that reached memory corrupted.
// Measured (profiles/r03/hazard_probe.txt): mode 0 corrupts 1.7 % of the stores -- hipcc's recognizer is wrong for gfx950 there --, mode 1
// none, mode 2 23 %, mode 3 still 0.9 % (ONE wait state is not enough behind an immediate soffset).
//   hipcc --offload-arch=gfx950 -O3 tools/hazard_probe.hip -o tools/hazard_probe && tools/hazard_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define STORE_AND_CLOBBER(SOFF, NOP)                                                                                           \
    asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"          \
                 "buffer_store_dwordx4 v[10:13], %0, %1, " SOFF " offen\n\t" NOP                                              \
                 "v_mov_b32 v11, 0xdeadbeef\n\tv_mov_b32 v10, 0xdeadbeef\n\tv_mov_b32 v13, 0xdeadbeef\n\tv_mov_b32 v12, 0xdeadbeef"   \
                 :: "v"(vo), "s"(r), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff) : "v10", "v11", "v12", "v13", "memory")

template <int MODE>
__global__ void __launch_bounds__(512) probe(uint32_t* out, int soff, int iters) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffff0, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const uint32_t idx = (uint32_t)((it * gridDim.x + blockIdx.x) * 512 + threadIdx.x);
        const uint32_t vo = idx * 16 + (MODE < 2 ? 0 : soff);              // every mode writes at byte idx * 16 + soff
        const uint32_t a = 0x11000000u + idx, b = 0x22000000u + idx, c = 0x33000000u + idx, d = 0x44000000u + idx;
        if (MODE == 0) STORE_AND_CLOBBER("%6", "");
        else if (MODE == 1) STORE_AND_CLOBBER("%6", "s_nop 0\n\t");
        else if (MODE == 2) STORE_AND_CLOBBER("0", "");
        else if (MODE == 3) STORE_AND_CLOBBER("0", "s_nop 0\n\t");
        else STORE_AND_CLOBBER("0", "s_nop 1\n\t");
    }
}

template <int MODE> void run(uint32_t* dev, size_t n16) {
    const int G = 1024, iters = (int)(n16 / (G * 512));
    (void)hipMemset(dev, 0, n16 * 16 + 4096);
    hipLaunchKernelGGL(probe<MODE>, dim3(G), dim3(512), 0, 0, dev, 128, iters);
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> h(n16 * 4);
    (void)hipMemcpy(h.data(), (char*)dev + 128, n16 * 16, hipMemcpyDeviceToHost);
    size_t bad = 0, beef = 0;
    for (size_t i = 0; i < (size_t)G * 512 * iters; ++i) {
        const uint32_t idx = (uint32_t)i;
        const bool ok = h[4 * i] == 0x11000000u + idx && h[4 * i + 1] == 0x22000000u + idx && h[4 * i + 2] == 0x33000000u + idx && h[4 * i + 3] == 0x44000000u + idx;
        bad += !ok;
        for (int e = 0; e < 4; ++e) beef += h[4 * i + e] == 0xdeadbeefu;
    }
    printf("mode %d: %zu of %zu stores corrupted (%zu words read 0xdeadbeef)\n", MODE, bad, (size_t)G * 512 * iters, beef);
}

int main() {
    const size_t n16 = (size_t)1024 * 512 * 16;           // 128 MiB
    uint32_t* dev; (void)hipMalloc(&dev, n16 * 16 + 4096);
    run<0>(dev, n16); run<1>(dev, n16); run<2>(dev, n16); run<3>(dev, n16); run<4>(dev, n16);
    return 0;
}
