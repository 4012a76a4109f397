// hipcc (ROCm 7.2.0) and the gfx950 row swaps: kernel k uses __builtin_amdgcn_permlane16_swap and adds its two results -- the emitted code adds the FIRST result
// to itself (v_add v1, v1, v1); kernel k2 does the same with inline asm and is right. Look at the ISA:
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o - tools/permlane_probe.hip | grep -E "permlane|v_add"
#include <hip/hip_runtime.h>
__global__ void k(float* p, float* q) {
    float x = p[threadIdx.x], y = q[threadIdx.x];
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    p[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k2(float* p) {
    float x = p[threadIdx.x];
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    p[threadIdx.x] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
