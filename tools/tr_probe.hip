// tr_probe.hip -- what ds_read_b64_tr_b16 returns on gfx950 (round 5 probe for an attention kernel that keeps V row-major in LDS).
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe && tools/tr_probe
// LDS holds u16 element i at byte 2 i. Test 1: lane-linear addresses (lane l -> byte 8 l). Test 2: the addresses a P.V B-operand read would use over a
// row-major V tile [key][64 dims + pad] (row pitch 144 B): group g = lane >> 4 reads keys 4 g .. 4 g + 3, dims 16 df .. 16 df + 15:
// lane s = lane & 15 supplies key 4 g + (s >> 2), dims 16 df + 4 (s & 3) .. + 3. Expected for an MFMA B operand: lane (c = lane & 15, g) gets
// V[4 g + j][16 df + c], j = 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(unsigned long long* out, int pitch_bytes, int df) {
    extern __shared__ uint16_t lds[];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, s = l & 15, g = l >> 4;
    const uint32_t a1 = (uint32_t)(l * 8);
    const uint32_t a2 = (uint32_t)((4 * g + (s >> 2)) * pitch_bytes + (16 * df + 4 * (s & 3)) * 2);
    unsigned long long r1, r2;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r1) : "v"(a1) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(a2) : "memory");
    out[l] = r1; out[64 + l] = r2;
}
int main() {
    unsigned long long *d, h[128];
    hipMalloc(&d, sizeof(h));
    const int pitch = 144, df = 2;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 32768, 0, d, pitch, df);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 4; ++j) {
            const int got1 = (int)((h[l] >> (16 * j)) & 0xffff), want1 = (l & 15) + 16 * j + 64 * (l >> 4);
            const int got2 = (int)((h[64 + l] >> (16 * j)) & 0xffff), want2 = ((4 * (l >> 4) + j) * pitch + (16 * df + (l & 15)) * 2) / 2;
            bad1 += got1 != want1; bad2 += got2 != want2;
            if (l < 20 || got1 != want1 || got2 != want2) printf("lane %2d elem %d: lane-linear got %5d want %5d | V-tile got %5d want %5d%s\n", l, j, got1, want1, got2, want2, (got1 != want1 || got2 != want2) ? "  <--" : "");
        }
    }
    printf("mismatches: lane-linear %d, V-tile %d of 256 each\n", bad1, bad2);
    return 0;
}
