// encoder.hip — Contriever (BERT-base) passage encoder for the index-refresh path, gfx950.
//
// Replaces what `copy.deepcopy(retriever).half().eval()` runs inside Atlas.build_index (src/atlas.py:54-59, 78):
// src/retrievers.py:22-60 on top of src/modeling_bert.py (BertEmbeddings :213-247, BertSelfAttention :290-366,
// BertSelfOutput :382-387, BertIntermediate :448-451, BertOutput :461-466, BertLayerNorm :104-114).
// Numerics follow the fp16 copy op by op: every tensor the reference materialises in fp16 is rounded to fp16 here at
// the same place (GEMM outputs after the fp32 bias add, the residual sums, softmax probabilities, GELU outputs, both
// steps of `weight * y + bias` in the NON-standard LayerNorm); softmax and LayerNorm statistics are fp32. What may
// differ from a given torch backend is only the fp32 summation order inside GEMMs and reductions.
//
// Token packing: only tokens with attention_mask != 0 are computed. Masked keys get probability exactly 0 in the
// reference (exp(-10000 + s - max) underflows to 0 in fp32) and masked tokens' hidden states are dropped by the
// pooling (retrievers.py:50), so leaving them out changes no result; a batch of queries padded to 512 with ~20 real
// tokens costs 20 tokens. The packing is done on the device (count_kernel + pack_kernel, no host sync): kernels are
// launched for the worst case n*L tokens and blocks beyond the packed count T = cu[n] exit immediately.
//
// Kernels
//   count_kernel / pack_kernel   per-passage real-token counts, exclusive scan cu[n+1], tokinfo[t] = (passage, position)
//   embed_ln_kernel     word + type (+= position) in fp16, LayerNorm                    (one wave per token)
//   gemm_bt_kernel      C[M,N] = A[M,K] . W[N,K]^T + bias, 128x128x64 tiles, MFMA 16x16x32 f16, LDS double buffer,
//                       epilogues: plain | exact-erf GELU | + residual        (MFMA-bound: the refresh roofline)
//   attention_kernel    per (passage, head): QK^T -> fp16 -> /8 + mask -> fp32 softmax -> fp16 P -> PV
//   ln_kernel           the reference's LayerNorm on a [M,768] fp16 tensor          (one wave per token)
//   pool_packed_kernel  mean over a passage's tokens with the reference's two roundings, row written into the slab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "common.h"
#include "../../include/atlas_hip.h"

using namespace atlas;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define HID 768
#define NHEAD 12
#define DHEAD 64

static __device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
static __device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // v_cvt_f16_f32, RNE
static __device__ __forceinline__ float rh(float f) { return (float)(_Float16)f; }                               // round through fp16
static __device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// ---- the reference's LayerNorm on one 768-vector held 12 per lane (element 64*i + lane) ----
// modeling_bert.py:104-114: mean and UNCENTRED second moment in fp32, y = fp16((x-mean)*rsqrt(E[x^2]+eps)),
// out = fp16(fp16(w*y) + b)
static __device__ __forceinline__ void layer_norm_768(const float (&x)[12], const uint16_t* __restrict__ w,
                                                      const uint16_t* __restrict__ b, float eps, int lane,
                                                      uint16_t* __restrict__ out) {
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { s += x[i]; s2 += x[i] * x[i]; }
    s = wave_sum(s); s2 = wave_sum(s2);
    const float mean = s * (1.0f / HID), var = s2 * (1.0f / HID);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = i * 64 + lane;
        const float y = rh((x[i] - mean) * rstd);
        out[c] = f2h(rh(h2f(w[c]) * y) + h2f(b[c]));
    }
}

// ---- token packing ----
// counts[b] = number of tokens with mask != 0 (one wave per passage)
__global__ void __launch_bounds__(256)
count_kernel(const int64_t* __restrict__ mask, int n, int L, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    int c = 0;
    for (int l = lane; l < L; l += 64) c += (mask[(size_t)b * L + l] != 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) counts[b] = c;
}

// cu[b] = sum counts[0..b) (every wave sums its own prefix: n is a few thousand at most and counts is L2-resident),
// tokinfo[cu[b] + rank] = (b, l) for the real tokens of passage b in position order
__global__ void __launch_bounds__(256)
pack_kernel(const int64_t* __restrict__ mask, int n, int L, const int* __restrict__ counts, int* __restrict__ cu,
            int2* __restrict__ tokinfo) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    int base = 0;
    for (int i = lane; i < b; i += 64) base += counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    if (lane == 0) {
        cu[b] = base;
        if (b == n - 1) cu[n] = base + counts[b];
    }
    int run = base;
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        const bool real = (l < L) && (mask[(size_t)b * L + l] != 0);
        const unsigned long long bal = __ballot(real);
        if (real) tokinfo[run + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(b, l);
        run += __popcll(bal);
    }
}

// one wave per packed token: embeddings (modeling_bert.py:213-247)
__global__ void __launch_bounds__(256)
embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids, int L, const int* __restrict__ cu, int n,
                const int2* __restrict__ tokinfo,
                const uint16_t* __restrict__ word, const uint16_t* __restrict__ pos, const uint16_t* __restrict__ type,
                const uint16_t* __restrict__ lnw, const uint16_t* __restrict__ lnb, float eps, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= cu[n]) return;
    const int2 ti = tokinfo[t];
    const size_t src = (size_t)ti.x * L + ti.y;
    const int64_t id = ids[src], ty = type_ids ? type_ids[src] : 0;
    const int p = ti.y;
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = i * 64 + lane;
        const float e = rh(h2f(word[id * HID + c]) + h2f(type[ty * HID + c]));    // inputs_embeds + token_type_embeddings (fp16)
        x[i] = rh(e + h2f(pos[(size_t)p * HID + c]));                              // embeddings += position_embeddings (fp16)
    }
    layer_norm_768(x, lnw, lnb, eps, lane, out + (size_t)t * HID);
}

// one wave per token: LayerNorm(x.float()).type_as(x) on an fp16 [M,768] tensor
__global__ void __launch_bounds__(256)
ln_kernel(const uint16_t* __restrict__ in, const int* __restrict__ Tdev, const uint16_t* __restrict__ lnw, const uint16_t* __restrict__ lnb,
          float eps, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= *Tdev) return;
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = h2f(in[(size_t)t * HID + i * 64 + lane]);
    layer_norm_768(x, lnw, lnb, eps, lane, out + (size_t)t * HID);
}

// ------------------------------------------------------------------------------------------
// GEMM: C[M,N] (fp16) = A[M,K] (fp16, row-major) . W[N,K]^T (fp16, row-major) + bias[N], fp32 accumulate.
// Computed transposed on the matrix cores (MFMA A operand = W rows, B operand = A rows) so that each lane ends up
// with 4 consecutive OUTPUT COLUMNS of one token: the epilogue reads/writes 8 contiguous bytes per fragment.
// 128 (cols) x 128 (tokens) x 64 tile, 4 waves as 2x2, each 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_f16.
// LDS: two 16 KB tiles per stage, 2 stages; 16-B chunks XOR-swizzled by (row & 7) so ds_read_b128 of 16 rows at one
// k-chunk spreads over 8 bank groups. Global->LDS through registers; the next stage's loads are issued before the
// current stage's MFMAs.
//   EPI 0: C = fp16(acc + bias)                 (QKV projection, torch Linear)
//   EPI 1: C = fp16(gelu_erf(fp16(acc + bias)))  (BertIntermediate)
//   EPI 2: C = fp16(fp16(acc + bias) + R)        (dense + residual of BertSelfOutput / BertOutput; LayerNorm follows)
// Requires N % 128 == 0, K % 64 == 0 (768, 2304, 3072 all are); M arbitrary.
// ------------------------------------------------------------------------------------------
template <int EPI, int BCOL, int BTOK, int WC, int WT>
__global__ void __launch_bounds__(WC * WT * 64)
gemm_bt_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias,
               const uint16_t* __restrict__ R, uint16_t* __restrict__ C, uint16_t* __restrict__ VT, const int* __restrict__ cu, int n,
               const int2* __restrict__ tokinfo, int N, int K, int Lp) {
    // tile: BCOL output columns x BTOK tokens x 64 (k); WC x WT waves, each (BCOL/WC) x (BTOK/WT) = FA x FB fragments
    constexpr int NWV = WC * WT, FA = BCOL / WC / 16, FB = BTOK / WT / 16;
    constexpr int W_U4 = BCOL * 8, A_U4 = BTOK * 8;                  // uint4 per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* sW = (uint4*)smem_raw;                                      // [2][W_U4]
    uint4* sA = sW + 2 * W_U4;                                         // [2][A_U4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WT, wj = wave % WT;
    // XCD-aware tile order (cdna guide T1): hardware block b runs on XCD b % 8, each XCD has its own L2. All column
    // tiles of one token tile are given to ONE XCD (token tile t -> XCD t % 8), so the big activation tile
    // (BTOK x K) is fetched into one L2 once instead of into up to 8; the weights (<= 4.7 MB) fit every L2.
    const int64_t M = cu[n];                                           // packed token count (grid covers n*L)
    const int ncol = N / BCOL;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ctile = jj % ncol;
    const int64_t ttile = (int64_t)(jj / ncol) * 8 + xcd;
    if (ttile * BTOK >= M) return;
    const int n0 = ctile * BCOL;
    const int64_t m0 = ttile * BTOK;
    const int lr = lane & 15, lg = lane >> 4;

    // global -> LDS without registers (global_load_lds_dwordx4): one wave instruction writes 1 KiB of LDS,
    // lane-linear, = 8 tile rows x 128 B. The bank-conflict swizzle therefore goes on the SOURCE address:
    // LDS[row][c] receives global chunk c ^ (row & 7); fragment reads apply the same XOR (cdna guide rule 21).
    auto stage = [&](const int buf, const int k0) {
        const int ch = (lane & 7) ^ (lane >> 3);
#pragma unroll
        for (int i = 0; i < BCOL / 8 / NWV; ++i) {
            const int rowbase = (wave * (BCOL / 8 / NWV) + i) * 8;
            const uint16_t* gw = W + (size_t)(n0 + rowbase + (lane >> 3)) * K + k0 + ch * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
                                             (__attribute__((address_space(3))) void*)&sW[buf * W_U4 + rowbase * 8], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BTOK / 8 / NWV; ++i) {
            const int rowbase = (wave * (BTOK / 8 / NWV) + i) * 8;
            int64_t ar = m0 + rowbase + (lane >> 3);
            if (ar >= M) ar = M - 1;                                   // clamped: tail rows are never stored
            const uint16_t* ga = A + (size_t)ar * K + k0 + ch * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)&sA[buf * A_U4 + rowbase * 8], 16, 0, 0);
        }
    };

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();                                   // (drains the LDS-DMA: hipcc puts vmcnt(0) in front of the barrier)
    const int nk = K / 64;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * 64);   // next tile streams in while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 fw[FA], fa[FB];
#pragma unroll
            for (int a = 0; a < FA; ++a) {
                const int row = wi * (FA * 16) + a * 16 + lr;
                fw[a] = __builtin_bit_cast(h8, sW[buf * W_U4 + row * 8 + ((ks * 4 + lg) ^ (row & 7))]);
            }
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int row = wj * (FB * 16) + b * 16 + lr;
                fa[b] = __builtin_bit_cast(h8, sA[buf * A_U4 + row * 8 + ((ks * 4 + lg) ^ (row & 7))]);
            }
#pragma unroll
            for (int a = 0; a < FA; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[a], fa[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: acc[a][b][r] = C[token m0 + wj*FB*16 + 16b + lr][col n0 + wi*FA*16 + 16a + 4lg + r]
    const bool v_part = (EPI == 3) && (n0 >= 2 * HID);      // QKV projection: the V columns are stored transposed
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int64_t tok = m0 + wj * (FB * 16) + b * 16 + lr;
        if (tok >= M) continue;
        int64_t pb = 0;
        int pos = 0;                                             // rank of the token inside its passage = V^T column
        if (EPI == 3 && v_part) { pb = tokinfo[tok].x; pos = (int)(tok - cu[pb]); }
#pragma unroll
        for (int a = 0; a < FA; ++a) {
            const int col = n0 + wi * (FA * 16) + a * 16 + lg * 4;
            const uint2 bb = *(const uint2*)(bias + col);
            const uint16_t bh[4] = {(uint16_t)(bb.x & 0xffff), (uint16_t)(bb.x >> 16), (uint16_t)(bb.y & 0xffff), (uint16_t)(bb.y >> 16)};
            uint16_t o[4];
            uint16_t rr[4] = {0, 0, 0, 0};
            if (EPI == 2) {
                const uint2 rv = *(const uint2*)(R + (size_t)tok * N + col);
                rr[0] = (uint16_t)(rv.x & 0xffff); rr[1] = (uint16_t)(rv.x >> 16); rr[2] = (uint16_t)(rv.y & 0xffff); rr[3] = (uint16_t)(rv.y >> 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = rh(acc[a][b][r] + h2f(bh[r]));                      // Linear output in fp16
                if (EPI == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // exact-erf GELU in fp32
                if (EPI == 2) v = rh(v) + h2f(rr[r]);                           // + input_tensor, fp16 add
                o[r] = f2h(v);
            }
            if (EPI == 3) {
                if (v_part) {
                    // V^T[passage][h*64+d][key]: the PV product wants consecutive KEYS per lane (attention_kernel)
#pragma unroll
                    for (int r = 0; r < 4; ++r) VT[((size_t)pb * HID + (col - 2 * HID + r)) * Lp + pos] = o[r];
                } else {
                    *(uint2*)(C + (size_t)tok * (2 * HID) + col) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
                }
            } else {
                *(uint2*)(C + (size_t)tok * N + col) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
            }
        }
    }
}

// tile configurations (ATLAS_GEMM_CFG selects at run time; tuning)
template <int EPI>
static void launch_gemm(int cfg, hipStream_t stream, const uint16_t* A, const uint16_t* W, const uint16_t* bias, const uint16_t* R,
                        uint16_t* C, uint16_t* VT, int64_t Mmax, const int* cu, int n, const int2* tokinfo, int N, int K, int Lp) {
    auto go = [&](auto kern, int bcol, int btok, int nthreads) {
        const size_t lds = (size_t)(bcol + btok) * 128 * 2;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const unsigned mtiles = (unsigned)((Mmax + btok - 1) / btok);
        hipLaunchKernelGGL(kern, dim3((mtiles + 7) / 8 * 8 * (N / bcol)), dim3(nthreads), lds, stream, A, W, bias, R, C, VT,
                           cu, n, tokinfo, N, K, Lp);
    };
    if (cfg == 1) go(gemm_bt_kernel<EPI, 256, 128, 4, 2>, 256, 128, 512);
    else if (cfg == 2) go(gemm_bt_kernel<EPI, 256, 256, 2, 4>, 256, 256, 512);
    else if (cfg == 3) go(gemm_bt_kernel<EPI, 128, 256, 2, 4>, 128, 256, 512);
    else go(gemm_bt_kernel<EPI, 128, 128, 2, 2>, 128, 128, 256);
}

// ------------------------------------------------------------------------------------------
// attention: one block (4 waves) per (passage, head). QK is [M][1536] (q | k), VT is [n][768][Lp] (V transposed).
//   scores = fp16(q.k^T) ; / 8 (exact) ; + fp16 mask (0 / -10000) ; softmax in fp32 ; P = fp16 ; ctx = fp16(P.v)
// Everything stays in registers:
//   S^T = K.Q^T on the matrix cores (A = K rows, B = Q rows) leaves lane (lr, lg) with query column lr and keys
//   16kf+4lg+r -- so a softmax row is spread over only 4 lanes (xor 16, 32), and the SAME registers are already the
//   A operand of P.V if the contraction index of that MFMA is enumerated as key(ks,g,e) = 32ks + 16(e/4) + 4g + e%4;
//   V^T then supplies the matching B operand with two 8-byte loads of consecutive keys. No LDS round trip for P, no
//   transposition of V here (the QKV GEMM epilogue wrote V^T).
// L <= 512; Lp = L rounded up to 32; MAXKF = compile-time bound on Lp/16.
// ------------------------------------------------------------------------------------------
template <int MAXKF>
__global__ void __launch_bounds__(256)
attention_kernel(const uint16_t* __restrict__ qk, const uint16_t* __restrict__ vt, const int* __restrict__ cu, int LpMax,
                 uint16_t* __restrict__ ctx) {
    // K (this head's [Lp][64] slice) and V^T ([64][Lp]) are staged in LDS ONCE per (passage, head) with coalesced
    // loads; the query fragments then run entirely out of LDS + registers (re-reading K/V^T from L2 for each of the
    // L/16 query fragments made the kernel latency-bound: 300 us -> see profiles). K chunks are XOR-swizzled by
    // (key & 7) for the ds_read_b128 fragment reads; V^T rows are padded by 16 B so dims spread over the banks.
    // L = this passage's packed length (all its keys are real), Lp = L rounded up to 32; keys in [L, Lp) are zero
    // rows / zero V^T columns with an additive mask of -inf.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x / NHEAD, h = blockIdx.x % NHEAD;
    const int tb = cu[b], L = cu[b + 1] - tb;
    if (L <= 0) return;
    const int Lp = (L + 31) & ~31;
    uint4* sK = (uint4*)smem;                                  // [Lp][8]
    const int vstride = Lp + 8;                                // halfs
    uint16_t* sVt = (uint16_t*)(sK + (size_t)Lp * 8);          // [64][Lp + 8]
    float* sMask = (float*)(sVt + 64 * vstride);               // [Lp] 0 for keys < L, -inf beyond
    const uint16_t* Qb = qk + (size_t)tb * (2 * HID) + h * DHEAD;
    const uint16_t* Kb = Qb + HID;
    const uint16_t* Vt = vt + ((size_t)b * HID + h * DHEAD) * LpMax;
    for (int j = tid; j < Lp; j += 256) sMask[j] = (j < L) ? 0.0f : -__builtin_inff();
    for (int idx = tid; idx < Lp * 8; idx += 256) {
        const int key = idx >> 3, ch = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < L) v = *(const uint4*)(Kb + (size_t)key * (2 * HID) + ch * 8);
        sK[key * 8 + (ch ^ (key & 7))] = v;
    }
    const int cpr = Lp / 8;                                    // 16-B chunks per V^T row
    for (int idx = tid; idx < 64 * cpr; idx += 256) {
        const int dim = idx / cpr, c = idx - dim * cpr;
        uint4 v = *(const uint4*)(Vt + (size_t)dim * LpMax + c * 8);
        const int left = L - c * 8;                            // columns >= L were never written: force them to 0
        if (left < 8) {
            uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e >= left) wv[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
            v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
        *(uint4*)(sVt + dim * vstride + c * 8) = v;
    }
    __syncthreads();
    const int nkf = Lp / 16;
    for (int qf = wave; qf * 16 < L; qf += 4) {
        int qrow = qf * 16 + lr; if (qrow >= L) qrow = L - 1;
        const h8 q0 = __builtin_bit_cast(h8, *(const uint4*)(Qb + (size_t)qrow * (2 * HID) + lg * 8));
        const h8 q1 = __builtin_bit_cast(h8, *(const uint4*)(Qb + (size_t)qrow * (2 * HID) + 32 + lg * 8));
        f4 s[MAXKF];
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf) {
            s[kf] = (f4){0.f, 0.f, 0.f, 0.f};
            if (kf < nkf) {
                const int krow = kf * 16 + lr;
                const h8 k0 = __builtin_bit_cast(h8, sK[krow * 8 + (lg ^ (krow & 7))]);
                const h8 k1 = __builtin_bit_cast(h8, sK[krow * 8 + ((4 + lg) ^ (krow & 7))]);
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q0, s[kf], 0, 0, 0);
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q1, s[kf], 0, 0, 0);
            }
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf)
            if (kf < nkf) {
                const float4 am = *(const float4*)(sMask + kf * 16 + lg * 4);
                const float a4[4] = {am.x, am.y, am.z, am.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // fp16(q.k) / sqrt(64) is exact in fp16; + mask is an fp16 add (modeling_bert.py:346-349)
                    const float v = rh(rh(s[kf][r]) * 0.125f + a4[r]);
                    s[kf][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf)
            if (kf < nkf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kf][r] - mx);            // exp(-inf) = 0 for padded keys
                    s[kf][r] = e;
                    sum += e;
                }
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ctx = P V, P straight from the registers above
        f4 o[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXKF / 2; ++ks)
            if (2 * ks < nkf) {
                h8 pa;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = (_Float16)(s[2 * ks][e] * inv);             // softmax(...).type_as(fp16)
                    pa[4 + e] = (_Float16)(s[2 * ks + 1][e] * inv);
                }
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const uint16_t* vrow = sVt + (df * 16 + lr) * vstride + ks * 32 + lg * 4;
                    const uint2 v0 = *(const uint2*)vrow, v1 = *(const uint2*)(vrow + 16);
                    const h8 vb = __builtin_bit_cast(h8, make_uint4(v0.x, v0.y, v1.x, v1.y));
                    o[df] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, vb, o[df], 0, 0, 0);
                }
            }
        // context_layer.permute(0,2,1,3).view(.., 768): [token][h*64 + dim]
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qf * 16 + lg * 4 + r;
                if (row < L) ctx[((size_t)tb + row) * HID + h * DHEAD + df * 16 + lr] = f2h(o[df][r]);
            }
    }
}

// masked mean pooling (retrievers.py:50-52) over the packed tokens of one passage, with the reference's two roundings:
// fp16(sum) (the fp16 tensor .sum(dim=1) returns), then fp16(that / count); the row goes straight to out (= a slab row,
// atlas.py:79). The sum itself is exact (fp16 addends in double), which no summation order of the reference beats.
__global__ void __launch_bounds__(192)
pool_packed_kernel(const uint16_t* __restrict__ x, const int* __restrict__ cu, uint16_t* __restrict__ out) {
    const int b = blockIdx.x;
    const int tb = cu[b], L = cu[b + 1] - tb;
    const uint16_t* base = x + (size_t)tb * HID + threadIdx.x * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int l = 0; l < L; ++l) {
        const uint2 v = *(const uint2*)(base + (size_t)l * HID);
        s[0] += f16_bits_to_f64((uint16_t)(v.x & 0xffff)); s[1] += f16_bits_to_f64((uint16_t)(v.x >> 16));
        s[2] += f16_bits_to_f64((uint16_t)(v.y & 0xffff)); s[3] += f16_bits_to_f64((uint16_t)(v.y >> 16));
    }
    const float cnt = (float)L;                                   // attention_mask.sum(dim=1): 0 -> 0/0 = NaN as in torch
    uint16_t o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f32_to_f16_bits(f16_bits_to_f32(f64_to_f16_bits(s[r])) / cnt);
    *(uint2*)(out + (size_t)b * HID + threadIdx.x * 4) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
}

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
}

extern "C" {

size_t atlas_contriever_workspace_bytes(int n, int L) {
    if (n <= 0 || L <= 0) return 0;
    const size_t M = (size_t)n * L, Lp = (size_t)((L + 31) & ~31);
    // x, ctx, u : [M,768]; qk : [M,1536]; vt : [n,768,Lp]; h : [M,3072]; counts[n], cu[n+1], tokinfo[M]
    return up256(M * HID * 2) * 3 + up256(M * 2 * HID * 2) + up256((size_t)n * HID * Lp * 2) + up256(M * 4 * HID * 2) +
           up256((size_t)n * 4) + up256((size_t)(n + 1) * 4) + up256(M * 8) + 256;
}

int atlas_contriever_embed(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask,
                           const int64_t* token_type_ids, int n, int L, void* out_f16, void* ws, size_t ws_bytes,
                           void* stream_) {
    if (!w || !input_ids || !attention_mask || !out_f16 || !ws) return ATLAS_E_BADARG;
    if (n <= 0 || L <= 0) return ATLAS_E_BADARG;
    if (L > 512 || w->hidden != HID || w->n_heads != NHEAD || w->intermediate != 4 * HID || w->n_layers < 1 ||
        w->n_layers > ATLAS_BERT_MAX_LAYERS || (int64_t)n * L > 0x7fffffff)
        return ATLAS_E_UNSUPPORTED;
    if (ws_bytes < atlas_contriever_workspace_bytes(n, L)) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t M = (int64_t)n * L;                 // worst case; the packed count lives on the device (cu[n])
    const int Lp = (L + 31) & ~31;
    unsigned char* p = (unsigned char*)ws;
    uint16_t* x = (uint16_t*)p;   p += up256((size_t)M * HID * 2);
    uint16_t* ctx = (uint16_t*)p; p += up256((size_t)M * HID * 2);
    uint16_t* u = (uint16_t*)p;   p += up256((size_t)M * HID * 2);
    uint16_t* qk = (uint16_t*)p;  p += up256((size_t)M * 2 * HID * 2);
    uint16_t* vt = (uint16_t*)p;  p += up256((size_t)n * HID * Lp * 2);
    uint16_t* hbuf = (uint16_t*)p; p += up256((size_t)M * 4 * HID * 2);
    int* counts = (int*)p;        p += up256((size_t)n * 4);
    int* cu = (int*)p;            p += up256((size_t)(n + 1) * 4);
    int2* tokinfo = (int2*)p;

    const unsigned tok_blocks = (unsigned)((M + 3) / 4), pas_blocks = (unsigned)((n + 3) / 4);
    const char* cfg_env = getenv("ATLAS_GEMM_CFG");
    const int cfg = cfg_env ? atoi(cfg_env) : 2;        // 256x256 tiles measured best (profiles/r01/e01)
    const size_t att_lds = (size_t)Lp * 128 + (size_t)64 * (Lp + 8) * 2 + (size_t)Lp * 4;
    (void)hipFuncSetAttribute((const void*)attention_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attention_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attention_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(count_kernel, dim3(pas_blocks), dim3(256), 0, stream, attention_mask, n, L, counts);
    hipLaunchKernelGGL(pack_kernel, dim3(pas_blocks), dim3(256), 0, stream, attention_mask, n, L, counts, cu, tokinfo);
    hipLaunchKernelGGL(embed_ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, input_ids, token_type_ids, L, cu, n, tokinfo,
                       (const uint16_t*)w->word_emb, (const uint16_t*)w->pos_emb, (const uint16_t*)w->type_emb,
                       (const uint16_t*)w->emb_ln_w, (const uint16_t*)w->emb_ln_b, w->eps, x);
    for (int l = 0; l < w->n_layers; ++l) {
        const atlas_bert_layer& ly = w->layers[l];
        launch_gemm<3>(cfg, stream, x, (const uint16_t*)ly.qkv_w, (const uint16_t*)ly.qkv_b, (const uint16_t*)nullptr, qk, vt, M,
                       cu, n, tokinfo, 3 * HID, HID, Lp);
        if (Lp <= 128)
            hipLaunchKernelGGL(attention_kernel<8>, dim3((unsigned)n * NHEAD), dim3(256), att_lds, stream, qk, vt, cu, Lp, ctx);
        else if (Lp <= 256)
            hipLaunchKernelGGL(attention_kernel<16>, dim3((unsigned)n * NHEAD), dim3(256), att_lds, stream, qk, vt, cu, Lp, ctx);
        else
            hipLaunchKernelGGL(attention_kernel<32>, dim3((unsigned)n * NHEAD), dim3(256), att_lds, stream, qk, vt, cu, Lp, ctx);
        launch_gemm<2>(cfg, stream, ctx, (const uint16_t*)ly.o_w, (const uint16_t*)ly.o_b, x, u, (uint16_t*)nullptr, M, cu, n, tokinfo,
                       HID, HID, Lp);
        hipLaunchKernelGGL(ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, u, cu + n, (const uint16_t*)ly.ln1_w,
                           (const uint16_t*)ly.ln1_b, w->eps, x);
        launch_gemm<1>(cfg, stream, x, (const uint16_t*)ly.ff1_w, (const uint16_t*)ly.ff1_b, (const uint16_t*)nullptr, hbuf,
                       (uint16_t*)nullptr, M, cu, n, tokinfo, 4 * HID, HID, Lp);
        launch_gemm<2>(cfg, stream, hbuf, (const uint16_t*)ly.ff2_w, (const uint16_t*)ly.ff2_b, x, u, (uint16_t*)nullptr, M, cu, n,
                       tokinfo, HID, 4 * HID, Lp);
        hipLaunchKernelGGL(ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, u, cu + n, (const uint16_t*)ly.ln2_w,
                           (const uint16_t*)ly.ln2_b, w->eps, x);
    }
    // rows written contiguously at out_f16 (which may point into the passage slab: slab + row_offset * 768)
    hipLaunchKernelGGL(pool_packed_kernel, dim3((unsigned)n), dim3(192), 0, stream, x, cu, (uint16_t*)out_f16);
    return (int)hipGetLastError();
}

}  // extern "C"
