// encoder.hip — Contriever (BERT-base) encoder for the index-refresh path and for query embedding, gfx950.
//
// Replaces src/retrievers.py:22-60 on top of src/modeling_bert.py (BertEmbeddings :213-247, BertSelfAttention :290-366,
// BertSelfOutput :382-387, BertIntermediate :448-451, BertOutput :461-466, BertLayerNorm :104-114) for inference:
//   fp16  the copy `copy.deepcopy(retriever).half().eval()` of Atlas.build_index / retrieve_with_rerank (src/atlas.py:54-59, 78, 168)
//   fp32 / bf16 / fp16  query embedding in --precision (src/atlas.py:104)
// Numerics follow the model dtype op by op: every tensor the reference materialises in that dtype is rounded to it here
// at the same place (GEMM outputs after the fp32 bias add, the residual sums, scores, softmax probabilities, GELU
// outputs, both steps of `weight * y + bias` in the NON-standard LayerNorm, the two roundings of the pooling); softmax
// and LayerNorm statistics are fp32; the erf GELU is evaluated in fp32 (16-bit dtypes: common.h::gelu_erf_poly, fp32:
// erff). What may differ from a given torch backend is only the fp32 summation order inside GEMMs and reductions.
//
// Token packing: only tokens with attention_mask != 0 are computed. Masked keys get probability exactly 0 in the
// reference (exp(-10000 + s - max) underflows to 0 in fp32) and masked tokens' hidden states are dropped by the
// pooling (retrievers.py:50), so leaving them out changes no result; a batch of queries padded to 512 with ~20 real
// tokens costs 20 tokens. The packing is done on the device (count_kernel + pack_kernel, no host sync): kernels are
// launched for the worst case n*L tokens and blocks beyond the packed count T = cu[n] exit immediately.
//
// Kernels (T = F16 | BF16 | F32 traits)
//   count_kernel / pack_kernel   per-passage real-token counts, exclusive scan cu[n+1], tokinfo[t] = (passage, position | rank << 16)
//   embed_ln_kernel<T>   word + type (+= position) embeddings, LayerNorm                     (one wave per token)
//   gemm_pp_kernel<T,EPI>   C[M,N] = A[M,K] . W[N,K]^T + bias on the matrix cores, 256 x 256 tiles, LDS-DMA staging, the
//                        two waves of a SIMD in opposite read / multiply phases; epilogues: QKV split + V^T | erf GELU |
//                        + residual, written out through LDS in whole rows        (MFMA-bound: the refresh roofline)
//   gemm_co_kernel<T,EPI>   the same GEMM as two co-resident 4-wave workgroups per CU on 256 x 128 tiles (serves FFN-1)
//   gemm_bt_kernel<T,EPI,..>  the single-phase version; 128 x 128 and 64 x 64 tiles serve small (query) batches
//   attention_kernel<T,MAXKF>  per (passage, head): S^T = K.Q^T -> dtype -> /8 -> fp32 softmax -> dtype P -> P.V, P in registers
// V^T layout: [passage][768][Lp + 8], the key of packed token t of passage b at column (t - cu[b]) + (cu[b] & 7) = t - (cu[b] & ~7): the
// passage's keys start cu[b] & 7 columns in, so that 8 consecutive PACKED tokens starting at a multiple of 8 (what a GEMM lane holds)
// are always one 16-byte-aligned run of a row, for ragged batches too; the attention kernel reads its rows from that offset on
//   attention_f32_kernel the same on v_mfma_f32_16x16x4_f32 for the fp32 model
//   ln_kernel<T>         the reference's LayerNorm on a [T,768] tensor                     (one wave per token)
//   pool_packed_kernel<T>  mean over a passage's tokens with the reference's two roundings, row written into the slab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

#ifndef ATLAS_TUNING
#define ATLAS_TUNING 0
#endif

#include "common.h"
#include "../../include/atlas_hip.h"

#ifndef ATLAS_PT_WDEFER
#define ATLAS_PT_WDEFER 0            // gemm_pt_kernel experiment (measured slower, see its `stage`): W pieces per wave and k-tile issued between the wave's own MFMAs
#endif
#ifndef ATLAS_PT_RSPLIT
#define ATLAS_PT_RSPLIT 1            // gemm_pt_kernel: a k-tile's k-step-0 fragments are read a phase AHEAD, between the wave's own MFMAs of the previous k-tile's k-step 1 (0 = rounds 3-4; 2 = a variant that spills)
#endif
#ifndef ATLAS_PT_WSTRIDE
#define ATLAS_PT_WSTRIDE 1           // ... one in front of every ATLAS_PT_WSTRIDE-th chunk of eight MFMAs
#endif

using namespace atlas;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));   // 16-byte MFMA operand chunk (see the note on LDS reads in gemm_bt_kernel)
typedef uint4 __attribute__((aligned(2))) uint4_a2;          // 16-byte global loads at the alignment of their elements (V^T rows of ragged batches)
typedef uint4 __attribute__((aligned(4))) uint4_a4;

#define HID 768
#define NHEAD 12
#define DHEAD 64

// ---- the three model precisions (opt.precision fp16 | bf16 | fp32; atlas.py / model_io.py cast the retriever) ----
// ld/st: element <-> fp32; rnd: "this tensor is materialised in the model dtype here" (identity for fp32);
// mma: one 16-byte operand chunk per lane pair -> v_mfma 16x16 (k = 32 halves / 32 bf16 / 4 x (k = 4) floats).
struct F16 {
    typedef uint16_t elem;
    static constexpr int DT = ATLAS_DT_F16;
    static __device__ __forceinline__ float ld(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ uint16_t st(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // v_cvt_f16_f32, RNE
    static __device__ __forceinline__ float rnd(float f) { return (float)(_Float16)f; }
    static __device__ __forceinline__ f4 mma(const u4v& a, const u4v& b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4 mma(const uint4& a, const uint4& b, f4 c) { return mma(__builtin_bit_cast(u4v, a), __builtin_bit_cast(u4v, b), c); }
};
struct BF16 {
    typedef uint16_t elem;
    static constexpr int DT = ATLAS_DT_BF16;
    static __device__ __forceinline__ float ld(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
    static __device__ __forceinline__ uint16_t st(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }     // v_cvt_pk_bf16_f32, RNE
    static __device__ __forceinline__ float rnd(float f) { return ld(st(f)); }
    static __device__ __forceinline__ f4 mma(const u4v& a, const u4v& b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4 mma(const uint4& a, const uint4& b, f4 c) { return mma(__builtin_bit_cast(u4v, a), __builtin_bit_cast(u4v, b), c); }
};
struct F32 {
    typedef float elem;
    static constexpr int DT = ATLAS_DT_F32;
    static __device__ __forceinline__ float ld(float b) { return b; }
    static __device__ __forceinline__ float st(float f) { return f; }
    static __device__ __forceinline__ float rnd(float f) { return f; }
    static __device__ __forceinline__ f4 mma(const uint4& a, const uint4& b, f4 c) { return mma(__builtin_bit_cast(u4v, a), __builtin_bit_cast(u4v, b), c); }
    static __device__ __forceinline__ f4 mma(const u4v& a, const u4v& b, f4 c) {
        // the contraction index of a chunk is enumerated (lane group, element): both operands use the same order.
        // (elements are copied to scalars first: __builtin_bit_cast applied directly to an ext-vector element
        // reads element 0 with this hipcc)
        const uint32_t a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, b0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a1), __builtin_bit_cast(float, b1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a2), __builtin_bit_cast(float, b2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a3), __builtin_bit_cast(float, b3), c, 0, 0, 0);
        return c;
    }
};
// 4 consecutive elements <-> 4 floats
template <class T> static __device__ __forceinline__ void load4(const typename T::elem* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<F32>(const float* p, float (&v)[4]) {
    const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <class T> static __device__ __forceinline__ void load4(const typename T::elem* p, float (&v)[4]) {
    const uint2 t = *(const uint2*)p;
    v[0] = T::ld((uint16_t)(t.x & 0xffff)); v[1] = T::ld((uint16_t)(t.x >> 16));
    v[2] = T::ld((uint16_t)(t.y & 0xffff)); v[3] = T::ld((uint16_t)(t.y >> 16));
}
template <class T> static __device__ __forceinline__ void store4(typename T::elem* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<F32>(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <class T> static __device__ __forceinline__ void store4(typename T::elem* p, const float (&v)[4]) {
    *(uint2*)p = make_uint2((uint32_t)T::st(v[0]) | ((uint32_t)T::st(v[1]) << 16), (uint32_t)T::st(v[2]) | ((uint32_t)T::st(v[3]) << 16));
}

// acc[a][b] += W-fragment a x activation-fragment b for one 16-byte k-chunk per lane. For fp32 a chunk is four k = 4 MFMAs:
// they are issued element-major, so that consecutive MFMAs go to DIFFERENT accumulators (a dependent v_mfma_f32_16x16x4_f32
// waits 40 cycles, an independent one issues after 32)
template <class T, int FA, int FB>
static __device__ __forceinline__ void mma_tile(const u4v (&fw)[FA], const u4v (&fa)[FB], f4 (&acc)[FA][FB]) {
    if constexpr (T::DT == ATLAS_DT_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < FA; ++a) {
                const uint32_t wa = fw[a][e];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const uint32_t xb = fa[b][e];
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wa), __builtin_bit_cast(float, xb), acc[a][b], 0, 0, 0);
                }
            }
    } else {
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
            for (int b = 0; b < FB; ++b) acc[a][b] = T::mma(fw[a], fa[b], acc[a][b]);
    }
}

static __device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// ---- the reference's LayerNorm on one 768-vector held 12 per lane ----
// The lane's values are elements 8 l .. 8 l + 7 and 512 + 4 l .. 512 + 4 l + 3 of the row: one 16-byte and one 8-byte access per lane, both
// contiguous across the wave (round 2 held element 64 i + l: twelve 2-byte accesses). This mapping and the order the statistics are summed
// in (the lane's twelve values in that order, then the xor butterfly) are shared by every LayerNorm here, so all of them give the same bits.
template <class T> static __device__ __forceinline__ void ln_load12(const typename T::elem* __restrict__ row, const int lane, float (&x)[12]) {
    if constexpr (sizeof(typename T::elem) == 2) {
        const uint4 a = *(const uint4*)(row + 8 * lane);
        const uint2 b = *(const uint2*)(row + 512 + 4 * lane);
        const uint32_t w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int i = 0; i < 6; ++i) { x[2 * i] = T::ld((uint16_t)(w[i] & 0xffff)); x[2 * i + 1] = T::ld((uint16_t)(w[i] >> 16)); }
    } else {
        const float4 a = *(const float4*)(row + 8 * lane), b = *(const float4*)(row + 8 * lane + 4), c = *(const float4*)(row + 512 + 4 * lane);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w; x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
    }
}
template <class T> static __device__ __forceinline__ void ln_store12(typename T::elem* __restrict__ row, const int lane, const float (&y)[12]) {
    if constexpr (sizeof(typename T::elem) == 2) {
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = (uint32_t)T::st(y[2 * i]) | ((uint32_t)T::st(y[2 * i + 1]) << 16);
        *(uint4*)(row + 8 * lane) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint2*)(row + 512 + 4 * lane) = make_uint2(w[4], w[5]);
    } else {
        *(float4*)(row + 8 * lane) = make_float4(y[0], y[1], y[2], y[3]);
        *(float4*)(row + 8 * lane + 4) = make_float4(y[4], y[5], y[6], y[7]);
        *(float4*)(row + 512 + 4 * lane) = make_float4(y[8], y[9], y[10], y[11]);
    }
}
// modeling_bert.py:104-114: mean and UNCENTRED second moment in fp32, y = dtype((x-mean)*rsqrt(E[x^2]+eps)),
// out = dtype(dtype(w*y) + b): two separately rounded ops (no fma), in fp32 as well.  x -> o (the caller stores o with ln_store12)
template <class T>
static __device__ __forceinline__ void layer_norm_768(const float (&x)[12], const float (&w)[12], const float (&b)[12], const float eps, float (&o)[12]) {
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { s += x[i]; s2 += x[i] * x[i]; }
    s = wave_sum(s); s2 = wave_sum(s2);
    const float mean = s * (1.0f / HID), var = s2 * (1.0f / HID);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float y = T::rnd(__fmul_rn(x[i] - mean, rstd));
        o[i] = T::rnd(__fadd_rn(T::rnd(__fmul_rn(w[i], y)), b[i]));
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
// tokinfo[cu[b] + rank] = (b, l | rank << 16) for the real tokens of passage b in position order (l, rank < 512: the rank inside the
// passage = the token's V^T column rides along, so that an epilogue that needs it does not have to chase cu[passage])
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
        const int at = run + __popcll(bal & ((1ull << lane) - 1ull));
        if (real) tokinfo[at] = make_int2(b, l | ((at - base) << 16));
        run += __popcll(bal);
    }
}

// one wave per packed token: embeddings (modeling_bert.py:213-247)
template <class T>
__global__ void __launch_bounds__(256)
embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids, int L, const int* __restrict__ cu, int n,
                const int2* __restrict__ tokinfo, int vocab, int type_vocab,
                const typename T::elem* __restrict__ word, const typename T::elem* __restrict__ pos, const typename T::elem* __restrict__ type,
                const typename T::elem* __restrict__ lnw, const typename T::elem* __restrict__ lnb, float eps, typename T::elem* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= cu[n]) return;
    const int2 ti = tokinfo[t];
    const size_t src = (size_t)ti.x * L + (ti.y & 0xffff);
    int64_t id = ids[src], ty = type_ids ? type_ids[src] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);            // never read outside the tables (atlas_hip.h)
    ty = ty < 0 ? 0 : (ty >= type_vocab ? type_vocab - 1 : ty);
    const int p = ti.y & 0xffff;
    float x[12], e1[12], e2[12], e3[12], w[12], b[12];
    ln_load12<T>(word + id * HID, lane, e1);
    ln_load12<T>(type + ty * HID, lane, e2);
    ln_load12<T>(pos + (size_t)p * HID, lane, e3);
    ln_load12<T>(lnw, lane, w);
    ln_load12<T>(lnb, lane, b);
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = T::rnd(T::rnd(e1[i] + e2[i]) + e3[i]);    // inputs_embeds + token_type_embeddings; embeddings += position_embeddings
    float o[12];
    layer_norm_768<T>(x, w, b, eps, o);
    ln_store12<T>(out + (size_t)t * HID, lane, o);
}

// one wave per token: LayerNorm(x.float()).type_as(x) on a [T,768] tensor
template <class T>
__global__ void __launch_bounds__(256)
ln_kernel(const typename T::elem* __restrict__ in, const int* __restrict__ Tdev, const typename T::elem* __restrict__ lnw,
          const typename T::elem* __restrict__ lnb, float eps, typename T::elem* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= *Tdev) return;
    float x[12], w[12], b[12], o[12];
    ln_load12<T>(in + (size_t)t * HID, lane, x);
    ln_load12<T>(lnw, lane, w);
    ln_load12<T>(lnb, lane, b);
    layer_norm_768<T>(x, w, b, eps, o);
    ln_store12<T>(out + (size_t)t * HID, lane, o);
}

// epilogue shared by the GEMM kernels: acc[a][b][r] = C[token m0 + wj*FB*16 + 16b + lr][col n0 + wi*FA*16 + 16a + 4lg + r]
template <class T, int EPI, int FA, int FB>
static __device__ __forceinline__ void gemm_epilogue(f4 (&acc)[FA][FB], int64_t m0, int n0, int wi, int wj, int lr, int lg, int64_t M, int N,
                                                     const typename T::elem* __restrict__ bias, const typename T::elem* __restrict__ R,
                                                     typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
                                                     const int* __restrict__ cu, const int2* __restrict__ tokinfo, int Lp) {
    const bool v_part = (EPI == 3) && (n0 >= 2 * HID);      // QKV projection: the V columns are stored transposed
#pragma unroll
    for (int b = 0; b < FB; ++b) {
        const int64_t tok = m0 + wj * (FB * 16) + b * 16 + lr;
        if (tok >= M) continue;
        int64_t pb = 0;
        int pos = 0;                                             // rank of the token inside its passage = V^T column
        if (EPI == 3 && v_part) { pb = tokinfo[tok].x; pos = (int)(tok - (cu[pb] & ~7)); }      // V^T column: rank + (cu[pb] & 7)
#pragma unroll
        for (int a = 0; a < FA; ++a) {
            const int col = n0 + wi * (FA * 16) + a * 16 + lg * 4;
            float bv[4], rr[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
            load4<T>(bias + col, bv);
            if (EPI == 2) load4<T>(R + (size_t)tok * N + col, rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = T::rnd(acc[a][b][r] + bv[r]);                                 // Linear output in the model dtype
                if (EPI == 1) v = (sizeof(typename T::elem) == 2) ? gelu_erf_poly(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // erf GELU in fp32 (common.h)
                if (EPI == 2) v = v + rr[r];                                            // + input_tensor
                o[r] = v;                                                               // (rounded by the store)
            }
            if (EPI == 3) {
                if (v_part) {
                    // V^T[passage][h*64+d][key]: the PV product wants consecutive KEYS per lane (attention kernels)
#pragma unroll
                    for (int r = 0; r < 4; ++r) VT[((size_t)pb * HID + (col - 2 * HID + r)) * Lp + pos] = T::st(o[r]);
                } else {
                    store4<T>(C + (size_t)tok * (2 * HID) + col, o);
                }
            } else {
                store4<T>(C + (size_t)tok * N + col, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// GEMM: C[M,N] = A[M,K] (row-major) . W[N,K]^T (row-major) + bias[N], fp32 accumulate, all tensors in the model dtype.
// Computed transposed on the matrix cores (MFMA A operand = W rows, B operand = A rows) so that each lane ends up
// with 4 consecutive OUTPUT COLUMNS of one token: the epilogue reads/writes 4 contiguous elements per fragment.
// Tile = BCOL output columns x BTOK tokens x 128 BYTES of k (64 halves / 32 floats); WC x WT waves, each
// (BCOL/WC) x (BTOK/WT) = FA x FB fragments of v_mfma 16x16. LDS: two stages of (BCOL + BTOK) x 128 B; the
// 16-B chunks of a row are XOR-swizzled by (row & 7) so a fragment read of 16 rows at one k-chunk spreads over 8
// bank groups. Global -> LDS by LDS-DMA (no registers); the next stage streams in under the current stage's MFMAs.
//   EPI 1: C = dt(gelu_erf(dt(acc + bias)))    (BertIntermediate)
//   EPI 2: C = dt(dt(acc + bias) + R)          (dense + residual of BertSelfOutput / BertOutput; LayerNorm follows)
//   EPI 3: QKV projection: q | k columns to C [M,1536], v columns transposed to VT [n][768][Lp]
// Requires N % BCOL == 0, K * sizeof(elem) % 128 == 0 (768, 2304, 3072 all are); M = cu[n] read from the device.
// ------------------------------------------------------------------------------------------
template <class T, int EPI, int BCOL, int BTOK, int WC, int WT>
__global__ void __launch_bounds__(WC * WT * 64)
gemm_bt_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp) {
    typedef typename T::elem E;
    constexpr int NWV = WC * WT, FA = BCOL / WC / 16, FB = BTOK / WT / 16;
    constexpr int W_U4 = BCOL * 8, A_U4 = BTOK * 8;                  // uint4 per stage
    constexpr int EPC = 16 / (int)sizeof(E);                           // elements per 16-B chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* sW = (uint4*)smem_raw;                                      // [2][W_U4]
    uint4* sA = sW + 2 * W_U4;                                         // [2][A_U4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WT, wj = wave % WT;
    // XCD-aware tile order (cdna guide T1): hardware block b runs on XCD b % 8, each XCD has its own L2. All column
    // tiles of one token tile are given to ONE XCD (token tile t -> XCD t % 8), so the big activation tile
    // (BTOK x K) is fetched into one L2 once instead of into up to 8; the weights (<= 9.4 MB) stay L2/MALL-resident.
    const int64_t M = cu[n];                                           // packed token count (the grid covers n*L)
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
    auto stage = [&](const int buf, const int kt) {
        const int ch = (lane & 7) ^ (lane >> 3);
        const int k0 = kt * (8 * EPC) + ch * EPC;                      // element offset inside a row
#pragma unroll
        for (int i = 0; i < BCOL / 8 / NWV; ++i) {
            const int rowbase = (wave * (BCOL / 8 / NWV) + i) * 8;
            const E* gw = W + (size_t)(n0 + rowbase + (lane >> 3)) * K + k0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
                                             (__attribute__((address_space(3))) void*)&sW[buf * W_U4 + rowbase * 8], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BTOK / 8 / NWV; ++i) {
            const int rowbase = (wave * (BTOK / 8 / NWV) + i) * 8;
            int64_t ar = m0 + rowbase + (lane >> 3);
            if (ar >= M) ar = M - 1;                                   // clamped: tail rows are never stored
            const E* ga = A + (size_t)ar * K + k0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)&sA[buf * A_U4 + rowbase * 8], 16, 0, 0);
        }
    };

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // Per k-tile: (1) all fragments of the tile (both 64-B halves) are read into registers, (2) the LDS-DMA for the
    // NEXT tile is issued into the other buffer, (3) the MFMAs run with that DMA in flight, (4) barrier.
    // Order (1) -> (2) matters to hipcc: its waitcnt insertion drains vmcnt(0) in front of any LDS read that follows
    // an in-flight LDS-DMA write it cannot prove disjoint, so a prefetch issued BEFORE the reads is drained at once
    // and nothing overlaps (measured: -14 %). Issued after them, the only drain is the one __syncthreads needs anyway.
    stage(0, 0);
    __syncthreads();                                   // (vmcnt(0) + barrier: every wave's DMA has landed)
    const int nk = K / (8 * EPC);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        uint4 fw[2][FA], fa[2][FB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < FA; ++a) {
                const int row = wi * (FA * 16) + a * 16 + lr;
                fw[ks][a] = sW[buf * W_U4 + row * 8 + ((ks * 4 + lg) ^ (row & 7))];
            }
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int row = wj * (FB * 16) + b * 16 + lr;
                fa[ks][b] = sA[buf * A_U4 + row * 8 + ((ks * 4 + lg) ^ (row & 7))];
            }
        }
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);       // streams in while this tile is multiplied
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u4v vw[FA], va[FB];
#pragma unroll
            for (int a = 0; a < FA; ++a) vw[a] = __builtin_bit_cast(u4v, fw[ks][a]);
#pragma unroll
            for (int b = 0; b < FB; ++b) va[b] = __builtin_bit_cast(u4v, fa[ks][b]);
            mma_tile<T, FA, FB>(vw, va, acc);
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the drain + barrier BEHIND the MFMAs (hipcc hoists it otherwise)
        __syncthreads();
    }

    gemm_epilogue<T, EPI, FA, FB>(acc, m0, n0, wi, wj, lr, lg, M, N, bias, R, C, VT, cu, tokinfo, Lp);
}

// Epilogue of the 256 x 256 kernel for the 16-bit dtypes: the C tile goes through LDS (free after the k-loop; 128 KiB)
// so that global memory sees whole 512-byte rows (16 B per lane) instead of the fragment layout's 32-byte pieces, and
// V^T sees 64 consecutive tokens per store. Rounding points are those of gemm_epilogue (dt(acc + bias) [gelu] is what
// is parked in LDS; the residual is added on the way out). LDS tile [token][col], 8-byte granules XOR-ed with
// (token & 15) << 2: conflict-free fragment writes and row reads. V tiles are parked TRANSPOSED ([col][token], transposed in
// registers by two DPP exchanges) so that V^T rows leave as 16-byte stores of 8 consecutive keys.
// BTOK = token rows of the tile: 256 (8 waves, gemm_pp_kernel) or 128 (4 waves, gemm_co_kernel); 256 columns either way
template <class T, int EPI, int BTOK = 256>
static __device__ __forceinline__ void gemm_epilogue_lds(f4 (&acc)[8][4], unsigned char* __restrict__ smem,
                                                         const uint16_t* __restrict__ s_bias /* LDS: bias[n0 .. n0+256) */, int64_t m0, int n0, int wave,
                                                         int lane, int64_t M, int N,
                                                         const uint16_t* __restrict__ R, uint16_t* __restrict__ C,
                                                         uint16_t* __restrict__ VT, const int* __restrict__ cu,
                                                         const int2* __restrict__ tokinfo, int Lp) {
    constexpr int WT = BTOK / 64;                               // waves along the tokens (wave tile = 128 columns x 64 tokens)
    constexpr int OCT = BTOK / 8;                               // token octets per tile row of the transposed (V) layout
    constexpr int CPI = 64 / OCT;                               // V columns per wave instruction
    constexpr int CW = 256 / (BTOK / 32);                       // V columns per wave
    const int wi = wave / WT, wj = wave % WT, lr = lane & 15, lg = lane >> 4;
    const bool v_tile = (EPI == 3) && (n0 >= 2 * HID);        // workgroup-uniform: the V columns of the QKV projection
    // EPI 2: the 16 residual row pieces this lane adds on the way out are requested FIRST, so that their latency runs under the
    // conversion, the LDS writes and the barrier (in the store loop, four at a time, it was paid four times: ~8 us of a 17 us epilogue)
    uint4 rv[16];
    if (EPI == 2) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            int64_t tok = m0 + wave * 32 + it * 2 + (lane >> 5);
            if (tok >= M) tok = M - 1;                                                   // clamped: tail rows are never stored
            rv[it] = *(const uint4*)(R + (size_t)tok * N + n0 + 8 * (lane & 31));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // V tile: where this lane's 8 consecutive tokens (8j .. 8j+7 of the tile, j = lane & 31) live in V^T; requested before the
    // conversion as well. One 16-byte store per column needs them in ONE passage at a key offset that is a multiple of 8.
    int64_t vt_off = 0;            // element offset of token 8j inside a V^T row block: passage * 768 * pitch + column
    int vt_mode = 3;               // 0: one 16-byte store (the 8 tokens are one passage's: always aligned, see the layout note) | 3: per element
    int64_t vt_off8[8];            // mode 3 only (groups that straddle passages / the end of the batch): per-token offsets, -1 = no token
    if (EPI == 3 && v_tile) {
        const int64_t tok0 = m0 + 8 * (lane % OCT);
        if (tok0 + 7 < M) {
            const int pb0 = tokinfo[tok0].x, pb7 = tokinfo[tok0 + 7].x;
            vt_off = (int64_t)pb0 * HID * Lp + (tok0 - (cu[pb0] & ~7));
            if (pb0 == pb7) vt_mode = 0;
        }
        if (vt_mode == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t tok = tok0 + e;
                vt_off8[e] = -1;
                if (tok < M) { const int64_t pb = tokinfo[tok].x; vt_off8[e] = pb * HID * Lp + (tok - (cu[pb] & ~7)); }
            }
        }
    }
    if (!(EPI == 3 && v_tile)) {
        // LDS tile [token][col]: 512-byte token rows, 8-byte granules XOR-ed with (token & 15) << 2
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int col = wi * 128 + a * 16 + lg * 4;
            float bv[4];
            load4<T>(s_bias + col, bv);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int t = wj * 64 + b * 16 + lr;
                uint16_t o[4];
                if (EPI == 1) {                                                             // erf GELU in fp32, two values per instruction (common.h)
                    const gelu_f2 g01 = gelu_erf_poly2((gelu_f2){T::rnd(acc[a][b][0] + bv[0]), T::rnd(acc[a][b][1] + bv[1])});
                    const gelu_f2 g23 = gelu_erf_poly2((gelu_f2){T::rnd(acc[a][b][2] + bv[2]), T::rnd(acc[a][b][3] + bv[3])});
                    o[0] = T::st(g01.x); o[1] = T::st(g01.y); o[2] = T::st(g23.x); o[3] = T::st(g23.y);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = T::st(acc[a][b][r] + bv[r]);        // Linear output in the model dtype
                }
                const int gr = (col >> 2) ^ ((t & 15) << 2);
                *(uint2*)(smem + t * 512 + gr * 8) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
            }
        }
    } else {
        // V tile: LDS tile [col][token] (transposed). A lane holds 4 columns of ONE token; two DPP exchanges inside each quad of
        // lanes (tokens 4q .. 4q+3 of a fragment) turn that into 4 tokens of ONE column = one 8-byte granule:
        //   after xor 1: lane keeps 2 of its 4 columns (even lane: 0,1; odd lane: 2,3) for the token pair of (lane, lane ^ 1)
        //   after xor 2: lane keeps 1 of those 2 columns for the 4 tokens of its quad
        // column of lane lr inside the group of 4: 2 * (lr & 1) + ((lr >> 1) & 1). Granules are XOR-ed with (col & 3) << 2: the 16
        // lanes of a row group hit 16 different granule slots, the 4 row groups the same ones (4 passes = the minimum for 512 B).
        const bool odd1 = lane & 1, odd2 = lane & 2;
        const int ciq = 2 * (lr & 1) + ((lr >> 1) & 1);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            float bv[4];
            load4<T>(s_bias + wi * 128 + a * 16 + lg * 4, bv);
            const int colT = wi * 128 + a * 16 + lg * 4 + ciq;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = T::st(acc[a][b][r] + bv[r]);             // Linear output in the model dtype
                const uint32_t p01 = (uint32_t)o[0] | ((uint32_t)o[1] << 16), p23 = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                const uint32_t keep1 = odd1 ? p23 : p01;
                const uint32_t recv1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd1 ? p01 : p23), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                const uint32_t ev = odd1 ? recv1 : keep1, od = odd1 ? keep1 : recv1;        // even / odd token of the pair
                const uint32_t cA = (ev & 0xffffu) | (od << 16), cB = (ev >> 16) | (od & 0xffff0000u);
                const uint32_t keep2 = odd2 ? cB : cA;
                const uint32_t recv2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd2 ? cA : cB), 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
                const uint2 g = odd2 ? make_uint2(recv2, keep2) : make_uint2(keep2, recv2);  // tokens 4q, 4q+1 | 4q+2, 4q+3
                const int gi = wj * 16 + b * 4 + (lr >> 2);
                *(uint2*)(smem + colT * (BTOK * 2) + ((gi ^ ((colT & 3) << 2)) * 8)) = g;
            }
        }
    }
    __syncthreads();
    if (!(EPI == 3 && v_tile)) {
        const int j = lane & 31;
        const int ldc = (EPI == 3) ? 2 * HID : N;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int t = wave * 32 + it * 2 + (lane >> 5);
            const int64_t tok = m0 + t;
            if (tok >= M) continue;
            uint4 v = *(const uint4*)(smem + t * 512 + (((2 * j) ^ ((t & 15) << 2)) * 8));
            if (EPI == 2) {
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, rw[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
                uint32_t ow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T::st(T::ld((uint16_t)(vw[e] & 0xffff)) + T::ld((uint16_t)(rw[e] & 0xffff)));   // + input_tensor
                    const uint16_t hi = T::st(T::ld((uint16_t)(vw[e] >> 16)) + T::ld((uint16_t)(rw[e] >> 16)));
                    ow[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                }
                v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
            *(uint4*)(C + (size_t)tok * ldc + n0 + 8 * j) = v;
        }
    } else {
        // V^T[passage][h*64+d][column]: a lane stores 8 consecutive keys of one column (16 B, always aligned: layout note at the top);
        // the token groups that straddle two passages or the end of the batch go out as single elements
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int colT = wave * CW + it * CPI + lane / OCT;
            const int j = lane % OCT;
            const uint4 v = *(const uint4*)(smem + colT * (BTOK * 2) + (((2 * j) ^ ((colT & 3) << 2)) * 8));
            const int64_t crow = (int64_t)(n0 - 2 * HID + colT) * Lp;
            uint16_t* dst = VT + vt_off + crow;
            if (vt_mode == 0) {
                *(uint4*)dst = v;
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (vt_off8[e] >= 0) VT[vt_off8[e] + crow] = (uint16_t)(w4[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same GEMM for the bulk refresh (256 x 256 x 128-byte tiles, 8 waves as 2 x 4, wave tile 128 x 64), with the two
// waves of every SIMD in opposite phases ("ping-pong"): while one does its LDS fragment reads and issues the LDS-DMA
// for a later k-tile, the other runs its 64 MFMAs, so the matrix pipe of the SIMD is fed from one of them all the
// time. In gemm_bt_kernel all 8 waves read, then all multiply (one barrier per k-tile): PMC showed the MFMA pipe busy
// 39 % there (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x GRBM_GUI_ACTIVE), profiles/r01/encoder_pmc.txt).
//
// Phases are separated by s_barrier (all 8 waves). Group A = waves 0-3, group B = waves 4-7 (wave w and w+4 share a SIMD):
//     phase 2k   : A reads tile k  (+ issues DMA k+1)        | B multiplies tile k-1 (+ issues DMA k+1 first)
//     phase 2k+1 : A multiplies tile k                       | B reads tile k
// DMA(k+1) is issued by BOTH groups in phase 2k into the buffer of tile k-1, which A read in phase 2k-2 and B in phase
// 2k-1; each wave waits vmcnt(0) for its own pieces at the end of phase 2k+1, so everything has landed before A's reads
// in phase 2k+2. Two LDS buffers suffice, and every DMA has more than a full phase of MFMAs to land.
// The fragment reads are inline asm: hipcc's waitcnt insertion would drain vmcnt(0) in front of any ds_read it sees
// after an LDS-DMA it cannot prove disjoint, which serialises exactly what this schedule overlaps.
// Measured (profiles/r01/gemm_phases.txt, s_memtime stamps): k-tile period ~3.6k cycles vs 2.2k of pure MFMA issue; the
// rest is LDS-DMA issue (~65 cycles per 1 KiB piece and wave), the pieces' ~2.6k-cycle landing time and barrier skew.
// LDS-DMA alone sustains ~45 B/clk/CU (tools/dma_bench.hip), i.e. >= 1.4k cycles per 64 KiB k-tile.
// ------------------------------------------------------------------------------------------
template <class T, int EPI>
__global__ void __launch_bounds__(512)
gemm_pp_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp,
               unsigned long long* __restrict__ dbg /* tuning only: cycle stamps of block 0; null in production */,
               int diag /* tuning only (ATLAS_GEMM_DIAG): 1 = no epilogue, 2 = no k-loop; 0 in production */) {
    typedef typename T::elem E;
    constexpr int BCOL = 256, BTOK = 256, WT = 4, FA = 8, FB = 4;
    constexpr int EPC = 16 / (int)sizeof(E);
    constexpr uint32_t STG = 256 * 128;                              // bytes per operand stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // W0 | W1 | A0 | A1, 32 KiB each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WT, wj = wave % WT;
    const bool grpB = wave >= 4;
    const int64_t M = cu[n];
    const int ncol = N / BCOL;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ctile = jj % ncol;
    const int64_t ttile = (int64_t)(jj / ncol) * 8 + xcd;
    if (ttile * BTOK >= M) return;
    const int n0 = ctile * BCOL;
    const int64_t m0 = ttile * BTOK;
    const int lr = lane & 15, lg = lane >> 4;

    const int nk = (diag & 2) ? 0 : K / (8 * EPC);
    // this wave's DMA pieces: 4 x 8 rows of W and 4 x 8 rows of the activations per k-tile (8 rows x 128 B each,
    // source-side XOR swizzle as in gemm_bt_kernel)
    const int ch = (lane & 7) ^ (lane >> 3);
    const E* gw = W + (size_t)(n0 + wave * 32 + (lane >> 3)) * K + ch * EPC;       // piece i: + i * 8 rows
    const E* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t ar = m0 + wave * 32 + i * 8 + (lane >> 3);
        if (ar >= M) ar = M - 1;                                   // clamped: tail rows are never stored
        ga[i] = A + (size_t)ar * K + ch * EPC;
    }
    auto stage = [&](const int buf, const int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + (size_t)i * 8 * K + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + buf * STG + (wave * 32 + i * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + 2 * STG + buf * STG + (wave * 32 + i * 8) * 128), 16, 0, 0);
    };

    // LDS byte addresses of this lane's fragment chunks (fragment a / b adds a * 2048: rows 16 apart keep row & 7)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const uint32_t aw0 = lds0 + (wi * 128 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t aw1 = lds0 + (wi * 128 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;
    const uint32_t aa0 = lds0 + 2 * STG + (wj * 64 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t aa1 = lds0 + 2 * STG + (wj * 64 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // the tile's 256 bias values are parked in LDS (behind the four stage buffers) for the epilogue: read from global there,
    // they were the first thing every wave waited for after the k-loop (one cold L2 / HBM latency per tile)
    uint16_t* s_bias = (uint16_t*)(smem_raw + 4 * STG);
    uint2 bias_reg = make_uint2(0u, 0u);
    if constexpr (sizeof(E) == 2) {
        if (tid < 64) bias_reg = *(const uint2*)(bias + n0 + 4 * tid);
    }
    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's pieces of tile 0 have landed
    if constexpr (sizeof(E) == 2) {
        if (tid < 64) *(uint2*)(s_bias + 4 * tid) = bias_reg;
    }
    __builtin_amdgcn_s_barrier();
    if (grpB) {                                        // B's phase 0: nothing to multiply yet
        if (nk > 1) stage(1, 1);
        __builtin_amdgcn_s_barrier();
    }
    const bool stamp = dbg != nullptr && blockIdx.x == 0 && lane == 0;
#define PP_STAMP(i) do { if (stamp && kt < 16) dbg[(wave * 16 + kt) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    if (stamp && wave == 0) { dbg[1024] = __builtin_readcyclecounter(); dbg[1025] = wall_clock64(); }   // shader clock vs 100 MHz clock over the k-loop
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        PP_STAMP(0);
        PP_STAMP(1);
        // ---- read phase: all 24 fragments of tile kt, then lgkmcnt(0), in one asm block ----
        u4v fw0[FA], fa0[FB], fw1[FA], fa1[FB];
        {
            const uint32_t w0 = aw0 + buf * STG, w1 = aw1 + buf * STG, a0 = aa0 + buf * STG, a1 = aa1 + buf * STG;
            asm volatile(
                "ds_read_b128 %0, %24\n ds_read_b128 %1, %24 offset:2048\n ds_read_b128 %2, %24 offset:4096\n ds_read_b128 %3, %24 offset:6144\n"
                "ds_read_b128 %4, %24 offset:8192\n ds_read_b128 %5, %24 offset:10240\n ds_read_b128 %6, %24 offset:12288\n ds_read_b128 %7, %24 offset:14336\n"
                "ds_read_b128 %8, %25\n ds_read_b128 %9, %25 offset:2048\n ds_read_b128 %10, %25 offset:4096\n ds_read_b128 %11, %25 offset:6144\n"
                "ds_read_b128 %12, %26\n ds_read_b128 %13, %26 offset:2048\n ds_read_b128 %14, %26 offset:4096\n ds_read_b128 %15, %26 offset:6144\n"
                "ds_read_b128 %16, %26 offset:8192\n ds_read_b128 %17, %26 offset:10240\n ds_read_b128 %18, %26 offset:12288\n ds_read_b128 %19, %26 offset:14336\n"
                "ds_read_b128 %20, %27\n ds_read_b128 %21, %27 offset:2048\n ds_read_b128 %22, %27 offset:4096\n ds_read_b128 %23, %27 offset:6144\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fw0[0]), "=&v"(fw0[1]), "=&v"(fw0[2]), "=&v"(fw0[3]), "=&v"(fw0[4]), "=&v"(fw0[5]), "=&v"(fw0[6]), "=&v"(fw0[7]),
                  "=&v"(fa0[0]), "=&v"(fa0[1]), "=&v"(fa0[2]), "=&v"(fa0[3]),
                  "=&v"(fw1[0]), "=&v"(fw1[1]), "=&v"(fw1[2]), "=&v"(fw1[3]), "=&v"(fw1[4]), "=&v"(fw1[5]), "=&v"(fw1[6]), "=&v"(fw1[7]),
                  "=&v"(fa1[0]), "=&v"(fa1[1]), "=&v"(fa1[2]), "=&v"(fa1[3])
                : "v"(w0), "v"(a0), "v"(w1), "v"(a1)
                : "memory");
        }
        if (!grpB) { if (kt + 1 < nk) stage(buf ^ 1, kt + 1); }   // after the reads: issued first, the DMA competes with them (measured -4 %)
        else __builtin_amdgcn_s_waitcnt(0x0F70);       // B: its pieces of tile kt+1 (issued a phase ago) have landed
        PP_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(3);
        // ---- multiply phase ----
        if (grpB && kt + 2 < nk) stage(buf, kt + 2);   // into the buffer both groups have finished reading
        PP_STAMP(4);
        mma_tile<T, FA, FB>(fw0, fa0, acc);
        mma_tile<T, FA, FB>(fw1, fa1, acc);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(5);
        if (!grpB) __builtin_amdgcn_s_waitcnt(0x0F70); // A: its pieces of tile kt+1 have landed
        PP_STAMP(6);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(7);
    }
#undef PP_STAMP
    if (stamp && wave == 0) { dbg[1026] = __builtin_readcyclecounter(); dbg[1027] = wall_clock64(); dbg[1028] = (unsigned long long)nk; }
    if (!grpB) __builtin_amdgcn_s_barrier();           // A matches B's extra barrier: every wave is past its last LDS read
    if (diag & 1) { if (acc[0][0][0] == 12345.678f) C[0] = 0; return; }
    if constexpr (sizeof(E) == 2)
        gemm_epilogue_lds<T, EPI>(acc, smem_raw, s_bias, m0, n0, wave, lane, M, N, R, C, VT, cu, tokinfo, Lp);
    else
        gemm_epilogue<T, EPI, FA, FB>(acc, m0, n0, wi, wj, lr, lg, M, N, bias, R, C, VT, cu, tokinfo, Lp);
}

// ------------------------------------------------------------------------------------------
// The bulk GEMM as TWO co-resident workgroups per CU (gemm_co_kernel): 256 columns x 128 tokens per workgroup, 4 waves (one per
// SIMD, wave tile 128 x 64 as in gemm_pp_kernel), ONE 48 KiB LDS stage: all 24 fragments of a k-tile are read into registers,
// then the stage is refilled by LDS-DMA under the 64 MFMAs. A workgroup alone stalls on the landing time of its pieces; the
// second workgroup of the CU (own stage, own two barriers per k-tile) fills those gaps with its MFMAs. 64 KiB of LDS per
// workgroup (the epilogue's [128 tokens][256 columns] tile) and <= 256 registers keep two resident. Same k order per element
// as every other configuration: identical bits.
// Measured in the same run against gemm_pp_kernel (profiles/r01/gemm_fixed_cost.txt): QKV 253 vs 253 us, out-proj 102 vs 102,
// FFN-1 377 vs 394, FFN-2 297 vs 293 -> it serves FFN-1 (12 column tiles: most W reuse per activation tile). The hope that one
// workgroup's epilogue would hide under the other's k-loop did not come true: the two stay in lock-step; starting the second
// one half a tile late (first dispatch round, HW_ID wave slot parity) or giving the slots different s_setprio changed nothing.
// ------------------------------------------------------------------------------------------
template <class T, int EPI>
__global__ void __launch_bounds__(256, 2)
gemm_co_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp, int diag /* tuning only: 1 = no epilogue */) {
    typedef typename T::elem E;
    static_assert(sizeof(E) == 2, "16-bit dtypes only");
    constexpr int BCOL = 256, BTOK = 128, FA = 8, FB = 4;
    constexpr int EPC = 8;
    constexpr uint32_t AOFF = 256 * 128;                             // W rows | activation rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t M = cu[n];
    const int ncol = N / BCOL;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ctile = jj % ncol;
    const int64_t ttile = (int64_t)(jj / ncol) * 8 + xcd;
    if (ttile * BTOK >= M) return;
    const int n0 = ctile * BCOL;
    const int64_t m0 = ttile * BTOK;
    const int lr = lane & 15, lg = lane >> 4;
    const int nk = K / (8 * EPC);
    const int ch = (lane & 7) ^ (lane >> 3);
    const E* gw = W + (size_t)(n0 + wave * 64 + (lane >> 3)) * K + ch * EPC;       // piece i: + i * 8 rows
    const E* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t ar = m0 + wave * 32 + i * 8 + (lane >> 3);
        if (ar >= M) ar = M - 1;
        ga[i] = A + (size_t)ar * K + ch * EPC;
    }
    auto stage = [&](const int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + (size_t)i * 8 * K + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + (wave * 64 + i * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + AOFF + (wave * 32 + i * 8) * 128), 16, 0, 0);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const uint32_t w0 = lds0 + (wi * 128 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t w1 = lds0 + (wi * 128 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;
    const uint32_t a0 = lds0 + AOFF + (wj * 64 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t a1 = lds0 + AOFF + (wj * 64 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    uint16_t* s_bias = (uint16_t*)(smem_raw + 64 * 1024);
    uint2 bias_reg = make_uint2(0u, 0u);
    if (tid < 64) bias_reg = *(const uint2*)(bias + n0 + 4 * tid);
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's pieces of tile kt have landed
        if (kt == 0 && tid < 64) *(uint2*)(s_bias + 4 * tid) = bias_reg;
        __builtin_amdgcn_s_barrier();                  // ... everybody's
        u4v fw0[FA], fa0[FB], fw1[FA], fa1[FB];
        asm volatile(
            "ds_read_b128 %0, %24\n ds_read_b128 %1, %24 offset:2048\n ds_read_b128 %2, %24 offset:4096\n ds_read_b128 %3, %24 offset:6144\n"
            "ds_read_b128 %4, %24 offset:8192\n ds_read_b128 %5, %24 offset:10240\n ds_read_b128 %6, %24 offset:12288\n ds_read_b128 %7, %24 offset:14336\n"
            "ds_read_b128 %8, %25\n ds_read_b128 %9, %25 offset:2048\n ds_read_b128 %10, %25 offset:4096\n ds_read_b128 %11, %25 offset:6144\n"
            "ds_read_b128 %12, %26\n ds_read_b128 %13, %26 offset:2048\n ds_read_b128 %14, %26 offset:4096\n ds_read_b128 %15, %26 offset:6144\n"
            "ds_read_b128 %16, %26 offset:8192\n ds_read_b128 %17, %26 offset:10240\n ds_read_b128 %18, %26 offset:12288\n ds_read_b128 %19, %26 offset:14336\n"
            "ds_read_b128 %20, %27\n ds_read_b128 %21, %27 offset:2048\n ds_read_b128 %22, %27 offset:4096\n ds_read_b128 %23, %27 offset:6144\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(fw0[0]), "=&v"(fw0[1]), "=&v"(fw0[2]), "=&v"(fw0[3]), "=&v"(fw0[4]), "=&v"(fw0[5]), "=&v"(fw0[6]), "=&v"(fw0[7]),
              "=&v"(fa0[0]), "=&v"(fa0[1]), "=&v"(fa0[2]), "=&v"(fa0[3]),
              "=&v"(fw1[0]), "=&v"(fw1[1]), "=&v"(fw1[2]), "=&v"(fw1[3]), "=&v"(fw1[4]), "=&v"(fw1[5]), "=&v"(fw1[6]), "=&v"(fw1[7]),
              "=&v"(fa1[0]), "=&v"(fa1[1]), "=&v"(fa1[2]), "=&v"(fa1[3])
            : "v"(w0), "v"(a0), "v"(w1), "v"(a1)
            : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave holds its fragments: the stage may be overwritten
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<T, FA, FB>(fw0, fa0, acc);
        mma_tile<T, FA, FB>(fw1, fa1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (diag & 1) { if (acc[0][0][0] == 12345.678f) C[0] = 0; return; }
    gemm_epilogue_lds<T, EPI, BTOK>(acc, smem_raw, s_bias, m0, n0, wave, lane, M, N, R, C, VT, cu, tokinfo, Lp);
}

// ------------------------------------------------------------------------------------------
// gemm_pt_kernel: the bulk GEMM as a PERSISTENT kernel -- one 8-wave workgroup per CU walks over its share of the 256 x 256 tiles with
// the ping-pong k-loop of gemm_pp_kernel running straight THROUGH the tile boundaries, and an epilogue that needs no LDS and no barrier.
// Why (profiles/r01/gemm_fixed_cost.txt): with one launch-scheduled workgroup per tile, everything between two tiles' k-loops was
// exposed -- workgroup dispatch, the dependent load of the token count, the first LDS-DMA pieces' round trip to L2 / HBM, and a 6 us
// epilogue that parked the C tile in LDS (it occupies all four stage buffers) behind two barriers: ~10 us per tile whatever its kind,
// 365 us of a 1 045 us layer with the k-loops switched off. Here
//   * the LDS-DMA of the next tile's first two k-tiles is issued from inside the current tile's last two iterations (the k index simply
//     runs on: group A stages its pieces of k-tile g+1 while it reads k-tile g, group B its pieces of k-tile g+2), so a tile's first MFMAs
//     find their operands in LDS. EVERY piece goes out from a read phase, between the phase's fragment reads (round 4: see `stage`);
//   * the C tile leaves STRAIGHT FROM THE ACCUMULATOR REGISTERS as 16-byte stores. A lane of v_mfma_f32_16x16x32 holds 4 consecutive
//     MFMA rows of one MFMA column per fragment; which weight row an MFMA row IS, is decided by the LDS-DMA source address alone. The
//     weight rows are therefore staged permuted -- LDS row 16 a + i of a wave's 128 holds weight row 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1)
//     + (i & 3) -- which makes the lane's registers of a fragment pair (2 j, 2 j + 1) 8 CONSECUTIVE output columns 32 j + 8 lg .. + 7 of
//     token lr: one dwordx4 store, a wave instruction = 16 tokens x 64 contiguous bytes. LDS addresses, swizzle and the k order of every
//     element are those of gemm_pp_kernel (bit-identical results); only the row the DMA fetches differs;
//   * V tiles of the QKV projection swap the MFMA operands (activations as the A operand: the same products summed in the same order)
//     and permute the staged TOKEN rows instead, so a lane holds 8 consecutive keys of one V^T row: again one 16-byte store;
//   * group A runs its epilogue while group B multiplies the tile's last k-tile, group B while group A multiplies the next tile's first;
//     epilogue stores are issued behind the wave's last wait of the iteration, so the next k-tiles never wait for them;
//   * operands go through buffer descriptors (one per tile, in SGPRs): rows past the token count read as nothing and are dropped on
//     store by the hardware bounds check, and a lane's address state is 4 VGPRs for the whole kernel.
// Tile order as in the launch-per-tile kernels: token tile t belongs to XCD t % 8 (workgroup b runs on XCD b % 8: speed only), whose
// 32 workgroups walk its (token tile, column tile) pairs in order, so one token tile's column tiles run side by side on one L2.
// ------------------------------------------------------------------------------------------
typedef unsigned int pt_u4 __attribute__((ext_vector_type(4)));
template <int B, int E, class F>
static __device__ __forceinline__ void pt_static_for(F&& f) {           // f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>)
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); pt_static_for<B + 1, E>(f); }
}
template <bool C, class A_, class B_>
static __device__ __forceinline__ auto& pt_pick(A_& a, B_& b) { if constexpr (C) return a; else return b; }
template <int OFF>
static __device__ __forceinline__ void pt_ds_read(u4v& dst, const uint32_t addr) {      // issued, NOT waited for
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// the lane id, recomputed where it is called (volatile: neither hoisted nor merged): what a tile's epilogue derives from the lane is then
// not carried through the k-loop, which has no register to spare (every value kept alive across it ended up in scratch)
static __device__ __forceinline__ int pt_fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
template <class T> static __device__ __forceinline__ uint32_t pack2(const float a, const float b) {       // two values -> one word of the model dtype (RNE)
    typedef float f2v __attribute__((ext_vector_type(2)));
    if constexpr (T::DT == ATLAS_DT_F16) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2v){a, b}, h2v));                     // v_cvt_pk_f16_f32
    } else {
        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2v){a, b}, b2v));                     // v_cvt_pk_bf16_f32
    }
}

// acc -> global memory for one wave. EPI 1 | 2 | 3 (plain, the q | k columns of the QKV projection): acc[a][b][r] = C[token 16 b + lr]
// [column 32 (a >> 1) + 8 lg + 4 (a & 1) + r] of the wave's 128 x 64; EPI 4 (the V columns, stored transposed): acc[a][b][r] =
// C[token 32 (b >> 1) + 8 lg + 4 (b & 1) + r][column 16 a + lr]. Rounding points are those of gemm_epilogue: dt(acc + bias), then [gelu] /
// [+ residual] on that, rounded by the store.
// EPI 1-3: what a store costs the wave is set by how many ROWS its 64 lanes touch, not by its bytes (tools/store_bench.hip, one 128 KiB
// tile per CU: 16 rows x 64 B per instruction = 3.2 us, 2 rows x 512 B = 0.85 us), and the fragment layout gives 16 rows x 64 B. So each
// 16-token slice of the wave's tile takes a turn through a WAVE-PRIVATE 4 KiB of LDS (no barrier: a wave's LDS operations execute in
// order) and leaves as 4 rows x 256 B per instruction; the residual of EPI 2 is fetched and added in that layout too.
//   s_tr: this wave's 4 KiB: [16 tokens][16 chunks of 8 columns], chunk c of token t at 16-byte slot c ^ t (conflict-free both ways)
//   cbuf: descriptor of C's rows [m0, M) (stores past M fall out of bounds);  bq4: the lane's 8 bias values of fragment pair j (packed)
//   rv: EPI 2: the residual piece of (slice b, store s) = rv[4 b + s];  EPI 4: bq4 word a = bias of column 16 a + lr, tki: see below
template <class T, int EPI, int AUX = 0>      // AUX: cache-policy bits of the C stores (0 = default; tuning builds A/B 2 = nt and 16 = sc1)
static __device__ __forceinline__ void pt_epilogue(const f4 (&acc)[8][4], unsigned char* __restrict__ s_tr,
                                                   const pt_u4 (&bq4)[4], const pt_u4 (&rv)[16], const int2 (&tki)[4],
                                                   const __amdgpu_buffer_rsrc_t cbuf,
                                                   const int64_t m0, const int n0, const int wi, const int wj,
                                                   const int64_t M, const int N, uint16_t* __restrict__ VT, const int2* __restrict__ tokinfo, const int Lp) {
    const int lane = pt_fresh_lane(), lr = lane & 15, lg = lane >> 4;
    if constexpr (EPI != 4) {
        const int tq = lane >> 4, cc = lane & 15;                       // transposed side: token 4 s + tq of the slice, column chunk cc
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bw[4] = {bq4[j][0], bq4[j][1], bq4[j][2], bq4[j][3]};
                uint4 o;
                uint32_t* ow = (uint32_t*)&o;
                if constexpr (EPI == 1) {
                    // erf GELU in fp32 on the Linear output in the model dtype (common.h), the lane's four pairs breadth first (gelu_erf_poly2x4)
                    gelu_f2 vin[4], g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f4 s = acc[2 * j + (e >> 1)][b];
                        vin[e] = (gelu_f2){T::rnd(((e & 1) ? s[2] : s[0]) + T::ld((uint16_t)(bw[e] & 0xffff))),
                                           T::rnd(((e & 1) ? s[3] : s[1]) + T::ld((uint16_t)(bw[e] >> 16)))};
                    }
                    gelu_erf_poly2x4(vin, g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ow[e] = pack2<T>(g[e].x, g[e].y);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                       // columns 2 e, 2 e + 1 of the lane's eight
                        const f4 s = acc[2 * j + (e >> 1)][b];
                        const float v0 = ((e & 1) ? s[2] : s[0]) + T::ld((uint16_t)(bw[e] & 0xffff));
                        const float v1 = ((e & 1) ? s[3] : s[1]) + T::ld((uint16_t)(bw[e] >> 16));
                        ow[e] = pack2<T>(v0, v1);
                    }
                }
                *(uint4*)(s_tr + lr * 256 + (((4 * j + lg) ^ lr) * 16)) = o;
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int tok = 4 * sidx + tq;
                const uint4 t = *(const uint4*)(s_tr + tok * 256 + ((cc ^ tok) * 16));
                pt_u4 o = {t.x, t.y, t.z, t.w};
                if (EPI == 2) {                                         // + input_tensor
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t rw = rv[b * 4 + sidx][e];
                        o[e] = pack2<T>(T::ld((uint16_t)(o[e] & 0xffff)) + T::ld((uint16_t)(rw & 0xffff)), T::ld((uint16_t)(o[e] >> 16)) + T::ld((uint16_t)(rw >> 16)));
                    }
                }
                // (the whole offset in the VGPR, soffset 0: for a 16-byte buffer store with an SGPR soffset hipcc does not pad the wait state the
                //  store's data registers need before the next VALU write -- measured on gfx950: tools/hazard_probe.hip)
                const uint32_t vo = (uint32_t)(((wj * 64 + b * 16 + tok) * N + n0 + wi * 128 + 8 * cc) * 2);
                __builtin_amdgcn_raw_buffer_store_b128(o, cbuf, (int)vo, 0, AUX);
            }
        }
    } else {
        // V^T[passage][h*64+d][column]: the lane's 8 consecutive tokens of a fragment pair are one aligned 16-byte run of a row whenever
        // they belong to one passage (layout note at the top of the file), ragged batches included; the few groups that straddle two
        // passages or the end of the batch go out token by token
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int64_t tok0 = m0 + wj * 64 + 32 * jj + 8 * lg;
            const int2 t0 = tki[2 * jj], t7 = tki[2 * jj + 1];
            const bool whole = (tok0 + 7 < M) && (t0.x == t7.x);
            // column of token tok0 in its passage's rows: tok0 - (cu[passage] & ~7), with cu[passage] = tok0 - rank
            const int64_t off0 = (int64_t)t0.x * HID * Lp + (tok0 - ((tok0 - (t0.y >> 16)) & ~(int64_t)7));
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float bv = T::ld((uint16_t)(bq4[a >> 1][(a & 1) * 2] & 0xffff));    // (words 0 and 2 of bq4[a >> 1]: see the loads)
                const f4 s0 = acc[a][2 * jj], s1 = acc[a][2 * jj + 1];
                const uint4 v = make_uint4(pack2<T>(s0[0] + bv, s0[1] + bv), pack2<T>(s0[2] + bv, s0[3] + bv),
                                           pack2<T>(s1[0] + bv, s1[1] + bv), pack2<T>(s1[2] + bv, s1[3] + bv));
                const int64_t crow = (int64_t)(n0 + wi * 128 + a * 16 + lr) * Lp;
                if (whole) {
                    *(uint4*)(VT + off0 + crow) = v;
                } else {
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll 1
                    for (int e = 0; e < 8; ++e)
                        if (tok0 + e < M) {
                            const int2 t = tokinfo[tok0 + e];
                            const int64_t tok = tok0 + e;
                            VT[(int64_t)t.x * HID * Lp + (tok - ((tok - (t.y >> 16)) & ~(int64_t)7)) + crow] = (uint16_t)(w4[e >> 1] >> ((e & 1) * 16));
                        }
                }
            }
        }
    }
}

// EPI 1: C = dt(gelu(dt(acc + bias)))   2: C = dt(dt(acc + bias) + R)   3: C = dt(acc + bias)   4: V^T = dt(acc + bias), transposed per passage
// (the QKV projection is two launches: its q | k columns with EPI 3 into [M, 1536], its v columns with EPI 4 into V^T)
template <class T, int EPI>
__global__ void __launch_bounds__(512)
gemm_pt_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp,
               int diag /* tuning build only: bit 0 = no epilogue, bits 2-3 = store policy (1 nt, 2 sc1), bit 4 / 5 = activation / weight loads aliased to the first tile, bit 6 = no MFMAs, bit 7 = no LDS-DMA pieces, bits 8.. = start stagger; 0 in production */,
               unsigned long long* __restrict__ dbg /* tuning build only: 100 MHz stamps of workgroup 0 around its tile boundaries; null in production */) {
    typedef typename T::elem E;
    static_assert(sizeof(E) == 2, "16-bit dtypes only");
#if ATLAS_TUNING
    // (bit 0 of the pointer: only the two stamps at the ends of workgroup 0 -- both clocks over an UNPERTURBED kernel: every other stamp is a
    //  store in the counted vmcnt stream)
    const bool ends_only = ((uintptr_t)dbg & 1) != 0;
    unsigned long long* const dbgp = (unsigned long long*)((uintptr_t)dbg & ~(uintptr_t)1);
    int tstamp = 0;                                                  // tile counter of the stamps
#define PT_STAMP(i) do { if (dbgp != nullptr && !ends_only && blockIdx.x == 0 && tstamp < 8 && pt_fresh_lane() == 0) dbgp[((int)wave * 8 + tstamp) * 16 + (i)] = wall_clock64(); } while (0)
    int itc = 0;                                                     // iteration counter of the per-iteration stamps: shader cycles, iterations 24 .. 55
#define PT_ISTAMP(i) do { if (dbgp != nullptr && !ends_only && blockIdx.x == 0 && itc >= 24 && itc < 56 && pt_fresh_lane() == 0) dbgp[2048 + ((int)wave * 32 + itc - 24) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PT_STAMP(i) do { } while (0)
#define PT_ISTAMP(i) do { } while (0)
#endif
    constexpr int FA = 8, FB = 4;
    constexpr bool VTR = (EPI == 4);                                 // V tile: token rows staged permuted, MFMA operands swapped
    constexpr uint32_t STG = 256 * 128;                              // bytes per operand stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // W0 | W1 | A0 | A1, 32 KiB each | 8 x 4 KiB: one transposition slot per wave = all 160 KiB
    constexpr uint32_t TR_OFF = 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;
    const bool grpB = wave >= 4;
    const int64_t M = cu[n];
    const int ncol = N >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int ntt = (int)((M + 255) >> 8);
    const int njobs = (ntt > xcd ? (ntt - xcd + 7) >> 3 : 0) * ncol;        // (token tile, column tile) pairs of this XCD
    if (slot >= njobs) return;
#if ATLAS_TUNING
    // (both clocks at the two ends of workgroup 0: what the shader clock is under this kernel's load)
    if (dbgp != nullptr && blockIdx.x == 0 && tid == 0) { dbgp[1024] = wall_clock64(); dbgp[1025] = __builtin_readcyclecounter(); }
#endif
    const int nk = K >> 6;                                                    // k-tiles of 128 bytes per tile (>= 2)
    const uint32_t K2 = (uint32_t)K * 2u;

    // tile j of this XCD: column tile j % ncol of token tile (j / ncol) * 8 + xcd. Operands go through buffer descriptors: W rows
    // [n0, n0 + 256) and rows [m0, M) of the activations / the output / the residual (all SGPR arithmetic, redone where it is needed
    // rather than carried: the kernel has no scalar registers to spare)
    auto tile_n0 = [&](const int j) { return (j % ncol) << 8; };
    auto tile_m0 = [&](const int j) { return (int64_t)((j / ncol) * 8 + xcd) << 8; };
    auto rows_rsrc = [&](const E* base, const int64_t m0, const int ld) {      // rows [m0, M) of a [M, ld] tensor
        // (a descriptor that ends with the TILE -- one 32-bit product instead of the clamped 64-bit one -- was tried in round 5: hipcc answers
        //  with vector compares and 35 % more instructions in the kernel; left as it is)
        int64_t rem = (M - m0) * (int64_t)ld * 2;
        if (rem > 0xfffffff0ll) rem = 0xfffffff0ll;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)m0 * ld), 0, (int)rem, 0x00020000);
    };

    // LDS-DMA: a wave instruction writes 8 LDS rows x 128 B, lane-linear; wave w stages LDS rows 32 w + 8 i + (lane >> 3), i = 0..3, of
    // both operands. The bank swizzle goes on the SOURCE chunk (as in gemm_bt_kernel), and so does the row permutation (header):
    //   identity      source row = LDS row                                                                           (piece i: + 8 i)
    //   permuted W    128-row groups: 32 (w & 3) + 16 (i & 1) + 8 (lane >> 5) + 4 (i >> 1) + ((lane >> 3) & 3)      [EPI 1, 2, 3]
    //   permuted A     64-row groups: 32 (w & 1) + 16 (i & 1) + 8 (lane >> 5) + 4 (i >> 1) + ((lane >> 3) & 3)      [EPI 4]
    // The activation offsets are complete in the bounds-checked voffset (rows past M are not fetched); weight rows are always there,
    // so their piece offset rides in the scalar offset.
    const uint32_t chb = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    const uint32_t perm_lane = (uint32_t)(8 * (lane >> 5) + ((lane >> 3) & 3));
    // Which rows a wave stages (round 4; before, wave w staged LDS rows [32 w, 32 w + 32) of both operands, group A in its read phase, group B in
    // front of its MFMAs): every piece is issued from a READ phase, between the phase's fragment reads -- in front of a group's own MFMAs a piece
    // costs ~100 cycles of an idle matrix pipe, queued behind the other group's pieces. What makes that possible:
    //   * W rows 0..127 of a stage are read by group A only, rows 128..255 by group B only: each half is refilled by the OTHER group in its next
    //     read phase (wave w takes the rows wave w ^ 4 used to: A the half-1 rows of k-tile g + 1 while it reads k-tile g, B the half-0 rows
    //     of k-tile g + 2 while it reads k-tile g);
    //   * activation rows [64 wj, 64 wj + 64) are read by the two waves (wj, wj + 4) of one SIMD only: wave wj + 4 refills the upper 32 of them
    //     for k-tile g + 2 right behind its own reads of k-tile g (its twin read them a phase earlier), wave wj the lower 32 a phase later.
    // 8 pieces per wave and phase, landing behind COUNTED s_waitcnt vmcnt (vector-memory operations of a wave complete in order).
    const int we = wave ^ 4, wa = 2 * wj + (grpB ? 1 : 0);          // whose rows this wave stages: W | activations (in units of 32 LDS rows)
    const uint32_t vw = VTR ? (uint32_t)(we * 32 + (lane >> 3)) * K2 + chb
                            : ((uint32_t)(128 * (we >> 2) + 32 * (we & 3)) + perm_lane) * K2 + chb;
    const uint32_t va = VTR ? ((uint32_t)(64 * (wa >> 1) + 32 * (wa & 1)) + perm_lane) * K2 + chb
                            : (uint32_t)(wa * 32 + (lane >> 3)) * K2 + chb;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // this wave's pieces of k-tile kt of tile j -> stage buffer buf; `between(i)` runs behind piece i. Group A: its activation pieces first (read
    // a phase and a half on: `vmcnt(4)` at the end of its multiply phase covers them), group B: its W pieces first (its activation pieces
    // overwrite rows it reads in this very phase: they go out behind those reads)
    // `mask_tag`: which of the eight pieces go out (bit p8; `between` runs for every slot either way). All of them in the product: ATLAS_PT_WDEFER
    // (round 5 experiment, default 0) takes the last WD of a wave's four W pieces out of its read phase and issues them BETWEEN ITS OWN MFMAs.
    // Why it was tried: a wave alone gets a ds_read_b128 through every ~30 cycles (tools/lds_read_probe.hip: 4 waves x 24 reads = 720 cycles,
    // whatever the other group does), every piece lengthens the read phase further, and tools/pt_cycles.py puts the k-tile at ~3 200 cycles
    // against 2 048 of MFMA issue -- the read phase, not the multiply phase, is the half period. What came out (same-process A/B of the builds,
    // bit-identical outputs, profiles/r05/enc_wdefer_ab.txt): 13.13 ms -> 13.63 (WD 4) / 13.75 (WD 2, 3) per 512 x 128 batch. A piece issued
    // by the multiplying wave stalls its MFMA issue for longer than the same piece costs a reading wave (round 4 found the same from the other side).
    auto stage = [&](auto grp_tag, auto mask_tag, const int buf, const int j, const int kt, auto&& between) __attribute__((always_inline)) {
        constexpr bool GB = decltype(grp_tag)::value;
        constexpr int MASK = decltype(mask_tag)::value;
        const uint32_t kb = (uint32_t)kt * 128u;
#if ATLAS_TUNING
        // experiment (results wrong, timing only): bit 4 = every tile loads the activations of the XCD's FIRST token tile, bit 5 = the weights of
        // column tile 0 -- the same instruction stream with that operand always an L2 hit: what part of the k-tile period is the operand's way in?
        const int ja = (diag & 16) ? (j % ncol) : j, jw = (diag & 32) ? (j - j % ncol) : j;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)tile_n0(jw) * K), 0, (int)(256u * K2), 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = rows_rsrc(A, tile_m0(ja), K);
#else
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)tile_n0(j) * K), 0, (int)(256u * K2), 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = rows_rsrc(A, tile_m0(j), K);
#endif
        unsigned char* const lw = smem_raw + buf * STG + (we * 32) * 128;
        unsigned char* const la = smem_raw + 2 * STG + buf * STG + (wa * 32) * 128;
        pt_static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
            constexpr int p8 = decltype(ic)::value;
            constexpr bool isw = GB ? (p8 < 4) : (p8 >= 4);
            constexpr int i = p8 & 3;
            if constexpr (((MASK >> p8) & 1) == 0) {
            } else if constexpr (isw) {
                constexpr uint32_t rw_ = (uint32_t)(VTR ? 8 * i : 16 * (i & 1) + 4 * (i >> 1));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(lw + i * 8 * 128), 16, (int)vw, (int)(rw_ * K2 + kb), 0, 0);
            } else {
                constexpr uint32_t ra_ = (uint32_t)(VTR ? 16 * (i & 1) + 4 * (i >> 1) : 8 * i);
                uint32_t vo = va;                      // (from a copy hipcc cannot hoist: hoisted, the four piece offsets live across the k-loop)
                asm volatile("" : "+v"(vo));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(la + i * 8 * 128), 16, (int)(vo + ra_ * K2), (int)kb, 0, 0);
            }
            between(ic);
        });
    };
    auto nothing = [](auto) __attribute__((always_inline)) {};
    constexpr int WD = VTR ? 0 : ATLAS_PT_WDEFER;                       // W pieces per wave and k-tile issued from the MFMA phase (0 .. 4; the V^T kernel of tuning cfg 10 keeps the old order)
    static_assert(WD >= 0 && WD <= 4 && ATLAS_PT_WSTRIDE * (WD - 1) < 8, "ATLAS_PT_WDEFER / ATLAS_PT_WSTRIDE");
    typedef std::integral_constant<int, 0xFF> all_pieces;
    typedef std::integral_constant<int, 0xFF & ~(((1 << WD) - 1) << (8 - WD))> read_pieces_a;      // group A: W pieces are slots 4..7
    typedef std::integral_constant<int, 0xFF & ~(((1 << WD) - 1) << (4 - WD))> read_pieces_b;      // group B: W pieces are slots 0..3
    // W piece i (0..3) of k-tile kt of tile j -> stage buffer buf, by itself (the deferred ones, from the MFMA phase)
    auto stage_w_deferred = [&](auto ic, const int buf, const int j, const int kt) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)tile_n0(j) * K), 0, (int)(256u * K2), 0x00020000);
        constexpr uint32_t rw_ = (uint32_t)(VTR ? 8 * i : 16 * (i & 1) + 4 * (i >> 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem_raw + buf * STG + (we * 32) * 128 + i * 8 * 128), 16, (int)vw, (int)(rw_ * K2 + (uint32_t)kt * 128u), 0, 0);
    };

    // LDS byte addresses of this lane's fragment chunks (fragment a / b adds a * 2048: rows 16 apart keep row & 7)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int lr0 = lane & 15, lg0 = lane >> 4;
    const uint32_t aw0 = lds0 + (wi * 128 + lr0) * 128 + ((0 + lg0) ^ (lr0 & 7)) * 16;

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    int jc = slot;                                     // the tile being multiplied; the next one is jc + nslots
    // EPI 2: the residual rows the epilogue adds were written a kernel or more ago and come from the Infinity Cache / HBM: ~2 us that
    // both groups used to sit out between their last MFMA and their epilogue. Three iterations before the end, right behind its own
    // wait (so the requests have a whole iteration to land before the wave waits again), every wave touches the 128 cache lines of
    // its 64 x 128 residual piece with two 4-byte LDS-DMA loads per lane into its own (idle) transposition slot: the lines are in L2
    // when the epilogue asks for them
    auto touch_residual = [&]() {
        const __amdgpu_buffer_rsrc_t rr = rows_rsrc(R, tile_m0(jc), N);
        const int fl = pt_fresh_lane();                // (not the kernel's `lane`: nothing lane-derived is carried through the k-loop)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = i * 64 + fl;              // 128-byte line `line & 1` of row `line >> 1` of the wave's 64 x 256 B
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr)(smem_raw + TR_OFF + wave * 4096 + i * 256), 4,
                                                     (int)((((wj * 64 + (line >> 1)) * N + tile_n0(jc) + wi * 128) * 2) + (line & 1) * 128), 0, 0, 0);
        }
    };
#if ATLAS_TUNING
    if (diag >> 8) {                                   // experiment: workgroups start in four classes, (diag >> 8) x 0.25 us apart (are the CUs' store bursts the epilogue's cost?)
        const unsigned long long until = wall_clock64() + (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned long long)(diag >> 8) * 25ull;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
#endif
    if (!grpB) stage(std::false_type{}, all_pieces{}, 0, jc, 0, nothing); else stage(std::true_type{}, all_pieces{}, 0, jc, 0, nothing);
    if ((ATLAS_PT_RSPLIT != 0 && !VTR) && grpB) stage(std::true_type{}, all_pieces{}, 1, jc, 1, nothing);      // (read half a phase earlier there: they land under this first wait)
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's pieces of k-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grpB) {                                        // B's phase 0: nothing to multiply yet; its pieces of k-tile 1
        if (!(ATLAS_PT_RSPLIT != 0 && !VTR)) stage(std::true_type{}, all_pieces{}, 1, jc, 1, nothing);
        __builtin_amdgcn_s_barrier();
    }
    int buf = 0;
    bool skip_wait = false;                            // B: its wait of a tile's first iteration was taken before the epilogue
    bool skip_b1 = false;                              // A: its first barrier of a tile was taken before the epilogue
    // READ-AHEAD (as gscan_kernel.h does it): group A, whose tile ends a phase before group B's, could first READ the next tile's first k-tile --
    // beside group B's last MFMAs, the fragments waiting in registers -- and run its epilogue behind that phase's barrier, in front of its own
    // MFMAs: the phases stay aligned across the tile boundary (no extra barrier, group B's pieces never deferred). Built and NOT kept: with
    // 96 fragment registers alive beside the 128 accumulators the epilogues spill (EPI 3: 152 B, EPI 4: 188 B, EPI 1: 216 B of scratch; EPI 2's
    // 16 residual pieces rule it out from the start), and a kernel with a private segment pays ~12 us per launch.
    // (git history: "gemm_pt: the last iteration of a tile waits for its activation pieces only" carries the code.)
    pt_u4 rv[16], bq[4];
    int2 tki[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) rv[i] = (pt_u4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) tki[i] = make_int2(0, 0);

    // One iteration = one k-tile; phases as in gemm_pp_kernel. The k-tiles staged here (one and two steps on) may be the next tile's.
    auto run_epilogue = [&](const int j) __attribute__((always_inline)) {       // tile j leaves the accumulators; they start the next tile from zero
#define PT_EPILOGUE(AUX) pt_epilogue<T, EPI, AUX>(acc, smem_raw + TR_OFF + wave * 4096, bq, rv, tki, rows_rsrc(C, tile_m0(j), N), tile_m0(j), tile_n0(j), wi, wj, M, N, VT, tokinfo, Lp)
#if ATLAS_TUNING
        if (diag & 1) { if (acc[0][0][0] == 12345.678f) C[0] = 0; }
        else if ((diag & 12) == 4) PT_EPILOGUE(2);
        else if ((diag & 12) == 8) PT_EPILOGUE(16);
        else
#endif
        PT_EPILOGUE(0);
#undef PT_EPILOGUE
        __builtin_amdgcn_sched_barrier(0);
        PT_STAMP(7);
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
            for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
    };
    // ATLAS_PT_RSPLIT (round 5): the read phase, not the multiply phase, is the half period (a lone wave per SIMD gets a ds_read_b128 through
    // every ~30 cycles: 24 reads = 720 cycles before the 8 pieces, against 1 024 cycles of MFMA issue; tools/lds_read_probe.hip, pt_cycles.py).
    // So a k-tile's k-step-0 fragments (12 of the 24 reads) are read a phase AHEAD: between the wave's own MFMAs of the previous k-tile's k-step 1,
    // into the registers its k-step-0 MFMAs have just released -- while the SIMD's other wave is in ITS read phase (two waves reading at once is
    // the fast case). A tile's first iteration reads all 24 in its read phase, its last one reads nothing ahead (the epilogue needs the registers).
    // What has to have landed earlier for that: the OTHER group's pieces of the next k-tile by the end of its multiply phase (both groups now
    // wait for all of their pieces there), a wave's own activation pieces by the middle of its multiply phase (vmcnt(4) in front of the four
    // activation reads; the eight W reads go first).
    constexpr bool RS = !VTR && ATLAS_PT_RSPLIT != 0;
    u4v cfw0[RS ? FA : 1], cfa0[RS ? FB : 1];                          // (RS) the k-step-0 fragments, carried from one iteration's multiply phase to the next
    auto iteration = [&](const int kt, auto last_tag, auto first_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool HALF = RS && !decltype(first_tag)::value;       // the read phase reads k-step 1 only
        const bool has_next = jc + nslots < njobs;
        PT_ISTAMP(0);
        u4v lfw0[RS ? 1 : FA], lfa0[RS ? 1 : FB], fw1[FA], fa1[FB];
        auto& fw0 = pt_pick<RS>(cfw0, lfw0);
        auto& fa0 = pt_pick<RS>(cfa0, lfa0);
        // (one lane-dependent address lives across the k-loop; k-step 1 = chunk (4 + lg) ^ (lr & 7) = k-step 0's with bit 2 flipped: byte
        //  address ^ 64 -- the dynamic LDS segment starts at a multiple of 128 --, the activations sit a wave-uniform distance behind the weights)
        const uint32_t w0 = aw0 + buf * STG, w1 = w0 ^ 64u, a0 = w0 + (uint32_t)(2 * STG + (wj * 64 - wi * 128) * 128), a1 = a0 ^ 64u;
        // what this wave stages in this phase. Group A: the k-tile after this one (the other stage). Group B: the one after that, into THIS stage
        // -- except in a tile's first iteration, where its reads share the phase with group A's (both groups come out of their epilogues side by
        // side, see the tile loop): there its pieces wait for the barrier behind its reads and go out in front of its MFMAs.
        const bool b_first = grpB && skip_wait;
        bool stages; int sj, skt;
        if (!grpB) { stages = !LAST || has_next; sj = LAST ? jc + nslots : jc; skt = LAST ? 0 : kt + 1; }
        else { stages = !b_first && (kt + 2 < nk || has_next); sj = kt + 2 < nk ? jc : jc + nslots; skt = kt + 2 < nk ? kt + 2 : kt + 2 - nk; }
#if ATLAS_TUNING
        if (diag & 128) stages = false;                // experiment (timing only): no LDS-DMA pieces after the prologue's -- fragment reads, barriers and MFMAs alone
#endif
        // (every fragment register is an in/out operand of the phase's last wait: nothing reads one in front of it; the reads are single
        //  asm statements so that they can sit between the pieces, and the registers they fill asynchronously must not be copied or spilled:
        //  tests/test_kernel_isa.py)
#define PT_READS_DONE_ALL() asm volatile("s_waitcnt lgkmcnt(0)" \
            : "+v"(fw0[0]), "+v"(fw0[1]), "+v"(fw0[2]), "+v"(fw0[3]), "+v"(fw0[4]), "+v"(fw0[5]), "+v"(fw0[6]), "+v"(fw0[7]), \
              "+v"(fw1[0]), "+v"(fw1[1]), "+v"(fw1[2]), "+v"(fw1[3]), "+v"(fw1[4]), "+v"(fw1[5]), "+v"(fw1[6]), "+v"(fw1[7]), \
              "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fa1[0]), "+v"(fa1[1]), "+v"(fa1[2]), "+v"(fa1[3]) :: "memory")
#define PT_READS_DONE_1() asm volatile("s_waitcnt lgkmcnt(0)" \
            : "+v"(fw1[0]), "+v"(fw1[1]), "+v"(fw1[2]), "+v"(fw1[3]), "+v"(fw1[4]), "+v"(fw1[5]), "+v"(fw1[6]), "+v"(fw1[7]), \
              "+v"(fa1[0]), "+v"(fa1[1]), "+v"(fa1[2]), "+v"(fa1[3]) :: "memory")
#define PT_READS_DONE_0() asm volatile("s_waitcnt lgkmcnt(0)" \
            : "+v"(fw0[0]), "+v"(fw0[1]), "+v"(fw0[2]), "+v"(fw0[3]), "+v"(fw0[4]), "+v"(fw0[5]), "+v"(fw0[6]), "+v"(fw0[7]), \
              "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]) :: "memory")
#define PT_READS_DONE() do { if constexpr (HALF) PT_READS_DONE_1(); else PT_READS_DONE_ALL(); } while (0)
        if (stages && !grpB) {
            __builtin_amdgcn_sched_barrier(0);
            stage(std::false_type{}, read_pieces_a{}, buf ^ 1, sj, skt, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                // (HALF: the 12 reads of k-step 1 behind the first six pieces, two each)
                pt_static_for<(HALF ? 12 + 2 * i : 3 * i), (HALF ? (i < 6 ? 14 + 2 * i : 12 + 2 * i) : 3 * i + 3)>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < 8) pt_ds_read<k * 2048>(fw0[k], w0);
                    else if constexpr (k < 12) pt_ds_read<(k - 8) * 2048>(fa0[k - 8], a0);
                    else if constexpr (k < 20) pt_ds_read<(k - 12) * 2048>(fw1[k - 12], w1);
                    else pt_ds_read<(k - 20) * 2048>(fa1[k - 20], a1);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            PT_READS_DONE();
        } else if (stages) {
            // group B: the activation fragments first; its four W pieces (rows only group A reads, and read a phase ago) with half of the W
            // reads; then -- the activation reads have returned: lgkmcnt(8) -- its activation pieces, which overwrite rows just read
            __builtin_amdgcn_sched_barrier(0);
            pt_static_for<0, 4>([&](auto bc) __attribute__((always_inline)) {
                constexpr int b = decltype(bc)::value;
                if constexpr (!HALF) pt_ds_read<b * 2048>(fa0[b], a0);
                pt_ds_read<b * 2048>(fa1[b], a1);
            });
            stage(std::true_type{}, read_pieces_b{}, buf, sj, skt, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HALF) {                   // (4 activation + 8 W reads of k-step 1: behind slot 3 all are out, lgkmcnt(8) = the activation reads are back)
                    if constexpr (i < 4) { pt_ds_read<(2 * i) * 2048>(fw1[2 * i], w1); pt_ds_read<(2 * i + 1) * 2048>(fw1[2 * i + 1], w1); }
                } else if constexpr (i < 4) { pt_ds_read<(2 * i) * 2048>(fw0[2 * i], w0); pt_ds_read<(2 * i + 1) * 2048>(fw0[2 * i + 1], w0); }
                else { pt_ds_read<(2 * i - 8) * 2048>(fw1[2 * i - 8], w1); pt_ds_read<(2 * i - 7) * 2048>(fw1[2 * i - 7], w1); }
                if constexpr (i == 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            });
            PT_READS_DONE();
        } else if constexpr (HALF) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile(
                "ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:2048\n ds_read_b128 %2, %12 offset:4096\n ds_read_b128 %3, %12 offset:6144\n"
                "ds_read_b128 %4, %12 offset:8192\n ds_read_b128 %5, %12 offset:10240\n ds_read_b128 %6, %12 offset:12288\n ds_read_b128 %7, %12 offset:14336\n"
                "ds_read_b128 %8, %13\n ds_read_b128 %9, %13 offset:2048\n ds_read_b128 %10, %13 offset:4096\n ds_read_b128 %11, %13 offset:6144\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fw1[0]), "=&v"(fw1[1]), "=&v"(fw1[2]), "=&v"(fw1[3]), "=&v"(fw1[4]), "=&v"(fw1[5]), "=&v"(fw1[6]), "=&v"(fw1[7]),
                  "=&v"(fa1[0]), "=&v"(fa1[1]), "=&v"(fa1[2]), "=&v"(fa1[3])
                : "v"(w1), "v"(a1)
                : "memory");
        } else {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile(
                "ds_read_b128 %0, %24\n ds_read_b128 %1, %24 offset:2048\n ds_read_b128 %2, %24 offset:4096\n ds_read_b128 %3, %24 offset:6144\n"
                "ds_read_b128 %4, %24 offset:8192\n ds_read_b128 %5, %24 offset:10240\n ds_read_b128 %6, %24 offset:12288\n ds_read_b128 %7, %24 offset:14336\n"
                "ds_read_b128 %8, %25\n ds_read_b128 %9, %25 offset:2048\n ds_read_b128 %10, %25 offset:4096\n ds_read_b128 %11, %25 offset:6144\n"
                "ds_read_b128 %12, %26\n ds_read_b128 %13, %26 offset:2048\n ds_read_b128 %14, %26 offset:4096\n ds_read_b128 %15, %26 offset:6144\n"
                "ds_read_b128 %16, %26 offset:8192\n ds_read_b128 %17, %26 offset:10240\n ds_read_b128 %18, %26 offset:12288\n ds_read_b128 %19, %26 offset:14336\n"
                "ds_read_b128 %20, %27\n ds_read_b128 %21, %27 offset:2048\n ds_read_b128 %22, %27 offset:4096\n ds_read_b128 %23, %27 offset:6144\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fw0[0]), "=&v"(fw0[1]), "=&v"(fw0[2]), "=&v"(fw0[3]), "=&v"(fw0[4]), "=&v"(fw0[5]), "=&v"(fw0[6]), "=&v"(fw0[7]),
                  "=&v"(fa0[0]), "=&v"(fa0[1]), "=&v"(fa0[2]), "=&v"(fa0[3]),
                  "=&v"(fw1[0]), "=&v"(fw1[1]), "=&v"(fw1[2]), "=&v"(fw1[3]), "=&v"(fw1[4]), "=&v"(fw1[5]), "=&v"(fw1[6]), "=&v"(fw1[7]),
                  "=&v"(fa1[0]), "=&v"(fa1[1]), "=&v"(fa1[2]), "=&v"(fa1[3])
                : "v"(w0), "v"(a0), "v"(w1), "v"(a1)
                : "memory");
        }
#undef PT_READS_DONE
#undef PT_READS_DONE_ALL
#undef PT_READS_DONE_1
        if (LAST) PT_STAMP(1);
        PT_ISTAMP(1);
        // what must have LANDED before the barrier: the pieces issued before this phase (the W rows the other group reads next, group B's
        // activation rows) -- all but this phase's 8. A tile's first iteration: nothing is owed (both groups drained before the epilogue) and
        // the epilogue's stores are still in flight -- no wait
        if (stages) { if (!(skip_b1 || skip_wait)) __builtin_amdgcn_s_waitcnt(0x0F70 | (8 - WD)); }
        else if (!(skip_b1 || skip_wait)) __builtin_amdgcn_s_waitcnt(0x0F70);
        PT_ISTAMP(2);
        if (EPI == 2 && !RS && grpB && kt == nk - 3) touch_residual();
        __builtin_amdgcn_sched_barrier(0);
        if (!skip_b1) __builtin_amdgcn_s_barrier();    // (group A took a tile's first barrier ahead of its epilogue: see the tile loop)
        skip_b1 = false;
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) PT_STAMP(9);
        if (LAST) PT_STAMP(2);
        PT_ISTAMP(3);
#if ATLAS_TUNING
        if (!(diag & 128))
#endif
        if (b_first && (kt + 2 < nk || has_next)) stage(std::true_type{}, all_pieces{}, buf, sj, skt, nothing);      // once per tile: in front of the MFMAs
        if constexpr (VTR) {                           // activations as the MFMA A operand: C^T fragments, the same products in the same order
#pragma unroll
            for (int a = 0; a < FA; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b) acc[a][b] = T::mma(fa0[b], fw0[a], acc[a][b]);
#pragma unroll
            for (int a = 0; a < FA; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b) acc[a][b] = T::mma(fa1[b], fw1[a], acc[a][b]);
        } else {
#if ATLAS_TUNING
            if (!(diag & 64))                          // experiment (timing only): no MFMAs -- what the feed side (pieces, fragment reads, barriers) takes alone
#endif
            {
            if constexpr (RS && ATLAS_PT_RSPLIT == 2 && !LAST) {
                // variant 2: W fragment a of the next k-tile is read as soon as row a of k-step 0 has been issued (its register is free then),
                // the activation fragments between the two k-steps
                const uint32_t w0n = aw0 + (buf ^ 1) * STG, a0n = w0n + (uint32_t)(2 * STG + (wj * 64 - wi * 128) * 128);
                __builtin_amdgcn_sched_barrier(0);
                pt_static_for<0, FA>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(ac)::value;
#pragma unroll
                    for (int b = 0; b < FB; ++b) acc[a][b] = T::mma(fw0[a], fa0[b], acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                    pt_ds_read<a * 2048>(fw0[a], w0n);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (!grpB) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
                pt_static_for<0, FB>([&](auto bc) __attribute__((always_inline)) { pt_ds_read<decltype(bc)::value * 2048>(fa0[decltype(bc)::value], a0n); });
                __builtin_amdgcn_sched_barrier(0);
                mma_tile<T, FA, FB>(fw1, fa1, acc);
                PT_READS_DONE_0();
            } else if constexpr (RS) {
                mma_tile<T, FA, FB>(fw0, fa0, acc);
                if constexpr (LAST) {
                    mma_tile<T, FA, FB>(fw1, fa1, acc);
                } else {
                    // the next k-tile's k-step-0 fragments, from the OTHER stage, into the registers the 32 MFMAs above have released: one W
                    // read in front of each of the eight rows of MFMAs of k-step 1, the activation reads with rows 4..7 -- behind the wave's
                    // own wait (group A's four activation pieces of this k-tile's read phase were issued first)
                    const uint32_t w0n = aw0 + (buf ^ 1) * STG, a0n = w0n + (uint32_t)(2 * STG + (wj * 64 - wi * 128) * 128);
                    __builtin_amdgcn_sched_barrier(0);
                    pt_static_for<0, FA>([&](auto ac) __attribute__((always_inline)) {
                        constexpr int a = decltype(ac)::value;
                        pt_ds_read<a * 2048>(fw0[a], w0n);
                        if constexpr (a == 4) { if (!grpB) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); }
                        if constexpr (a >= 4) pt_ds_read<(a - 4) * 2048>(fa0[a - 4], a0n);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < FB; ++b) acc[a][b] = T::mma(fw1[a], fa1[b], acc[a][b]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    PT_READS_DONE_0();
                }
            } else if constexpr (WD == 0) {
                mma_tile<T, FA, FB>(fw0, fa0, acc);
                mma_tile<T, FA, FB>(fw1, fa1, acc);
            } else {
                // the 64 MFMAs in eight chunks of eight (two W fragments x four activation fragments; every accumulator still takes k-step 0
                // before k-step 1: same bits); deferred W piece c goes out in front of chunk ATLAS_PT_WSTRIDE * c. ONE block of MFMAs, no
                // branch (a second copy of the block for the iterations that stage nothing sent hipcc's register allocation into 600 B of
                // scratch): in those iterations -- the last of the workgroup's last tile, group B's first of every tile -- the pieces are
                // real loads of the same weight rows into the wave's own epilogue slot, which nothing reads before the vmcnt(0) in front of
                // the epilogue. Descriptor and LDS base are formed once, in front of the first MFMA.
                const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)tile_n0(sj) * K), 0, (int)(256u * K2), 0x00020000);
                unsigned char* const lwd = stages ? smem_raw + (grpB ? buf : buf ^ 1) * STG + (we * 32) * 128 : smem_raw + TR_OFF + wave * 4096;
                const uint32_t kbd = (uint32_t)skt * 128u;
                __builtin_amdgcn_sched_barrier(0);
                pt_static_for<0, 8>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c % ATLAS_PT_WSTRIDE == 0 && c / ATLAS_PT_WSTRIDE < WD) {
                        constexpr int i = 4 - WD + c / ATLAS_PT_WSTRIDE;
                        constexpr uint32_t rw_ = (uint32_t)(VTR ? 8 * i : 16 * (i & 1) + 4 * (i >> 1));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwd, (lds_ptr)(lwd + (i & 3) * 8 * 128), 16, (int)vw, (int)(rw_ * K2 + kbd), 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int a = 2 * (c & 3); a < 2 * (c & 3) + 2; ++a)
#pragma unroll
                        for (int b = 0; b < FB; ++b) acc[a][b] = (c < 4) ? T::mma(fw0[a], fa0[b], acc[a][b]) : T::mma(fw1[a], fa1[b], acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) PT_STAMP(10);
        if (LAST) PT_STAMP(3);
        PT_ISTAMP(5);
        if (LAST) {
            // what the epilogue adds is requested behind the tile's last MFMAs (the fragment registers are free now) and lands under the
            // wait / barrier that follows: the lane's bias values and, for EPI 2, its 16 residual pieces
            const int n0 = tile_n0(jc);
            const int fl = pt_fresh_lane(), lr = fl & 15, lg = fl >> 4;
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bias + n0), 0, 512, 0x00020000);
            if constexpr (EPI != 4) {                  // the lane's 8 bias values of every fragment pair
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (wi * 128 + 8 * lg) * 2, 64 * j, 0);
            } else {                                   // column 16 a + lr of the wave's 128: word (a & 1) * 2 of bq[a >> 1], low half
#pragma unroll
                for (int a = 0; a < 8; ++a) bq[a >> 1][(a & 1) * 2] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rb, (wi * 128 + a * 16 + lr) * 2, 0, 0);
                const int64_t mlast = M - 1;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {       // where the lane's two runs of 8 tokens live in V^T (pt_epilogue)
                    const int64_t tok0 = tile_m0(jc) + wj * 64 + 32 * jj + 8 * lg;
                    tki[2 * jj] = tokinfo[tok0 < mlast ? tok0 : mlast];
                    tki[2 * jj + 1] = tokinfo[tok0 + 7 < mlast ? tok0 + 7 : mlast];
                }
            }
            if constexpr (EPI == 2) {                  // the residual in the epilogue's transposed layout: (slice b, store s) = token 16 b + 4 s + (lane >> 4), chunk lane & 15
                const __amdgpu_buffer_rsrc_t rr = rows_rsrc(R, tile_m0(jc), N);
                // (one lane offset, the 16 pieces differ by scalar offsets: fewer address registers alive behind the MFMAs. The hardware's
                //  bounds check sees the lane offset only, so a piece whose row lies past M in the batch's LAST token tile is read for real:
                //  from the encoder workspace behind x -- R is always x, its first buffer, and the tile is 393 KB -- and is never stored)
                const int vo = (int)(((wj * 64 + lg) * N + n0 + wi * 128 + 8 * lr) * 2);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx)
                        rv[b * 4 + sidx] = __builtin_amdgcn_raw_buffer_load_b128(rr, vo, (b * 16 + 4 * sidx) * N * 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // A: its ACTIVATION pieces of the next k-tile (issued first; read in the next phase) have landed, the four W pieces may still fly.
        // (The last iteration of a tile: the epilogue's operands were requested behind the pieces and are not waited for HERE -- they land
        // while group B multiplies its last k-tile and are waited for in front of the epilogue: EPL loads on top of the four W pieces.)
        constexpr int EPL = EPI == 4 ? 12 : EPI == 2 ? 20 : 4;
        constexpr int VMA = LAST ? 4 + EPL : 4;     // (s_waitcnt simm16: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14)
        if constexpr (RS) {                        // both groups: ALL pieces of this iteration have landed (the other group reads some of them half a phase on)
            constexpr int VMR = LAST ? EPL : 0;
            if (stages || b_first) __builtin_amdgcn_s_waitcnt(0x0F70 | (VMR & 15) | ((VMR >> 4) << 14)); else __builtin_amdgcn_s_waitcnt(0x0F70);
        } else
        if (!grpB) { if (stages) __builtin_amdgcn_s_waitcnt(0x0F70 | (VMA & 15) | ((VMA >> 4) << 14)); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        if (EPI == 2 && (RS || !grpB) && kt == nk - 3) touch_residual();          // (RS: group B too touches BEHIND its wait -- it now waits for all its loads at the end of every multiply phase)
        if (LAST) PT_STAMP(4);
        PT_ISTAMP(6);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) PT_STAMP(11);
        if (LAST) PT_STAMP(5);
        PT_ISTAMP(7);
#if ATLAS_TUNING
        ++itc;
#endif
        buf ^= 1;
        skip_wait = false;
    };

    for (;;) {
        PT_STAMP(0);
#pragma unroll 1
        for (int kt = 0; kt < (RS ? 1 : nk - 1); ++kt) iteration(kt, std::false_type{}, std::true_type{});
        if constexpr (RS) {
#pragma unroll 1
            for (int kt = 1; kt < nk - 1; ++kt) iteration(kt, std::false_type{}, std::false_type{});
            iteration(nk - 1, std::true_type{}, std::false_type{});
        } else iteration(nk - 1, std::true_type{}, std::true_type{});
        // Epilogue of this tile, BOTH GROUPS SIDE BY SIDE: an epilogue is a chain of dependent VALU work (bias, GELU polynomial, packing)
        // that one wave per SIMD runs at ~7 cycles per instruction; two waves per SIMD hide each other's latencies. The barrier between a
        // group's reads and its MFMAs only keeps the two groups in opposite phases (no LDS hazard hangs on it: every buffer hand-over
        // goes through the OTHER barrier), so group A takes the next tile's first one here, ahead of its epilogue -- it meets group B
        // coming out of the tile's last MFMAs -- and then skips it in the next tile's first iteration (at the very end it is the
        // barrier group B's last iteration still owes). The stores stay in flight behind the wave's next wait. (Group A writing a first
        // slice in front of that barrier, while group B still multiplies, was measured: nothing on the full batch, -1.9 % on the ragged
        // one -- profiles/r03/enc_builds_split_epilogue.txt.)
        // (The wait: group B's pieces of the next tile's k-tile 1, the bias and the residual; group A waited in front of the tile's last
        // barrier -- for it this is a no-op that tells hipcc's wait insertion that nothing is in flight.)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        PT_STAMP(6);
        if (grpB) skip_wait = true;
        if (!grpB) { __builtin_amdgcn_s_barrier(); skip_b1 = true; }
        run_epilogue(jc);
        PT_STAMP(8);
#if ATLAS_TUNING
        ++tstamp;
#endif
        jc += nslots;
        if (jc >= njobs) break;
    }
#if ATLAS_TUNING
    if (dbgp != nullptr && blockIdx.x == 0 && tid == 0) { dbgp[1026] = wall_clock64(); dbgp[1027] = __builtin_readcyclecounter(); }
#endif
#undef PT_READS_DONE_0
#undef PT_STAMP
#undef PT_ISTAMP
}

// ------------------------------------------------------------------------------------------
// The GEMM for SMALL batches (query embedding: tens of tokens per query after packing, <= 4096 token slots): 64 x 64
// tiles so that a 768-wide GEMM of 1 300 tokens still makes ~250 workgroups, and a DEEP LDS-DMA pipeline. With tiles
// this small the MFMAs of a k-tile take 130 (16-bit) to 1 000 (fp32) cycles while an LDS-DMA piece needs ~2 000 cycles
// to land: one stage of prefetch (gemm_bt_kernel) pays that latency on every k-tile (FFN-2, 48 k-tiles: ~100 us of a
// 108 us layer). Here STAGES - 1 k-tiles are in flight behind counted `s_waitcnt vmcnt(n)`; one barrier per k-tile;
// fragment reads are inline asm (see gemm_pp_kernel). 3 stages x 16 KiB (fp16 / bf16) leave room for 3 workgroups per
// CU, which matters as much as the depth: 8 stages (one workgroup per CU) were slower than the single-stage kernel.
// Per forward of 64 queries x ~20 tokens (profiles/r01/query_gemm_configs.txt): 1.05 ms -> 0.93 ms (fp16), 3.83 -> 3.55 ms (fp32).
// ------------------------------------------------------------------------------------------
template <int N> static __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));     // vmcnt(N), expcnt / lgkmcnt untouched
}
template <int STAGES> static __device__ __forceinline__ void wait_tiles_in_flight(int younger) {   // 4 pieces per k-tile and wave
    if constexpr (STAGES >= 8) {
        if (younger >= 6) { wait_vmcnt<24>(); return; }
        if (younger == 5) { wait_vmcnt<20>(); return; }
        if (younger == 4) { wait_vmcnt<16>(); return; }
        if (younger == 3) { wait_vmcnt<12>(); return; }
    }
    if (younger >= 2) wait_vmcnt<8>();
    else if (younger == 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
}

template <class T, int EPI, int STAGES>
__global__ void __launch_bounds__(256)
gemm_ms_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp) {
    typedef typename T::elem E;
    constexpr int BCOL = 64, BTOK = 64, FA = 2, FB = 2;
    constexpr int EPC = 16 / (int)sizeof(E);
    constexpr uint32_t STG = 64 * 128;                                 // bytes per operand stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // W stages | A stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t M = cu[n];
    const int ncol = N / BCOL;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ctile = jj % ncol;
    const int64_t ttile = (int64_t)(jj / ncol) * 8 + xcd;
    if (ttile * BTOK >= M) return;
    const int n0 = ctile * BCOL;
    const int64_t m0 = ttile * BTOK;
    const int lr = lane & 15, lg = lane >> 4;
    const int nk = K / (8 * EPC);

    // this wave's 2 + 2 DMA pieces per k-tile (8 rows x 128 B each, source-side XOR swizzle)
    const int ch = (lane & 7) ^ (lane >> 3);
    const E* gw = W + (size_t)(n0 + wave * 16 + (lane >> 3)) * K + ch * EPC;
    const E* ga[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t ar = m0 + wave * 16 + i * 8 + (lane >> 3);
        if (ar >= M) ar = M - 1;
        ga[i] = A + (size_t)ar * K + ch * EPC;
    }
    auto stage = [&](const int kt) {
        const int buf = kt % STAGES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + (size_t)i * 8 * K + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + buf * STG + (wave * 16 + i * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + STAGES * STG + buf * STG + (wave * 16 + i * 8) * 128), 16, 0, 0);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const uint32_t aw0 = lds0 + (wi * 32 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t aw1 = lds0 + (wi * 32 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;
    const uint32_t aa0 = lds0 + STAGES * STG + (wj * 32 + lr) * 128 + ((0 + lg) ^ (lr & 7)) * 16;
    const uint32_t aa1 = lds0 + STAGES * STG + (wj * 32 + lr) * 128 + ((4 + lg) ^ (lr & 7)) * 16;

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < STAGES - 1 && kt < nk; ++kt) stage(kt);
    for (int kt = 0; kt < nk; ++kt) {
        // k-tile kt has landed when at most the pieces of the younger tiles already issued are outstanding
        int younger = nk - 1 - kt;
        if (younger > STAGES - 2) younger = STAGES - 2;
        wait_tiles_in_flight<STAGES>(younger);
        __builtin_amdgcn_s_barrier();                  // every wave's pieces of tile kt have landed; everybody is past tile kt-1
        if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1);    // into the buffer of tile kt-1
        const uint32_t bo = (uint32_t)(kt % STAGES) * STG;
        u4v fw0[FA], fa0[FB], fw1[FA], fa1[FB];
        {
            const uint32_t w0 = aw0 + bo, w1 = aw1 + bo, a0 = aa0 + bo, a1 = aa1 + bo;
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:2048\n ds_read_b128 %2, %9\n ds_read_b128 %3, %9 offset:2048\n"
                "ds_read_b128 %4, %10\n ds_read_b128 %5, %10 offset:2048\n ds_read_b128 %6, %11\n ds_read_b128 %7, %11 offset:2048\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fw0[0]), "=&v"(fw0[1]), "=&v"(fa0[0]), "=&v"(fa0[1]), "=&v"(fw1[0]), "=&v"(fw1[1]), "=&v"(fa1[0]), "=&v"(fa1[1])
                : "v"(w0), "v"(a0), "v"(w1), "v"(a1)
                : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<T, FA, FB>(fw0, fa0, acc);
        mma_tile<T, FA, FB>(fw1, fa1, acc);
    }
    gemm_epilogue<T, EPI, FA, FB>(acc, m0, n0, wi, wj, lr, lg, M, N, bias, R, C, VT, cu, tokinfo, Lp);
}

#if ATLAS_TUNING     // an experiment kept selectable in the tuning build only (slower than the ping-pong kernel: profiles/r02/gemm_wr_ab.txt)
// ------------------------------------------------------------------------------------------
// gemm_wr_kernel: the bulk GEMM with the WEIGHT operand loaded straight into registers and only the activations staged in LDS.
// What holds the 256 x 256 kernels above at ~1 190 TFLOP/s in the k-loop is feeding LDS: 64 KiB of LDS-DMA per k-tile, whose pieces
// take ~3 000 cycles to land with one k-tile in flight (profiles/r01/gemm_phases.txt), against 2 100 cycles of MFMA issue. Plain
// 16-byte loads to registers run at 64-87 B/clk/CU, the LDS-DMA path at 43-48. So: 8 waves side by side along the COLUMNS -- wave w owns
// columns [32 w, 32 w + 32) of the 256-column tile for all 256 tokens (2 x 16 fragments, the same 128 accumulator registers). A weight
// row is then needed by exactly one wave: its MFMA A fragments (row l & 15, 16 bytes at k-group l >> 4) are 16-byte loads from the
// row-major weight matrix, no LDS, no redundancy, prefetched two k-tiles ahead in registers. Only the activation tile (32 KiB per
// k-tile, half the LDS-DMA bytes and half the pieces to issue) goes through LDS, in FOUR stages (three k-tiles in flight behind
// counted vmcnt waits), one barrier per k-tile. Same k order per element as every other configuration: identical bits.
// ------------------------------------------------------------------------------------------
template <class T, int EPI>
__global__ void __launch_bounds__(512)
gemm_wr_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W, const typename T::elem* __restrict__ bias,
               const typename T::elem* __restrict__ R, typename T::elem* __restrict__ C, typename T::elem* __restrict__ VT,
               const int* __restrict__ cu, int n, const int2* __restrict__ tokinfo, int N, int K, int Lp, int diag) {
    typedef typename T::elem E;
    static_assert(sizeof(E) == 2, "16-bit dtypes only");
    constexpr int BCOL = 256, BTOK = 256, FA = 2, FB = 16, ST = 4;
    constexpr int EPC = 8;
    constexpr uint32_t STG = 256 * 128;                              // bytes per activation stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // ST activation stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t M = cu[n];
    const int ncol = N / BCOL;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ctile = jj % ncol;
    const int64_t ttile = (int64_t)(jj / ncol) * 8 + xcd;
    if (ttile * BTOK >= M) return;
    const int n0 = ctile * BCOL;
    const int64_t m0 = ttile * BTOK;
    const int lr = lane & 15, lg = lane >> 4;
    const int nk = K / (8 * EPC);

    // activations: this wave's 4 DMA pieces per k-tile (8 rows x 128 B each, source-side XOR swizzle as in gemm_bt_kernel)
    const int ch = (lane & 7) ^ (lane >> 3);
    const E* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t ar = m0 + wave * 32 + i * 8 + (lane >> 3);
        if (ar >= M) ar = M - 1;                                   // clamped: tail rows are never stored
        ga[i] = A + (size_t)ar * K + ch * EPC;
    }
    auto stage = [&](const int kt) {
        const uint32_t so = (uint32_t)(kt % ST) * STG;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * (8 * EPC)),
                                             (__attribute__((address_space(3))) void*)(smem_raw + so + (wave * 32 + i * 8) * 128), 16, 0, 0);
    };
    // weights: lane's 16 bytes of fragment a (rows n0 + 32 wave + 16 a + lr), k-step ks of k-tile kt
    const E* gw = W + (size_t)(n0 + wave * 32 + lr) * K + lg * EPC;
    auto load_w = [&](u4v (&w)[2][2], const int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) w[ks][a] = *(const u4v*)(gw + (size_t)a * 16 * K + kt * (8 * EPC) + ks * 4 * EPC);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const uint32_t aa0 = lds0 + lr * 128 + ((0 + lg) ^ (lr & 7)) * 16;          // fragment b adds b * 2048 (rows 16 apart keep row & 7)
    const uint32_t aa1 = lds0 + lr * 128 + ((4 + lg) ^ (lr & 7)) * 16;

    f4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // prologue: weights of k-tiles 0 and 1 and activation stages 0, 1, 2 in flight (issue order = the order the waits below count on)
    u4v w0[2][2], w1[2][2], w2[2][2];
    load_w(w0, 0);
    stage(0);
    if (nk > 1) load_w(w1, 1); else load_w(w1, 0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    // steady state at the top of iteration kt, oldest -> newest: W(kt) A(kt) W(kt+1) A(kt+1) A(kt+2); the iteration issues W(kt+2), A(kt+3)
    auto body = [&](const int kt, u4v (&wc)[2][2], u4v (&wn2)[2][2]) {
        if (kt + 2 < nk) load_w(wn2, kt + 2);
        // W(kt) and every wave's A(kt) have landed: at most the 4 + 4 + 4 (+ 4) younger operations of this wave may still be in flight
        if (kt + 2 < nk) wait_vmcnt<16>(); else if (kt + 1 < nk) wait_vmcnt<8>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // ... everybody's pieces of A(kt); and everybody is past its reads of stage (kt - 1) % ST
        if (kt + 3 < nk) stage(kt + 3);                // into the stage k-tile kt - 1 occupied
        const uint32_t so = (uint32_t)(kt % ST) * STG;
        // 4 token fragments (8 ds_read_b128) per group; the reads of group g + 1 are in flight under the 16 MFMAs of group g (two
        // register sets; the waits are counted by hand: the compiler does not see inline-asm LDS reads)
        u4v fx[8], fy[8];
        auto issue = [&](u4v (&f)[8], const int b) {
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:2048\n ds_read_b128 %2, %8 offset:4096\n ds_read_b128 %3, %8 offset:6144\n"
                "ds_read_b128 %4, %9\n ds_read_b128 %5, %9 offset:2048\n ds_read_b128 %6, %9 offset:4096\n ds_read_b128 %7, %9 offset:6144"
                : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
                : "v"(aa0 + so + b * 2048), "v"(aa1 + so + b * 2048)
                : "memory");
        };
        auto landed = [&](u4v (&f)[8], const bool younger_in_flight) {       // the 8 reads into f have completed (8 younger ones may still fly)
            if (younger_in_flight)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory");
        };
        auto multiply = [&](const u4v (&f)[8], const int b) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int a = 0; a < FA; ++a) {
                    acc[a][b + bb] = T::mma(wc[0][a], f[bb], acc[a][b + bb]);
                    acc[a][b + bb] = T::mma(wc[1][a], f[4 + bb], acc[a][b + bb]);
                }
        };
        issue(fx, 0);
        issue(fy, 4);  landed(fx, true);  multiply(fx, 0);
        issue(fx, 8);  landed(fy, true);  multiply(fy, 4);
        issue(fy, 12); landed(fx, true);  multiply(fx, 8);
        landed(fy, false); multiply(fy, 12);
    };
    for (int kt = 0; kt < nk; kt += 3) {               // the three weight register sets rotate by name: no copies
        body(kt, w0, w2);
        if (kt + 1 < nk) body(kt + 1, w1, w0);
        if (kt + 2 < nk) body(kt + 2, w2, w1);
    }
    if (diag & 1) { if (acc[0][0][0] == 12345.678f) C[0] = 0; return; }
    gemm_epilogue<T, EPI, FA, FB>(acc, m0, n0, wave, 0, lr, lg, M, N, bias, R, C, VT, cu, tokinfo, Lp);
}
#endif

// GEMM configurations. The encoder picks by worst-case token slots n * L: > 16384 -> 4, > 4096 -> 0, else 3;
// the tuning build's atlas_tune_set_gemm_cfg(n) forces one (tuning and the bit-equality test: every configuration gives the same bits).
//   9  gemm_pt_kernel  256 x 256, PERSISTENT ping-pong, register epilogue    (index refresh, 16-bit dtypes: every GEMM)
//   4  gemm_pp_kernel  256 x 256, ping-pong schedule, LDS epilogue          (fp32 bulk; A/B reference for 9; FFN-1 of the 16-bit dtypes goes to 6)
//   6  gemm_co_kernel  256 x 128, two co-resident workgroups per CU         (16-bit dtypes; every GEMM when forced)
//   7  gemm_pp_kernel  for every GEMM                                        (A/B reference for the 4 / 6 split)
//   2  gemm_bt_kernel  256 x 256, single phase                              (A/B reference for 4)
//   0  gemm_bt_kernel  128 x 128
//   3  gemm_ms_kernel  64 x 64, 3 (16-bit) / 4 (fp32) LDS-DMA stages        (query batches)
//   5  gemm_bt_kernel  64 x 64, single stage                                (A/B reference for 3)
#if ATLAS_TUNING
unsigned long long* g_gemm_dbg = nullptr;    // atlas_tune_set_gemm_stamps
int g_pt_stamp_nth = 0, g_pt_launches = 0;   // atlas_tune_set_gemm_stamps_nth: only the nth gemm_pt launch from now on gets the stamp buffer
int g_gemm_diag = 0;                         // atlas_tune_set_gemm_diag
int g_gemm_cfg = -1;                         // atlas_tune_set_gemm_cfg: -1 = by size (what the product library always does)
int g_att_pf = 0;                            // atlas_tune_set_att_pf: 0 = attention_kernel<.., VROW> (one workgroup per item, no prefetch); 2 / 3 = attention_pf_kernel with that many workgroups per CU
int g_att_xmap = 1;                          // atlas_tune_set_att_xmap: 0 = item = blockIdx (rounds 1-5)
int g_skip_ln = 0;                           // atlas_tune_set_skip_ln: 1 = the two ln_kernel launches of a layer are left out (RESULTS WRONG: the bound of any LayerNorm fusion)
#else
constexpr unsigned long long* g_gemm_dbg = nullptr;
constexpr int g_gemm_diag = 0;
constexpr int g_gemm_cfg = -1;
constexpr int g_att_pf = 0;
constexpr int g_skip_ln = 0;
constexpr int g_att_xmap = 1;
#endif

static int encoder_device_cus() {     // CU count of the current device, asked every time (an attribute read; no cached state)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) return 256;
    return cus;
}

template <class T, int EPI>
static void launch_gemm(int cfg, hipStream_t stream, const typename T::elem* A, const typename T::elem* W, const typename T::elem* bias,
                        const typename T::elem* R, typename T::elem* C, typename T::elem* VT, int64_t Mmax, const int* cu, int n,
                        const int2* tokinfo, int N, int K, int Lp) {
    auto go = [&](auto kern, int bcol, int btok, int nthreads) {
        const size_t lds = (size_t)(bcol + btok) * 128 * 2;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const unsigned mtiles = (unsigned)((Mmax + btok - 1) / btok);
        hipLaunchKernelGGL(kern, dim3((mtiles + 7) / 8 * 8 * (N / bcol)), dim3(nthreads), lds, stream, A, W, bias, R, C, VT,
                           cu, n, tokinfo, N, K, Lp);
    };
    if ((cfg == 9 || cfg == 10) && sizeof(typename T::elem) != 2) cfg = 4;   // the persistent kernel serves the 16-bit dtypes
    if (cfg == 9 || cfg == 10) {              // 10 (tuning build only): rounds 3-4's two-launch QKV with the V^T epilogue, the A/B reference of 9
        if constexpr (sizeof(typename T::elem) == 2) {
            // one workgroup per CU, a multiple of 8 so that workgroup b's tiles are those of XCD b % 8; workgroups without a tile exit
            const unsigned grid = (unsigned)(encoder_device_cus() / 8 * 8);
            auto go_pt = [&](auto kern, const typename T::elem* Wp, const typename T::elem* bp, int Np) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                unsigned long long* dbg = g_gemm_dbg;
#if ATLAS_TUNING
                if (g_pt_stamp_nth > 0 && ++g_pt_launches != g_pt_stamp_nth) dbg = nullptr;
#endif
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, stream, A, Wp, bp, R, C, VT, cu, n, tokinfo, Np, K, Lp, g_gemm_diag, dbg);
            };
            if constexpr (EPI == 3) {
                // QKV projection: ONE launch, all 2304 columns row-major into [M, 2304] (round 5: attention_kernel<.., VROW> takes V from there;
                // rounds 3-4: q | k -> [M, 1536], then a second launch with the V^T epilogue, EPI 4 -- 31 us per V tile against 25.5 for a
                // q | k tile, and one more kernel fill + drain per layer)
#if ATLAS_TUNING
                if (cfg == 10) {
                    go_pt(gemm_pt_kernel<T, 3>, W, bias, 2 * HID);
                    go_pt(gemm_pt_kernel<T, 4>, W + (size_t)2 * HID * K, bias + 2 * HID, HID);
                } else
#endif
                go_pt(gemm_pt_kernel<T, 3>, W, bias, 3 * HID);
            } else {
                go_pt(gemm_pt_kernel<T, EPI>, W, bias, N);
            }
        }
        return;
    }
    // The PRODUCT library reaches cfg 9 (16-bit bulk), 4 (fp32 bulk, from 9), 0 and 3 only: everything else -- and the 16-bit instantiations of
    // gemm_pp_kernel -- exists in the tuning build alone (A/B references, the bit-equality test), so that libatlas_hip.so carries no kernel
    // its dispatch cannot launch
    constexpr bool TUNE = ATLAS_TUNING != 0;
    constexpr bool IS16 = sizeof(typename T::elem) == 2;
    if (cfg == 4 && EPI == 1 && IS16) cfg = 6;                              // FFN-1: the two-workgroup kernel is 4 % faster there
    if (cfg == 7) cfg = 4;                                                  // 7 = gemm_pp_kernel for every GEMM (A/B)
    if (!TUNE && (cfg == 2 || cfg == 5 || cfg == 6 || cfg == 8 || (cfg == 4 && IS16))) cfg = 0;      // (unreachable: g_gemm_cfg is -1 there)
    if (cfg == 4) {
        if constexpr (TUNE || !IS16) {
            (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const unsigned mtiles = (unsigned)((Mmax + 255) / 256);
            hipLaunchKernelGGL((gemm_pp_kernel<T, EPI>), dim3((mtiles + 7) / 8 * 8 * (N / 256)), dim3(512), 128 * 1024 + 512, stream, A, W, bias, R, C,
                               VT, cu, n, tokinfo, N, K, Lp, g_gemm_dbg, g_gemm_diag);
        }
    }
#if ATLAS_TUNING
    else if (cfg == 6) {
        if constexpr (sizeof(typename T::elem) == 2) {
            (void)hipFuncSetAttribute((const void*)gemm_co_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const unsigned mtiles = (unsigned)((Mmax + 127) / 128);
            hipLaunchKernelGGL((gemm_co_kernel<T, EPI>), dim3((mtiles + 7) / 8 * 8 * (N / 256)), dim3(256), 64 * 1024 + 512, stream, A, W, bias, R, C,
                               VT, cu, n, tokinfo, N, K, Lp, g_gemm_diag);
        } else go(gemm_bt_kernel<T, EPI, 256, 256, 2, 4>, 256, 256, 512);
    }
    else if (cfg == 8) {                               // weights in registers, activations through four LDS stages
        if constexpr (sizeof(typename T::elem) == 2) {
            (void)hipFuncSetAttribute((const void*)gemm_wr_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const unsigned mtiles = (unsigned)((Mmax + 255) / 256);
            hipLaunchKernelGGL((gemm_wr_kernel<T, EPI>), dim3((mtiles + 7) / 8 * 8 * (N / 256)), dim3(512), 4 * 256 * 128, stream, A, W, bias, R, C,
                               VT, cu, n, tokinfo, N, K, Lp, g_gemm_diag);
        } else go(gemm_bt_kernel<T, EPI, 256, 256, 2, 4>, 256, 256, 512);
    }
    else if (cfg == 2) go(gemm_bt_kernel<T, EPI, 256, 256, 2, 4>, 256, 256, 512);
    else if (cfg == 5) go(gemm_bt_kernel<T, EPI, 64, 64, 2, 2>, 64, 64, 256);
#endif
    else if (cfg == 3) {
        constexpr int ST = (sizeof(typename T::elem) == 2) ? 3 : 4;   // measured: 3 x 16 KiB (3 workgroups / CU) best for 16-bit
        (void)hipFuncSetAttribute((const void*)gemm_ms_kernel<T, EPI, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const unsigned mtiles = (unsigned)((Mmax + 63) / 64);
        hipLaunchKernelGGL((gemm_ms_kernel<T, EPI, ST>), dim3((mtiles + 7) / 8 * 8 * (N / 64)), dim3(256), (size_t)ST * 2 * 64 * 128, stream, A, W,
                           bias, R, C, VT, cu, n, tokinfo, N, K, Lp);
    }
    else go(gemm_bt_kernel<T, EPI, 128, 128, 2, 2>, 128, 128, 256);
}

// Reductions over the four 16-lane rows of a wave (lane ^ 16, lane ^ 32) with gfx950's row swaps: `v_permlane16_swap` exchanges the odd rows of
// its first operand with the even rows of its second, `v_permlane32_swap` the upper half of the first with the lower half of the second -- with
// both operands = x the two results hold, on every lane, the two values of the pair, so one VALU op finishes the step. __shfl_xor compiles to
// ds_bpermute_b32 for these distances: an LDS round trip of ~120 cycles, and a softmax row needs FOUR of them in one dependent chain (max over
// rows, then the sum) per query fragment. Same operands, commutative operation: bit-identical results.
// Inline asm, not __builtin_amdgcn_permlane16/32_swap: with this hipcc (ROCm 7.2.0) the builtin's SECOND result reads the register of the first
// (r[0] + r[1] compiles to `v_add v1, v1, v1` whatever the operands are: tools/permlane_probe.hip; every encoder test caught it). The two
// v_nop are the wait states the swap needs behind a VALU write of either operand (LLVM's gfx950 hazard rule; hipcc does not look inside asm).
static __device__ __forceinline__ void rows_swap16(unsigned& a, unsigned& b) { asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
static __device__ __forceinline__ void rows_swap32(unsigned& a, unsigned& b) { asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
static __device__ __forceinline__ float rows4_max(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    rows_swap16(a, b);
    x = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    a = b = __builtin_bit_cast(unsigned, x);
    rows_swap32(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
static __device__ __forceinline__ float rows4_sum(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    rows_swap16(a, b);
    x = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    a = b = __builtin_bit_cast(unsigned, x);
    rows_swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
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
// VROW (round 5, the 16-bit bulk path): the QKV projection is ONE GEMM launch that leaves q | k | v row-major in `qk` ([M][2304]; `vt` unused). V is
// staged like K -- coalesced 16-byte chunks, rows of 64 dims at a 144-byte pitch -- and the P.V B operand (4 consecutive KEYS of one dim per
// lane) comes out of that row-major tile through gfx950's transposing LDS read: `ds_read_b64_tr_b16` with lane s of a 16-lane group pointing at
// key 4 g + (s >> 2), dims 16 df + 4 (s & 3) .. + 3 returns V[4 g + j][16 df + (lane & 15)], j = 0..3, to lane (lane & 15, g) -- checked element
// by element on the hardware (tools/tr_probe.hip, profiles/r05/tr_probe.txt). The same values in the same registers as the V^T path: results
// are bit-identical (tools/lib_ab.py compares the builds' outputs). What it buys is on the GEMM side: no V^T epilogue, one launch less per layer.
template <class T, int MAXKF, bool VROW = false>
__global__ void __launch_bounds__(256)
attention_kernel(const uint16_t* __restrict__ qk, const uint16_t* __restrict__ vt, const int* __restrict__ cu, int LpMax,
                 uint16_t* __restrict__ ctx, const int n_passages, const int xmap) {
    // K (this head's [Lp][64] slice) and V^T ([64][Lp]) are staged in LDS ONCE per (passage, head) with coalesced
    // loads; the query fragments then run entirely out of LDS + registers (re-reading K/V^T from L2 for each of the
    // L/16 query fragments made the kernel latency-bound: 300 us -> see profiles). K chunks are XOR-swizzled by
    // (key & 7) for the ds_read_b128 fragment reads; V^T rows are padded by 16 B so dims spread over the banks.
    // L = this passage's packed length (all its keys are real), Lp = L rounded up to 32; keys in [L, Lp) are zero
    // rows / zero V^T columns with an additive mask of -inf.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    // (passage, head) of this workgroup. Round 6: workgroup g runs on XCD g % 8 (the dispatcher deals workgroups round-robin), and with
    // item = blockIdx the 12 heads of a passage -- 12 x 128 B of every 4 608-byte q | k | v row -- were fetched by 12 workgroups on 8 different
    // XCDs at different times: scattered 128-byte requests, 4.2 TB/s. XMAP: XCD x takes the passages b = 8 j + x, its slots walk them head by
    // head, so the 12 pieces of a row are requested from ONE XCD within a few microseconds of each other (tuning: atlas_tune_set_att_xmap)
    int b, h;
    if (xmap) { const int slot = blockIdx.x >> 3; b = (slot / NHEAD) * 8 + (blockIdx.x & 7); h = slot % NHEAD; if (b >= n_passages) return; }
    else { b = blockIdx.x / NHEAD; h = blockIdx.x % NHEAD; if (b >= n_passages) return; }
    const int tb = cu[b], L = cu[b + 1] - tb;
    if (L <= 0) return;
    const int Lp = (L + 31) & ~31;
    uint4* sK = (uint4*)smem;                                  // [Lp][8]
    const int vstride = Lp + 8;                                // halfs
    uint16_t* sVt = (uint16_t*)(sK + (size_t)Lp * 8);          // [64][Lp + 8]   (VROW: [Lp][72]: the same bytes at Lp = 64, 9 / 8.5 of them above)
    constexpr int VPITCH = 160;                                // VROW: bytes per key row of the V tile (round 6: 160, not 144 -- 36 r mod 64 put rows 0 and 7 of a 32-lane group of the tr read on four common banks, 27 % of the kernel's LDS cycles; 40 r mod 64 covers the 64 banks exactly once)
    float* sMask = VROW ? (float*)((unsigned char*)sVt + (size_t)Lp * VPITCH) : (float*)(sVt + 64 * vstride);      // [Lp] 0 for keys < L, -inf beyond
    constexpr int QLD = VROW ? 3 * HID : 2 * HID;              // row pitch of `qk`
    const uint16_t* Qb = qk + (size_t)tb * QLD + h * DHEAD;
    const uint16_t* Kb = Qb + HID;
    const uint16_t* Vb = Qb + 2 * HID;                         // VROW: this head's V rows, [key][64] at pitch QLD
    const uint16_t* Vt = VROW ? nullptr : vt + ((size_t)b * HID + h * DHEAD) * LpMax + (tb & 7);     // the passage's keys start tb & 7 columns into its rows (layout note)
    // the first query fragment of this wave is requested before K / V^T are staged (its latency runs under the staging), the
    // next one before the current one is computed
    auto load_q = [&](const int qf, uint4& qa, uint4& qb) {
        int qrow = qf * 16 + lr; if (qrow >= L) qrow = L - 1;
        qa = *(const uint4*)(Qb + (size_t)qrow * QLD + lg * 8);
        qb = *(const uint4*)(Qb + (size_t)qrow * QLD + 32 + lg * 8);
    };
    uint4 q0n = make_uint4(0, 0, 0, 0), q1n = q0n;
    if (wave * 16 < L) load_q(wave, q0n, q1n);
    for (int j = tid; j < Lp; j += 256) sMask[j] = (j < L) ? 0.0f : -__builtin_inff();
    // staging: four 16-byte chunks of K and four of V^T per thread are requested before the first is written to LDS (one chunk per
    // loop iteration was eight serialised memory latencies per workgroup at L = 128)
    const int cpr = Lp / 8;                                    // 16-B chunks per V^T row
#pragma unroll
    for (int base = 0; base < MAXKF * 16 * 8; base += 4 * 256) {     // compile-time trip count (Lp <= MAXKF * 16): no loop-entry drain
        if (base >= Lp * 8) break;
        uint4 kv[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * 256 + tid;
            const int key = idx >> 3, ch = idx & 7;
            kv[i] = *(const uint4*)(Kb + (size_t)(key < L ? key : L - 1) * QLD + ch * 8);   // unconditional (clamped): no branch, all in flight
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * 256 + tid;              // V^T has 64 * cpr = Lp * 8 chunks as well
            if constexpr (VROW) {                              // the same (key, chunk) walk as K
                const int key = idx >> 3, ch = idx & 7;
                vv[i] = *(const uint4*)(Vb + (size_t)(key < L ? key : L - 1) * QLD + ch * 8);
            } else {
                const int idc = idx < Lp * 8 ? idx : Lp * 8 - 1;
                const int dim = idc / cpr, c = idc - dim * cpr;
                vv[i] = *(const uint4_a2*)(Vt + (size_t)dim * LpMax + c * 8);     // 2-byte aligned for ragged batches: one global_load_dwordx4 all the same
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * 256 + tid;
            const int key = idx >> 3, ch = idx & 7;
            if (idx < Lp * 8) sK[key * 8 + (ch ^ (key & 7))] = (key < L) ? kv[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * 256 + tid;
            if (idx >= Lp * 8) continue;
            if constexpr (VROW) {                              // key rows of 144 B; rows >= L (another passage's tokens) are zero
                const int key = idx >> 3, ch = idx & 7;
                *(uint4*)((unsigned char*)sVt + key * VPITCH + ch * 16) = (key < L) ? vv[i] : make_uint4(0, 0, 0, 0);
                continue;
            }
            const int dim = idx / cpr, c = idx - dim * cpr;
            uint4 v = vv[i];
            const int left = L - c * 8;                        // columns >= L were never written: force them to 0
            if (left < 8) {
                uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e >= left) wv[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            *(uint4*)(sVt + dim * vstride + c * 8) = v;
        }
    }
    __syncthreads();
    const int nkf = Lp / 16;
    // One query fragment (16 queries x all keys), compiled for NKF key fragments: the passage's own padded length picks the
    // variant (Lp / 16 = 2, 4, ... MAXKF), so ragged batches pay for their own keys and not for run-time guards up to the
    // batch's longest passage. FULL: all keys are real (L = Lp): no mask. For MAXKF = 32 (passages > 256 tokens) there is one
    // guarded variant instead of sixteen. Every variant does the same arithmetic per element in the same order, so a passage
    // gives the same bits whichever one serves it.
    auto fragment = [&](const int qf, const uint4 q0, const uint4 q1, auto nkf_tag, auto full_tag) {
        constexpr int NKF = decltype(nkf_tag)::value;
        asm volatile("" ::: "memory");      // keeps hipcc from hoisting one variant's LDS reads over the switch (154 -> 108 VGPRs: four workgroups per CU again)
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool GUARD = (MAXKF > 16);                    // NKF = MAXKF with run-time guards on nkf
        f4 s[NKF];
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
            s[kf] = (f4){0.f, 0.f, 0.f, 0.f};
            if (!GUARD || FULL || kf < nkf) {
                const int krow = kf * 16 + lr;
                const uint4 k0 = sK[krow * 8 + (lg ^ (krow & 7))];
                const uint4 k1 = sK[krow * 8 + ((4 + lg) ^ (krow & 7))];
                s[kf] = T::mma(k0, q0, s[kf]);
                s[kf] = T::mma(k1, q1, s[kf]);
            }
        }
        uint32_t pk[NKF][2];                                      // softmax(...).type_as(model dtype), two keys per word
        if constexpr (T::DT == ATLAS_DT_F16) {
            // fp16 model: the three model-dtype steps (dt(q.k), / sqrt(64) -- exact --, + mask) run on PACKED halves
            // (v_cvt_pk_f16_f32, v_pk_mul_f16, v_pk_add_f16, v_pk_max_f16); the softmax itself is fp32 as in the reference
            // (modeling_bert.py:352): e = 2^(v * log2 e - max * log2 e), one mixed-precision fma + v_exp_f32 per key
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            h2 v[NKF][2];
            h2 mx2 = {(_Float16)(-__builtin_inff()), (_Float16)(-__builtin_inff())};
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        h2 x = {(_Float16)s[kf][2 * e], (_Float16)s[kf][2 * e + 1]};
                        x = x * (h2){(_Float16)0.125f, (_Float16)0.125f};
                        if (!FULL) {
                            const float2 am = *(const float2*)(sMask + kf * 16 + lg * 4 + 2 * e);
                            x = x + (h2){(_Float16)am.x, (_Float16)am.y};
                        }
                        v[kf][e] = x;
                        mx2 = __builtin_elementwise_max(mx2, x);
                    }
                }
            float mx = fmaxf((float)mx2.x, (float)mx2.y);
            mx = rows4_max(mx);
            const float nmx = -mx * 1.4426950408889634f;
            f2 ex[NKF][2];
            f2 sum2 = {0.f, 0.f};
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f2 ee = {__builtin_amdgcn_exp2f(__builtin_fmaf((float)v[kf][e].x, 1.4426950408889634f, nmx)),
                                       __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[kf][e].y, 1.4426950408889634f, nmx))};   // 2^-inf = 0 for padded keys
                        ex[kf][e] = ee;
                        sum2 += ee;
                    }
                }
            float sum = sum2.x + sum2.y;
            sum = rows4_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f2 pe = ex[kf][e] * (f2){inv, inv};
                        const h2 ph = {(_Float16)pe.x, (_Float16)pe.y};
                        pk[kf][e] = __builtin_bit_cast(uint32_t, ph);
                    }
                }
        } else {
            float mx = -__builtin_inff();
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
                    const float4 am = *(const float4*)(sMask + kf * 16 + lg * 4);
                    const float a4[4] = {am.x, am.y, am.z, am.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // dt(q.k) / sqrt(64) is exact in fp16/bf16; + mask is an add in the model dtype (modeling_bert.py:346-349)
                        const float v = T::rnd(T::rnd(s[kf][r]) * 0.125f + a4[r]);
                        s[kf][r] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            mx = rows4_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(s[kf][r] - mx);            // exp(-inf) = 0 for padded keys
                        s[kf][r] = e;
                        sum += e;
                    }
                }
            sum = rows4_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf)
                if (!GUARD || FULL || kf < nkf) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        pk[kf][e] = (uint32_t)T::st(s[kf][2 * e] * inv) | ((uint32_t)T::st(s[kf][2 * e + 1] * inv) << 16);
                }
        }
        // ctx = P V, P straight from the registers above
        f4 o[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] = (f4){0.f, 0.f, 0.f, 0.f};
        if constexpr (VROW) {
            // keys 32 ks + 4 lg .. + 3 (first read of a pair) and + 16 (second) of dim 16 df + lr, out of the row-major tile: the lane's address selects
            // key 4 lg + (lr >> 2) and dims 4 (lr & 3) .. + 3, the instruction transposes inside the 16-lane group (header note).
            // Round 6: the eight reads of key step ks + 1 are issued BEFORE the MFMAs of step ks and waited for with a counted lgkmcnt (LDS
            // operations return in order): rounds 1-5 issued a step's reads, drained lgkmcnt and only then multiplied -- an LDS latency of
            // ~120 cycles exposed NKF / 2 times per query fragment. The wait statements name the registers they release ("+v"), so the
            // MFMAs that read them cannot be scheduled above them.
            const uint32_t va0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sVt + (uint32_t)((4 * lg + (lr >> 2)) * VPITCH + 8 * (lr & 3));
            unsigned long long tq[2][8];
            auto issue = [&](const int ks, unsigned long long (&t)[8]) {
                const uint32_t va = va0 + (uint32_t)(32 * ks * VPITCH);
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:2560\n\t"
                             "ds_read_b64_tr_b16 %2, %8 offset:32\n\tds_read_b64_tr_b16 %3, %8 offset:2592\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:64\n\tds_read_b64_tr_b16 %5, %8 offset:2624\n\t"
                             "ds_read_b64_tr_b16 %6, %8 offset:96\n\tds_read_b64_tr_b16 %7, %8 offset:2656"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]) : "v"(va) : "memory");
            };
            issue(0, tq[0]);
#pragma unroll
            for (int ks = 0; ks < NKF / 2; ++ks)
                if (!GUARD || FULL || 2 * ks < nkf) {
                    unsigned long long (&t)[8] = tq[ks & 1];
                    const bool more = (ks + 1 < NKF / 2) && (!GUARD || FULL || 2 * (ks + 1) < nkf);
                    if (more) {
                        issue(ks + 1, tq[(ks + 1) & 1]);
                        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]) :: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]) :: "memory");
                    }
                    const uint4 pa = make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]);
#pragma unroll
                    for (int df = 0; df < 4; ++df)
                        // (round 6: operands swapped -- ctx^T = V^T . P^T, the same products over the same keys -- so that o[df][r] = ctx[query 16 qf + lr][dim 16 df + 4 lg + r])
                        o[df] = T::mma(make_uint4((uint32_t)t[2 * df], (uint32_t)(t[2 * df] >> 32), (uint32_t)t[2 * df + 1], (uint32_t)(t[2 * df + 1] >> 32)), pa, o[df]);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < NKF / 2; ++ks)
                if (!GUARD || FULL || 2 * ks < nkf) {
                    const uint4 pa = make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]);
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        const uint16_t* vrow = sVt + (df * 16 + lr) * vstride + ks * 32 + lg * 4;
                        const uint2 v0 = *(const uint2*)vrow, v1 = *(const uint2*)(vrow + 16);
                        o[df] = T::mma(pa, make_uint4(v0.x, v0.y, v1.x, v1.y), o[df]);
                    }
                }
        }
        // context_layer.permute(0,2,1,3).view(.., 768): [token][h*64 + dim]
        if constexpr (VROW) {                                      // four consecutive dims of one query row per lane: 4 stores of 8 bytes
            const int row = qf * 16 + lr;
            if (FULL || row < L) {
                uint16_t* dst = ctx + ((size_t)tb + row) * HID + h * DHEAD + 4 * lg;
#pragma unroll
                for (int df = 0; df < 4; ++df)
                    *(uint2*)(dst + df * 16) = make_uint2((uint32_t)T::st(o[df][0]) | ((uint32_t)T::st(o[df][1]) << 16),
                                                          (uint32_t)T::st(o[df][2]) | ((uint32_t)T::st(o[df][3]) << 16));
            }
        } else {
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qf * 16 + lg * 4 + r;
                if (FULL || row < L) ctx[((size_t)tb + row) * HID + h * DHEAD + df * 16 + lr] = T::st(o[df][r]);
            }
        }
    };
    const bool full = (L == Lp);                                 // workgroup-uniform
    auto run = [&](auto nkf_tag) {
        for (int qf = wave; qf * 16 < L; qf += 4) {
            const uint4 q0 = q0n, q1 = q1n;
            if ((qf + 4) * 16 < L) load_q(qf + 4, q0n, q1n);
            if (full) fragment(qf, q0, q1, nkf_tag, std::true_type{});
            else fragment(qf, q0, q1, nkf_tag, std::false_type{});
        }
    };
    if constexpr (MAXKF > 16) {
        run(std::integral_constant<int, MAXKF>{});
    } else {
        switch (nkf) {                                           // Lp is a multiple of 32: nkf is even
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            case 8: run(std::integral_constant<int, 8>{}); break;
            default:
                if constexpr (MAXKF > 8) {
                    switch (nkf) {
                        case 10: run(std::integral_constant<int, 10>{}); break;
                        case 12: run(std::integral_constant<int, 12>{}); break;
                        case 14: run(std::integral_constant<int, 14>{}); break;
                        default: run(std::integral_constant<int, 16>{}); break;
                    }
                }
                break;
        }
    }
}

#if ATLAS_TUNING   // (measured and NOT adopted: the tuning build keeps it for the A/B of tools/enc_knob_ab.py, profiles/r06/enc_knob_ab.txt)
// ------------------------------------------------------------------------------------------
// attention_pf_kernel (round 6): the VROW attention of the 16-bit bulk path, PERSISTENT and PREFETCHING.
// What bounded attention_kernel<.., VROW> (96 us per layer at 512 x 128 tokens = 4.2 TB/s of its 400 MB, 12 % MFMA-busy, VALU a third of the
// time): bytes in flight. A workgroup requested its 32 KB of K | V, waited, and only then computed; with four workgroups per CU in
// different phases about ONE of them had loads outstanding at any time: 32 KB per CU / ~2 us = 16 GB/s per CU = 4 TB/s. Here a workgroup
// walks its (passage, head) items -- item = blockIdx + i * gridDim, so neighbouring workgroups read neighbouring heads of one passage's
// q | k | v rows -- and requests the NEXT item's K | V chunks into registers right after the current item's tile went to LDS: the loads are in
// flight for the whole of the current item's arithmetic, every resident workgroup has 32 KB outstanding all the time. The barriers are
// `s_waitcnt lgkmcnt(0); s_barrier` (what LDS visibility needs), not __syncthreads (whose fence also drains vmcnt: the prefetch and the
// query loads must stay in flight across them).
// Also: the V tile's row pitch is 160 B (144 B put rows 0 and 7 of a 32-lane group of `ds_read_b64_tr_b16` on four common banks: 27 % of the
// kernel's LDS cycles were conflicts; 40 r mod 64 covers the 64 banks with 8 rows of 8 banks exactly once), and P.V runs with the MFMA operands
// swapped -- ctx^T = V^T . P^T, the same products summed over the same keys -- so that a lane ends up with FOUR CONSECUTIVE dims of one
// query row: 4 stores of 8 bytes per query fragment instead of 16 of 2 bytes (with their 16 address computations).
// Arithmetic per element is that of attention_kernel, in the same order: embeddings are bit-identical (tests/test_gpu_encoder.py).
// ------------------------------------------------------------------------------------------
#define ATT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <class T, int MAXKF, int WPC>
__global__ void __launch_bounds__(256, WPC)                     // WPC workgroups per CU: the registers must leave room for them (3: 168 VGPRs, 2: 256)
attention_pf_kernel(const uint16_t* __restrict__ qkv, const int* __restrict__ cu, const int nitems, const int LpMax, uint16_t* __restrict__ ctx) {
    static_assert(MAXKF == 8, "longer passages keep attention_kernel<T, MAXKF, true> (the prefetch of Lp / 16 chunks per thread does not fit their registers)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VPITCH = 160, QLD = 3 * HID;
    constexpr int NCH = MAXKF / 2;                             // 16-byte chunks of K (and of V) per thread: MAXKF * 16 keys x 8 chunks / 256 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    uint4* const sK = (uint4*)smem;                            // [LpMax][8], chunk c of key r at c ^ (r & 7)
    unsigned char* const sV = smem + (size_t)LpMax * 128;      // [LpMax] key rows of 64 dims at a 160-byte pitch
    float* const sMask = (float*)(sV + (size_t)LpMax * VPITCH);

    uint4 kv[NCH], vv[NCH];
    // the K | V chunks of `item`, requested and not waited for (keys past the passage's end are clamped: they are zeroed on their way to LDS)
    auto request = [&](const int item) {
        const int b = item / NHEAD, h = item - b * NHEAD;
        const int tb = cu[b], L = cu[b + 1] - tb;
        if (L <= 0) return;
        const int Lp8 = ((L + 31) & ~31) * 8;
        const uint16_t* Kb = qkv + (size_t)tb * QLD + HID + h * DHEAD;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i * 256 >= Lp8) break;                          // workgroup-uniform
            const int idx = i * 256 + tid, key = idx >> 3, ch = idx & 7;
            const uint16_t* src = Kb + (size_t)(key < L ? key : L - 1) * QLD + ch * 8;
            kv[i] = *(const uint4*)src;
            vv[i] = *(const uint4*)(src + HID);
        }
    };
    int item = blockIdx.x;
    if (item >= nitems) return;
    request(item);
    for (; item < nitems; item += gridDim.x) {
        const int b = item / NHEAD, h = item - b * NHEAD;
        const int tb = cu[b], L = cu[b + 1] - tb;
        const int nxt = item + gridDim.x;
        if (L <= 0) {                                           // (an all-masked passage: nothing to compute; workgroup-uniform, no barrier skipped unevenly)
            if (nxt < nitems) request(nxt);
            continue;
        }
        const int Lp = (L + 31) & ~31, nkf = Lp / 16;
        const uint16_t* Qb = qkv + (size_t)tb * QLD + h * DHEAD;
        auto load_q = [&](const int qf, uint4& qa, uint4& qb) {
            int qrow = qf * 16 + lr; if (qrow >= L) qrow = L - 1;
            qa = *(const uint4*)(Qb + (size_t)qrow * QLD + lg * 8);
            qb = *(const uint4*)(Qb + (size_t)qrow * QLD + 32 + lg * 8);
        };
        // this wave's first two query fragments: requested BEFORE the next item's prefetch, so that waiting for them never waits for it
        uint4 qa0 = make_uint4(0, 0, 0, 0), qa1 = qa0, qb0 = qa0, qb1 = qa0;
        if (wave * 16 < L) load_q(wave, qa0, qa1);
        if ((wave + 4) * 16 < L) load_q(wave + 4, qb0, qb1);
        // the tile of THIS item: registers -> LDS
        for (int j = tid; j < Lp; j += 256) sMask[j] = (j < L) ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i * 256 >= Lp * 8) break;
            const int idx = i * 256 + tid, key = idx >> 3, ch = idx & 7;
            const bool real = key < L;
            sK[key * 8 + (ch ^ (key & 7))] = real ? kv[i] : make_uint4(0, 0, 0, 0);
            *(uint4*)(sV + key * VPITCH + ch * 16) = real ? vv[i] : make_uint4(0, 0, 0, 0);
        }
        ATT_LDS_BARRIER();
        if (nxt < nitems) request(nxt);                         // in flight under everything below

        auto fragment = [&](const int qf, const uint4 q0, const uint4 q1, auto nkf_tag, auto full_tag) {
            constexpr int NKF = decltype(nkf_tag)::value;
            asm volatile("" ::: "memory");
            constexpr bool FULL = decltype(full_tag)::value;
            f4 s[NKF];
#pragma unroll
            for (int kf = 0; kf < NKF; ++kf) {
                const int krow = kf * 16 + lr;
                const uint4 k0 = sK[krow * 8 + (lg ^ (krow & 7))];
                const uint4 k1 = sK[krow * 8 + ((4 + lg) ^ (krow & 7))];
                s[kf] = T::mma(k0, q0, (f4){0.f, 0.f, 0.f, 0.f});
                s[kf] = T::mma(k1, q1, s[kf]);
            }
            uint32_t pk[NKF][2];
            if constexpr (T::DT == ATLAS_DT_F16) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                h2 v[NKF][2];
                h2 mx2 = {(_Float16)(-__builtin_inff()), (_Float16)(-__builtin_inff())};
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        h2 x = {(_Float16)s[kf][2 * e], (_Float16)s[kf][2 * e + 1]};
                        x = x * (h2){(_Float16)0.125f, (_Float16)0.125f};
                        if (!FULL) {
                            const float2 am = *(const float2*)(sMask + kf * 16 + lg * 4 + 2 * e);
                            x = x + (h2){(_Float16)am.x, (_Float16)am.y};
                        }
                        v[kf][e] = x;
                        mx2 = __builtin_elementwise_max(mx2, x);
                    }
                float mx = fmaxf((float)mx2.x, (float)mx2.y);
                mx = rows4_max(mx);
                const float nmx = -mx * 1.4426950408889634f;
                f2 ex[NKF][2];
                f2 sum2 = {0.f, 0.f};
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f2 ee = {__builtin_amdgcn_exp2f(__builtin_fmaf((float)v[kf][e].x, 1.4426950408889634f, nmx)),
                                       __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[kf][e].y, 1.4426950408889634f, nmx))};
                        ex[kf][e] = ee;
                        sum2 += ee;
                    }
                float sum = sum2.x + sum2.y;
                sum = rows4_sum(sum);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f2 pe = ex[kf][e] * (f2){inv, inv};
                        const h2 ph = {(_Float16)pe.x, (_Float16)pe.y};
                        pk[kf][e] = __builtin_bit_cast(uint32_t, ph);
                    }
            } else {
                float mx = -__builtin_inff();
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf) {
                    const float4 am = *(const float4*)(sMask + kf * 16 + lg * 4);
                    const float a4[4] = {am.x, am.y, am.z, am.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float vv_ = T::rnd(T::rnd(s[kf][r]) * 0.125f + a4[r]);
                        s[kf][r] = vv_;
                        mx = fmaxf(mx, vv_);
                    }
                }
                mx = rows4_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(s[kf][r] - mx);
                        s[kf][r] = e;
                        sum += e;
                    }
                sum = rows4_sum(sum);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        pk[kf][e] = (uint32_t)T::st(s[kf][2 * e] * inv) | ((uint32_t)T::st(s[kf][2 * e + 1] * inv) << 16);
            }
            // ctx^T = V^T . P^T: o[df][r] = ctx[query 16 qf + lr][dim 16 df + 4 lg + r]
            f4 o[4];
#pragma unroll
            for (int df = 0; df < 4; ++df) o[df] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKF / 2; ++ks) {
                const uint4 pa = make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]);
                const uint32_t va = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sV +
                                    (uint32_t)((32 * ks + 4 * lg + (lr >> 2)) * VPITCH + 8 * (lr & 3));
                unsigned long long t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:2560\n\t"
                             "ds_read_b64_tr_b16 %2, %8 offset:32\n\tds_read_b64_tr_b16 %3, %8 offset:2592\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:64\n\tds_read_b64_tr_b16 %5, %8 offset:2624\n\t"
                             "ds_read_b64_tr_b16 %6, %8 offset:96\n\tds_read_b64_tr_b16 %7, %8 offset:2656\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(va) : "memory");
                const unsigned long long tv[4][2] = {{t0, t1}, {t2, t3}, {t4, t5}, {t6, t7}};
#pragma unroll
                for (int df = 0; df < 4; ++df)
                    o[df] = T::mma(make_uint4((uint32_t)tv[df][0], (uint32_t)(tv[df][0] >> 32), (uint32_t)tv[df][1], (uint32_t)(tv[df][1] >> 32)), pa, o[df]);
            }
            const int row = qf * 16 + lr;
            if (FULL || row < L) {
                uint16_t* dst = ctx + ((size_t)tb + row) * HID + h * DHEAD + 4 * lg;
#pragma unroll
                for (int df = 0; df < 4; ++df)
                    *(uint2*)(dst + df * 16) = make_uint2((uint32_t)T::st(o[df][0]) | ((uint32_t)T::st(o[df][1]) << 16),
                                                          (uint32_t)T::st(o[df][2]) | ((uint32_t)T::st(o[df][3]) << 16));
            }
        };
        const bool full = (L == Lp);
        auto run = [&](auto nkf_tag) {
            for (int qf = wave; qf * 16 < L; qf += 4) {
                const uint4 q0 = qa0, q1 = qa1;
                qa0 = qb0; qa1 = qb1;
                if ((qf + 8) * 16 < L) load_q(qf + 8, qb0, qb1);
                if (full) fragment(qf, q0, q1, nkf_tag, std::true_type{});
                else fragment(qf, q0, q1, nkf_tag, std::false_type{});
            }
        };
        switch (nkf) {                                           // Lp is a multiple of 32: nkf is even
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            case 8: run(std::integral_constant<int, 8>{}); break;
            default:
                if constexpr (MAXKF > 8) {
                    switch (nkf) {
                        case 10: run(std::integral_constant<int, 10>{}); break;
                        case 12: run(std::integral_constant<int, 12>{}); break;
                        case 14: run(std::integral_constant<int, 14>{}); break;
                        default: run(std::integral_constant<int, 16>{}); break;
                    }
                }
                break;
        }
        ATT_LDS_BARRIER();                                       // every wave is done with this item's tile before the next one overwrites it
    }
}

#endif

// fp32 model precision (query embedding with --precision fp32, atlas.py:104): same S^T trick with v_mfma 16x16x4 f32.
// After S^T = K.Q^T lane (lr, lg) holds query lr x keys 16kf+4lg+r; taking the contraction index of P.V step (kf, r)
// as lane group lg <-> key 16kf+4lg+r makes those registers the A operand as they are, and the B operand
// V^T[dim][16kf+4lg .. +3] one 16-byte load. Operands come straight from global memory / L2: this path serves query
// batches (tens of tokens each after packing), not the bulk refresh. L <= 512.
template <int MAXKF>
__global__ void __launch_bounds__(256)
attention_f32_kernel(const float* __restrict__ qk, const float* __restrict__ vt, const int* __restrict__ cu, int LpMax,
                     float* __restrict__ ctx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x / NHEAD, h = blockIdx.x % NHEAD;
    const int tb = cu[b], L = cu[b + 1] - tb;
    if (L <= 0) return;
    const int nkf = (L + 15) >> 4;
    const float* Qb = qk + (size_t)tb * (2 * HID) + h * DHEAD;
    const float* Kb = Qb + HID;
    const float* Vt = vt + ((size_t)b * HID + h * DHEAD) * LpMax + (tb & 7);
    for (int qf = wave; qf * 16 < L; qf += 4) {
        int qrow = qf * 16 + lr; if (qrow >= L) qrow = L - 1;
        uint4 q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = *(const uint4*)(Qb + (size_t)qrow * (2 * HID) + c * 16 + lg * 4);
        f4 s[MAXKF];
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf) {
            s[kf] = (f4){0.f, 0.f, 0.f, 0.f};
            if (kf < nkf) {
                int krow = kf * 16 + lr; if (krow >= L) krow = L - 1;      // (masked below)
#pragma unroll
                for (int c = 0; c < 4; ++c) s[kf] = F32::mma(*(const uint4*)(Kb + (size_t)krow * (2 * HID) + c * 16 + lg * 4), q[c], s[kf]);
            }
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf)
            if (kf < nkf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (kf * 16 + lg * 4 + r < L) ? s[kf][r] * 0.125f : -__builtin_inff();   // / sqrt(64), + 0 mask
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
                    const float e = expf(s[kf][r] - mx);
                    s[kf][r] = e;
                    sum += e;
                }
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        f4 o[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < MAXKF; ++kf)
            if (kf < nkf) {
                const int key0 = kf * 16 + lg * 4;
                const uint4 pa = make_uint4(__builtin_bit_cast(uint32_t, s[kf][0] / sum), __builtin_bit_cast(uint32_t, s[kf][1] / sum),
                                            __builtin_bit_cast(uint32_t, s[kf][2] / sum), __builtin_bit_cast(uint32_t, s[kf][3] / sum));
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    uint4 vv = *(const uint4_a4*)(Vt + (size_t)(df * 16 + lr) * LpMax + key0);
                    if (key0 + 0 >= L) vv.x = 0u;                          // columns >= L were never written
                    if (key0 + 1 >= L) vv.y = 0u;
                    if (key0 + 2 >= L) vv.z = 0u;
                    if (key0 + 3 >= L) vv.w = 0u;
                    o[df] = F32::mma(pa, vv, o[df]);
                }
            }
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qf * 16 + lg * 4 + r;
                if (row < L) ctx[((size_t)tb + row) * HID + h * DHEAD + df * 16 + lr] = o[df][r];
            }
    }
}

// masked mean pooling (retrievers.py:50-52) over the packed tokens of one passage, with the reference's two roundings:
// dt(sum) (the tensor .sum(dim=1) returns), then dt(that / count); the row goes straight to out (= a slab row,
// atlas.py:79). The sum is accumulated in double (exact for fp16/bf16 addends), which no summation order of the
// reference beats.
template <class T>
__global__ void __launch_bounds__(192)
pool_packed_kernel(const typename T::elem* __restrict__ x, const int* __restrict__ cu, const int2* __restrict__ tokinfo, int mode,
                   void* __restrict__ out_, const int64_t* __restrict__ out_rows /* nullable: row of `out` that passage b goes to */) {
    const int b = blockIdx.x;
    const int64_t ob = out_rows ? out_rows[b] : (int64_t)b;
    const int tb = cu[b], L = cu[b + 1] - tb;
    const typename T::elem* base = x + (size_t)tb * HID + threadIdx.x * 4;
    if (mode == ATLAS_POOL_CLS) {
        // last_hidden[:, 0] after masked_fill (retrievers.py:50, 55-56): position 0 is the first packed token if unmasked
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (L > 0 && (tokinfo[tb].y & 0xffff) == 0) load4<T>(base, v);
        store4<T>((typename T::elem*)out_ + (size_t)ob * HID + threadIdx.x * 4, v);
        return;
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int l = 0; l < L; ++l) {
        float v[4];
        load4<T>(base + (size_t)l * HID, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += (double)v[r];
    }
    const float cnt = (float)L;                                   // attention_mask.sum(dim=1): 0 -> 0/0 = NaN as in torch
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sum;
        if (T::DT == ATLAS_DT_F16) sum = f16_bits_to_f32(f64_to_f16_bits(s[r]));     // single rounding double -> fp16
        else if (T::DT == ATLAS_DT_BF16) sum = T::rnd((float)s[r]);                  // (double -> float -> bf16: the float step is
        else sum = (float)s[r];                                                       //  exact unless > 24 significant bits are live)
        o[r] = (mode == ATLAS_POOL_SQRT) ? sum / sqrtf(cnt) : sum / cnt;              // retrievers.py:53-54 / :51-52
    }
    if (mode == ATLAS_POOL_SQRT)     // dtype tensor / fp32 tensor promotes: the reference returns fp32 here
        *(float4*)((float*)out_ + (size_t)ob * HID + threadIdx.x * 4) = make_float4(o[0], o[1], o[2], o[3]);
    else
        store4<T>((typename T::elem*)out_ + (size_t)ob * HID + threadIdx.x * 4, o);
}

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
inline size_t esize(int dt) { return dt == ATLAS_DT_F32 ? 4 : 2; }

template <class T>
int run_encoder(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                int n, int L, void* out, const int64_t* out_rows, void* ws, hipStream_t stream) {
    typedef typename T::elem E;
    const size_t es = sizeof(E);
    const int64_t M = (int64_t)n * L;                 // worst case; the packed count lives on the device (cu[n])
    const int Lp = (L + 31) & ~31;
    unsigned char* p = (unsigned char*)ws;
    E* x = (E*)p;    p += up256((size_t)M * HID * es);
    E* ctx = (E*)p;  p += up256((size_t)M * HID * es);
    E* u = (E*)p;    p += up256((size_t)M * HID * es);
    E* qk = (E*)p;   p += up256((size_t)M * 2 * HID * es);
    const int LpS = Lp + 8;                            // V^T row pitch: a passage's keys start up to 7 columns in (layout note)
    E* vt = (E*)p;   p += up256((size_t)n * HID * LpS * es);
    E* hbuf = (E*)p; p += up256((size_t)M * 4 * HID * es);
    int* counts = (int*)p;        p += up256((size_t)n * 4);
    int* cu = (int*)p;            p += up256((size_t)(n + 1) * 4);
    int2* tokinfo = (int2*)p;

    const unsigned tok_blocks = (unsigned)((M + 3) / 4), pas_blocks = (unsigned)((n + 3) / 4);
    // configuration table: see launch_gemm. Small batches (queries) need more, smaller tiles to cover the 256 CUs: 64 queries
    // x ~20 tokens are 21 x 12 tiles of 64x64 for a 768-wide GEMM
    const int cfg = g_gemm_cfg >= 0 ? g_gemm_cfg : (M > 16384 ? 9 : (M > 4096 ? 0 : 3));
    hipLaunchKernelGGL(count_kernel, dim3(pas_blocks), dim3(256), 0, stream, attention_mask, n, L, counts);
    hipLaunchKernelGGL(pack_kernel, dim3(pas_blocks), dim3(256), 0, stream, attention_mask, n, L, counts, cu, tokinfo);
    hipLaunchKernelGGL(embed_ln_kernel<T>, dim3(tok_blocks), dim3(256), 0, stream, input_ids, token_type_ids, L, cu, n, tokinfo,
                       w->vocab_size, w->type_vocab,
                       (const E*)w->word_emb, (const E*)w->pos_emb, (const E*)w->type_emb, (const E*)w->emb_ln_w,
                       (const E*)w->emb_ln_b, w->eps, x);
    for (int l = 0; l < w->n_layers; ++l) {
        const atlas_bert_layer& ly = w->layers[l];
        launch_gemm<T, 3>(cfg, stream, x, (const E*)ly.qkv_w, (const E*)ly.qkv_b, (const E*)nullptr, qk, vt, M, cu, n, tokinfo,
                          3 * HID, HID, LpS);
        if constexpr (T::DT == ATLAS_DT_F32) {
            if (Lp <= 128)
                hipLaunchKernelGGL(attention_f32_kernel<8>, dim3((unsigned)n * NHEAD), dim3(256), 0, stream, qk, vt, cu, LpS, ctx);
            else
                hipLaunchKernelGGL(attention_f32_kernel<32>, dim3((unsigned)n * NHEAD), dim3(256), 0, stream, qk, vt, cu, LpS, ctx);
        } else {
            // (the persistent bulk GEMM -- cfg 9, 16-bit -- left q | k | v row-major in [M, 2304] starting at `qk`: the V^T region behind it is
            //  part of that buffer then; every other configuration keeps q | k [M, 1536] + V^T)
            const bool vrow = (cfg == 9);
            const size_t att_lds = vrow ? (size_t)Lp * 128 + (size_t)Lp * 160 + (size_t)Lp * 4
                                        : (size_t)Lp * 128 + (size_t)64 * (Lp + 8) * 2 + (size_t)Lp * 4;
            auto att = [&](auto kern) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                const unsigned grid = g_att_xmap ? (unsigned)((n + 7) / 8 * 8) * NHEAD : (unsigned)n * NHEAD;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), att_lds, stream, qk, vt, cu, LpS, ctx, n, (int)g_att_xmap);
            };
#if ATLAS_TUNING
            if (vrow && Lp <= 128 && g_att_pf) {
                // round 6: persistent, prefetching (attention_pf_kernel) for batches of up to 128 tokens per passage; longer passages keep one
                // workgroup per item (their prefetch does not fit the registers). g_att_pf (tuning build): 2 / 3 = workgroups per CU
                const int nitems = n * NHEAD;
                const size_t pf_lds = (size_t)Lp * (128 + 160 + 4);
                const int per_cu = g_att_pf == 3 ? 3 : 2;
                int grid = encoder_device_cus() * per_cu;
                if (grid > nitems) grid = nitems;
                auto attp = [&](auto kern) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), pf_lds, stream, (const uint16_t*)qk, cu, nitems, Lp, (uint16_t*)ctx);
                };
                if (per_cu == 3) attp(attention_pf_kernel<T, 8, 3>); else attp(attention_pf_kernel<T, 8, 2>);
            } else
#endif
            if (vrow) {
                if (Lp <= 128) att(attention_kernel<T, 8, true>);
                else if (Lp <= 256) att(attention_kernel<T, 16, true>);
                else att(attention_kernel<T, 32, true>);
            } else if (Lp <= 128) att(attention_kernel<T, 8>);
            else if (Lp <= 256) att(attention_kernel<T, 16>);
            else att(attention_kernel<T, 32>);
        }
        launch_gemm<T, 2>(cfg, stream, ctx, (const E*)ly.o_w, (const E*)ly.o_b, x, u, (E*)nullptr, M, cu, n, tokinfo, HID, HID, Lp);
        if (!g_skip_ln)
        hipLaunchKernelGGL(ln_kernel<T>, dim3(tok_blocks), dim3(256), 0, stream, u, cu + n, (const E*)ly.ln1_w, (const E*)ly.ln1_b,
                           w->eps, x);
        launch_gemm<T, 1>(cfg, stream, x, (const E*)ly.ff1_w, (const E*)ly.ff1_b, (const E*)nullptr, hbuf, (E*)nullptr, M, cu, n,
                          tokinfo, 4 * HID, HID, Lp);
        launch_gemm<T, 2>(cfg, stream, hbuf, (const E*)ly.ff2_w, (const E*)ly.ff2_b, x, u, (E*)nullptr, M, cu, n, tokinfo, HID,
                          4 * HID, Lp);
        if (!g_skip_ln)
        hipLaunchKernelGGL(ln_kernel<T>, dim3(tok_blocks), dim3(256), 0, stream, u, cu + n, (const E*)ly.ln2_w, (const E*)ly.ln2_b,
                           w->eps, x);
    }
    // rows written contiguously at out (which may point into the passage slab: slab + row_offset * 768)
    hipLaunchKernelGGL(pool_packed_kernel<T>, dim3((unsigned)n), dim3(192), 0, stream, x, cu, tokinfo, w->pooling, out, out_rows);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" {

#if ATLAS_TUNING
// tuning build only (libatlas_hip_tune.so; not part of include/atlas_hip.h)
void atlas_tune_set_gemm_stamps(unsigned long long* p) { g_gemm_dbg = p; g_pt_stamp_nth = 0; }
void atlas_tune_set_gemm_stamps_nth(unsigned long long* p, int nth) { g_gemm_dbg = p; g_pt_stamp_nth = nth; g_pt_launches = 0; }
void atlas_tune_set_gemm_diag(int d) { g_gemm_diag = d; }
void atlas_tune_set_gemm_cfg(int c) { g_gemm_cfg = c; }
void atlas_tune_set_att_pf(int v) { g_att_pf = v; }
void atlas_tune_set_skip_ln(int v) { g_skip_ln = v; }
void atlas_tune_set_att_xmap(int v) { g_att_xmap = v; }
#endif

size_t atlas_contriever_workspace_bytes(int n, int L, int dtype) {
    if (n <= 0 || L <= 0) return 0;
    const size_t M = (size_t)n * L, Lp = (size_t)((L + 31) & ~31), es = esize(dtype);
    // x, ctx, u : [M,768]; qk : [M,1536]; vt : [n,768,Lp]; h : [M,3072]; counts[n], cu[n+1], tokinfo[M]
    return up256(M * HID * es) * 3 + up256(M * 2 * HID * es) + up256((size_t)n * HID * (Lp + 8) * es) + up256(M * 4 * HID * es) +
           up256((size_t)n * 4) + up256((size_t)(n + 1) * 4) + up256(M * 8) + 256;
}

int atlas_contriever_embed(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask,
                           const int64_t* token_type_ids, int n, int L, void* out, void* ws, size_t ws_bytes,
                           void* stream_) {
    return atlas_contriever_embed_rows(w, input_ids, attention_mask, token_type_ids, n, L, out, nullptr, ws, ws_bytes, stream_);
}

int atlas_contriever_embed_rows(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask,
                                const int64_t* token_type_ids, int n, int L, void* out, const int64_t* out_rows, void* ws,
                                size_t ws_bytes, void* stream_) {
    if (!w || !input_ids || !attention_mask || !out || !ws) return ATLAS_E_BADARG;
    if (n <= 0 || L <= 0) return ATLAS_E_BADARG;
    if (L > 512 || w->hidden != HID || w->n_heads != NHEAD || w->intermediate != 4 * HID || w->n_layers < 1 ||
        w->n_layers > ATLAS_BERT_MAX_LAYERS || (int64_t)n * L > 0x7fffffff)
        return ATLAS_E_UNSUPPORTED;
    if (w->dtype != ATLAS_DT_F16 && w->dtype != ATLAS_DT_BF16 && w->dtype != ATLAS_DT_F32) return ATLAS_E_UNSUPPORTED;
    if (w->pooling < ATLAS_POOL_AVERAGE || w->pooling > ATLAS_POOL_CLS) return ATLAS_E_UNSUPPORTED;
    if (w->vocab_size < 1 || w->type_vocab < 1 || w->max_positions < 1 || L > w->max_positions) return ATLAS_E_BADARG;
    if (ws_bytes < atlas_contriever_workspace_bytes(n, L, w->dtype)) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (w->dtype == ATLAS_DT_F16) return run_encoder<F16>(w, input_ids, attention_mask, token_type_ids, n, L, out, out_rows, ws, stream);
    if (w->dtype == ATLAS_DT_BF16) return run_encoder<BF16>(w, input_ids, attention_mask, token_type_ids, n, L, out, out_rows, ws, stream);
    return run_encoder<F32>(w, input_ids, attention_mask, token_type_ids, n, L, out, out_rows, ws, stream);
}

}  // extern "C"
