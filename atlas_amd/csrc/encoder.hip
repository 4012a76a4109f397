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
// Kernels
//   embed_ln_kernel     word + type (+= position) in fp16, LayerNorm                    (one wave per token)
//   gemm_bt_kernel      C[M,N] = A[M,K] . W[N,K]^T + bias, 128x128x64 tiles, MFMA 16x16x32 f16, LDS double buffer,
//                       epilogues: plain | exact-erf GELU | + residual        (MFMA-bound: the refresh roofline)
//   attention_kernel    per (passage, head): QK^T -> fp16 -> /8 + mask -> fp32 softmax -> fp16 P -> PV
//   ln_kernel           the reference's LayerNorm on a [M,768] fp16 tensor          (one wave per token)
//   (pooling + slab row write: pool_write_kernel in atlas_hip.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

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

// one wave per token: embeddings (modeling_bert.py:213-247)
__global__ void __launch_bounds__(256)
embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids, int L, int64_t M,
                const uint16_t* __restrict__ word, const uint16_t* __restrict__ pos, const uint16_t* __restrict__ type,
                const uint16_t* __restrict__ lnw, const uint16_t* __restrict__ lnb, float eps, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    const int64_t id = ids[t], ty = type_ids ? type_ids[t] : 0;
    const int p = (int)(t % L);
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
ln_kernel(const uint16_t* __restrict__ in, int64_t M, const uint16_t* __restrict__ lnw, const uint16_t* __restrict__ lnb,
          float eps, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
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
template <int EPI>
__global__ void __launch_bounds__(256)
gemm_bt_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias,
               const uint16_t* __restrict__ R, uint16_t* __restrict__ C, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) uint4 sW[2][128 * 8];
    __shared__ __attribute__((aligned(16))) uint4 sA[2][128 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int n0 = blockIdx.x * 128;
    const int64_t m0 = (int64_t)blockIdx.y * 128;
    const int lr = lane & 15, lg = lane >> 4;

    // staging: chunk idx = tid + 256*i -> row idx>>3, 16-B chunk idx&7
    uint4 rw[4], ra[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
            rw[i] = *(const uint4*)(W + (size_t)(n0 + row) * K + k0 + ch * 8);
            int64_t ar = m0 + row;
            if (ar >= M) ar = M - 1;                                   // clamped: tail rows are never stored
            ra[i] = *(const uint4*)(A + (size_t)ar * K + k0 + ch * 8);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
            sW[buf][row * 8 + (ch ^ (row & 7))] = rw[i];
            sA[buf][row * 8 + (ch ^ (row & 7))] = ra[i];
        }
    };

    f4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / 64;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 fw[4], fa[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = wi * 64 + a * 16 + lr;
                fw[a] = __builtin_bit_cast(h8, sW[buf][row * 8 + ((ks * 4 + lg) ^ (row & 7))]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = wj * 64 + b * 16 + lr;
                fa[b] = __builtin_bit_cast(h8, sA[buf][row * 8 + ((ks * 4 + lg) ^ (row & 7))]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[a], fa[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: acc[a][b][r] = C[token m0+64wj+16b+lr][col n0+64wi+16a+4lg+r]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t tok = m0 + wj * 64 + b * 16 + lr;
        if (tok >= M) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n0 + wi * 64 + a * 16 + lg * 4;
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
            *(uint2*)(C + (size_t)tok * N + col) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
    }
}

// ------------------------------------------------------------------------------------------
// attention: one block (4 waves) per (passage, head); QKV is [M][3*768] with q | k | v column blocks.
//   scores = fp16(q.k^T) ; / 8 (exact) ; + fp16 mask (0 / -10000) ; softmax in fp32 ; P = fp16 ; ctx = fp16(P.v)
// K rows feed the MFMA B operand straight from global memory (8 consecutive head dims per lane); V is staged
// TRANSPOSED in LDS ([64 dims][Lp keys]) because the PV product needs 8 consecutive keys per lane; P goes through
// LDS once to turn the C-fragment layout into A fragments. Each wave owns query fragments w, w+4, ...
// L <= 512; Lp = L rounded up to 32.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
attention_kernel(const uint16_t* __restrict__ qkv, const int64_t* __restrict__ mask, int L, uint16_t* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Lp = (L + 31) & ~31;
    uint16_t* sVt = (uint16_t*)smem;                          // [64][Lp + 8]  (+8 halfs pad: staggers banks)
    const int vstride = Lp + 8;
    uint16_t* sP = sVt + 64 * vstride;                        // [4 waves][16][Lp + 8]
    float* sMask = (float*)(sP + 4 * 16 * vstride);           // [Lp] additive mask as fp32 (0 or -10000), -inf beyond L
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x / NHEAD, h = blockIdx.x % NHEAD;
    const uint16_t* base = qkv + (size_t)b * L * (3 * HID);
    const uint16_t* Qb = base + h * DHEAD;
    const uint16_t* Kb = base + HID + h * DHEAD;
    const uint16_t* Vb = base + 2 * HID + h * DHEAD;

    for (int j = tid; j < Lp; j += 256)
        sMask[j] = (j < L) ? ((mask[(size_t)b * L + j] != 0) ? 0.0f : -10000.0f) : -__builtin_inff();
    // V^T into LDS: thread handles (key j, 8-dim chunk c)
    for (int idx = tid; idx < Lp * 8; idx += 256) {
        const int j = idx >> 3, c = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (j < L) v = *(const uint4*)(Vb + (size_t)j * (3 * HID) + c * 8);
        const uint16_t e[8] = {(uint16_t)(v.x & 0xffff), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffff), (uint16_t)(v.y >> 16),
                               (uint16_t)(v.z & 0xffff), (uint16_t)(v.z >> 16), (uint16_t)(v.w & 0xffff), (uint16_t)(v.w >> 16)};
#pragma unroll
        for (int d = 0; d < 8; ++d) sVt[(c * 8 + d) * vstride + j] = e[d];
    }
    __syncthreads();

    const int nkf = Lp / 16;                                   // key fragments (<= 32)
    uint16_t* myP = sP + wave * 16 * vstride;
    for (int qf = wave; qf * 16 < L; qf += 4) {
        // Q fragment: lane (row lr, k-group lg), 2 k-steps of 32 dims
        int qrow = qf * 16 + lr; if (qrow >= L) qrow = L - 1;
        const h8 q0 = __builtin_bit_cast(h8, *(const uint4*)(Qb + (size_t)qrow * (3 * HID) + lg * 8));
        const h8 q1 = __builtin_bit_cast(h8, *(const uint4*)(Qb + (size_t)qrow * (3 * HID) + 32 + lg * 8));
        // S = Q K^T : MFMA(A = Q rows, B = K rows) -> lane holds key col lr of fragment kf, query rows 4lg+r
        f4 s[32];
#pragma unroll
        for (int kf = 0; kf < 32; ++kf) {
            s[kf] = (f4){0.f, 0.f, 0.f, 0.f};
            if (kf < nkf) {
                int krow = kf * 16 + lr; if (krow >= L) krow = L - 1;
                const h8 k0 = __builtin_bit_cast(h8, *(const uint4*)(Kb + (size_t)krow * (3 * HID) + lg * 8));
                const h8 k1 = __builtin_bit_cast(h8, *(const uint4*)(Kb + (size_t)krow * (3 * HID) + 32 + lg * 8));
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0, k0, s[kf], 0, 0, 0);
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1, k1, s[kf], 0, 0, 0);
            }
        }
        // softmax over keys for each of this lane's 4 query rows; a row's keys live in the 16 lanes sharing lg
        float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int kf = 0; kf < 32; ++kf)
            if (kf < nkf) {
                const float am = sMask[kf * 16 + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // fp16(q.k) / sqrt(64) is exact in fp16; + mask is an fp16 add (modeling_bert.py:346-349)
                    const float v = rh(rh(s[kf][r]) * 0.125f + am);
                    s[kf][r] = v;
                    mx[r] = fmaxf(mx[r], v);
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o));
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < 32; ++kf)
            if (kf < nkf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kf][r] - mx[r]);          // exp(-inf) = 0 for padded key columns
                    s[kf][r] = e;
                    sum[r] += e;
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o);
            sum[r] = 1.0f / sum[r];
        }
        // P (fp16) -> LDS [query row][key]
#pragma unroll
        for (int kf = 0; kf < 32; ++kf)
            if (kf < nkf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) myP[(lg * 4 + r) * vstride + kf * 16 + lr] = f2h(s[kf][r] * sum[r]);
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's P is in LDS (same-wave reads follow)
        // ctx = P V : MFMA(A = P rows [16 x keys], B = V^T rows [dims x keys]) -> lane holds dim col lr of fragment df, query rows 4lg+r
        f4 o[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < Lp / 32; ++ks) {
            const h8 pa = __builtin_bit_cast(h8, *(const uint4*)(myP + lr * vstride + ks * 32 + lg * 8));
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const h8 vb = __builtin_bit_cast(h8, *(const uint4*)(sVt + (df * 16 + lr) * vstride + ks * 32 + lg * 8));
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, vb, o[df], 0, 0, 0);
            }
        }
        // context_layer.permute(0,2,1,3).view(.., 768): [token][h*64 + dim]
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qf * 16 + lg * 4 + r;
                if (row < L) ctx[((size_t)b * L + row) * HID + h * DHEAD + df * 16 + lr] = f2h(o[df][r]);
            }
    }
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
    const size_t M = (size_t)n * L;
    // x, ctx, u : [M,768]; qkv : [M,2304]; h : [M,3072]; pooled input reuses x
    return up256(M * HID * 2) * 3 + up256(M * 3 * HID * 2) + up256(M * 4 * HID * 2) + 256;
}

int atlas_contriever_embed(const atlas_bert_weights* w, const int64_t* input_ids, const int64_t* attention_mask,
                           const int64_t* token_type_ids, int n, int L, void* out_f16, void* ws, size_t ws_bytes,
                           void* stream_) {
    if (!w || !input_ids || !attention_mask || !out_f16 || !ws) return ATLAS_E_BADARG;
    if (n <= 0 || L <= 0) return ATLAS_E_BADARG;
    if (L > 512 || w->hidden != HID || w->n_heads != NHEAD || w->intermediate != 4 * HID || w->n_layers < 1 ||
        w->n_layers > ATLAS_BERT_MAX_LAYERS)
        return ATLAS_E_UNSUPPORTED;
    if (ws_bytes < atlas_contriever_workspace_bytes(n, L)) return ATLAS_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t M = (int64_t)n * L;
    unsigned char* p = (unsigned char*)ws;
    uint16_t* x = (uint16_t*)p;   p += up256((size_t)M * HID * 2);
    uint16_t* ctx = (uint16_t*)p; p += up256((size_t)M * HID * 2);
    uint16_t* u = (uint16_t*)p;   p += up256((size_t)M * HID * 2);
    uint16_t* qkv = (uint16_t*)p; p += up256((size_t)M * 3 * HID * 2);
    uint16_t* hbuf = (uint16_t*)p;

    const unsigned tok_blocks = (unsigned)((M + 3) / 4);
    const dim3 mt((unsigned)((M + 127) / 128));
    hipLaunchKernelGGL(embed_ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, input_ids, token_type_ids, L, M,
                       (const uint16_t*)w->word_emb, (const uint16_t*)w->pos_emb, (const uint16_t*)w->type_emb,
                       (const uint16_t*)w->emb_ln_w, (const uint16_t*)w->emb_ln_b, w->eps, x);
    const int Lp = (L + 31) & ~31;
    const size_t att_lds = (size_t)(64 + 4 * 16) * (Lp + 8) * 2 + (size_t)Lp * 4;
    (void)hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int l = 0; l < w->n_layers; ++l) {
        const atlas_bert_layer& ly = w->layers[l];
        hipLaunchKernelGGL(gemm_bt_kernel<0>, dim3(3 * HID / 128, mt.x), dim3(256), 0, stream, x, (const uint16_t*)ly.qkv_w,
                           (const uint16_t*)ly.qkv_b, (const uint16_t*)nullptr, qkv, M, 3 * HID, HID);
        hipLaunchKernelGGL(attention_kernel, dim3((unsigned)n * NHEAD), dim3(256), att_lds, stream, qkv, attention_mask, L, ctx);
        hipLaunchKernelGGL(gemm_bt_kernel<2>, dim3(HID / 128, mt.x), dim3(256), 0, stream, ctx, (const uint16_t*)ly.o_w,
                           (const uint16_t*)ly.o_b, x, u, M, HID, HID);
        hipLaunchKernelGGL(ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, u, M, (const uint16_t*)ly.ln1_w,
                           (const uint16_t*)ly.ln1_b, w->eps, x);
        hipLaunchKernelGGL(gemm_bt_kernel<1>, dim3(4 * HID / 128, mt.x), dim3(256), 0, stream, x, (const uint16_t*)ly.ff1_w,
                           (const uint16_t*)ly.ff1_b, (const uint16_t*)nullptr, hbuf, M, 4 * HID, HID);
        hipLaunchKernelGGL(gemm_bt_kernel<2>, dim3(HID / 128, mt.x), dim3(256), 0, stream, hbuf, (const uint16_t*)ly.ff2_w,
                           (const uint16_t*)ly.ff2_b, x, u, M, HID, 4 * HID);
        hipLaunchKernelGGL(ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, u, M, (const uint16_t*)ly.ln2_w,
                           (const uint16_t*)ly.ln2_b, w->eps, x);
    }
    // masked mean pooling with the reference's double rounding, rows written contiguously at out_f16
    // (out_f16 may point into the passage slab: slab + row_offset * 768)
    return atlas_pool_write(x, attention_mask, out_f16, n, 0, n, L, HID, stream_);
}

}  // extern "C"
