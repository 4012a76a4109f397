// common.h — arithmetic helpers shared by every kernel of the retrieval path.
//
// Everything here is `ATLAS_HD` (host+device) on purpose: tests/test_host_helpers.py
// compiles this header with g++ into a tiny host library and checks each helper against
// the independent restatement in oracle/oracle.c — the device arithmetic that decides
// scores, keys and pruning margins is therefore testable without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define ATLAS_HD __host__ __device__ __forceinline__
#else
#define ATLAS_HD static inline
#endif

namespace atlas {

// ---- bit casts ---------------------------------------------------------------------
ATLAS_HD uint32_t f32_bits(float f) { union { float f; uint32_t u; } x; x.f = f; return x.u; }
ATLAS_HD float bits_f32(uint32_t u) { union { float f; uint32_t u; } x; x.u = u; return x.f; }
ATLAS_HD uint64_t f64_bits(double f) { union { double f; uint64_t u; } x; x.f = f; return x.u; }
ATLAS_HD double bits_f64(uint64_t u) { union { double f; uint64_t u; } x; x.u = u; return x.f; }

// ---- fp16 <-> wider, bit-level, round-to-nearest-even, subnormals kept ----------------
// (src/index.py:117 `.half()` on queries and the single fp16 rounding of each score)
ATLAS_HD double f16_bits_to_f64(uint16_t h) {
    const uint32_t s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v;
    if (e == 0) v = (double)m * 5.9604644775390625e-08;                 // m * 2^-24
    else if (e == 31) v = m ? bits_f64(0x7ff8000000000000ull) : bits_f64(0x7ff0000000000000ull);
    else v = bits_f64(((uint64_t)(e - 15 + 1023) << 52) | ((uint64_t)m << 42));
    return s ? -v : v;
}
ATLAS_HD float f16_bits_to_f32(uint16_t h) { return (float)f16_bits_to_f64(h); }  // exact

// double -> fp16 bits, one rounding (RNE), no intermediate float (avoids double rounding)
ATLAS_HD uint16_t f64_to_f16_bits(double x) {
    const uint64_t u = f64_bits(x);
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000);
    const uint64_t a = u & 0x7fffffffffffffffull;
    if (a > 0x7ff0000000000000ull) return (uint16_t)(sign | 0x7e00);          // NaN
    const int e = (int)(a >> 52) - 1023;                                      // unbiased
    if (e >= 16) return (uint16_t)(sign | 0x7c00);                            // >= 2^16 -> inf
    if (e < -25) return sign;                                                 // < 2^-25 -> 0
    uint64_t m = (a & 0x000fffffffffffffull) | 0x0010000000000000ull;         // 53-bit
    // target: value = mant * 2^(q) with q = max(e,-14) - 10 ; shift = 52 - 10 + (sub ? -14-e : 0)
    int shift = 42 + (e < -14 ? (-14 - e) : 0);
    uint64_t keep = m >> shift;
    const uint64_t rem = m & ((1ull << shift) - 1), half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (keep & 1))) keep++;
    uint32_t out;
    if (e < -14) out = (uint32_t)keep;                    // subnormal (may carry into normal)
    else out = (uint32_t)(((uint32_t)(e + 15 - 1) << 10) + keep);   // keep has hidden bit
    if (out >= 0x7c00) out = 0x7c00;                                        // rounded up to inf
    return (uint16_t)(sign | out);
}
ATLAS_HD uint16_t f32_to_f16_bits(float x) { return f64_to_f16_bits((double)x); }  // exact widen

// Same result as f64_to_f16_bits in ~10 instructions on the GPU: double -> float with ROUND-TO-ODD
// (truncate, then OR the sticky bit into the LSB), then the hardware float -> half RNE. Round-to-odd
// into 24 bits followed by RNE into <= 11 bits equals one RNE from the exact value (24 >= 11 + 2).
ATLAS_HD uint16_t f64_to_f16_bits_rto(double x) {
    const uint64_t u = f64_bits(x);
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000);
    const double a = bits_f64(u & 0x7fffffffffffffffull);
    if (a != a) return (uint16_t)(sign | 0x7e00);
    const float f = (float)a;                        // RNE (v_cvt_f32_f64); inf above FLT_MAX
    uint32_t fb = f32_bits(f);
    const double back = (double)f;
    if (back != a) {                                 // inexact: make it truncation + sticky
        if (back > a) fb -= 1u;                      // step back towards zero (also turns +inf into FLT_MAX)
        fb |= 1u;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 h = (_Float16)bits_f32(fb);       // v_cvt_f16_f32: RNE, fp16 subnormals kept
    return (uint16_t)(sign | __builtin_bit_cast(uint16_t, h));
#else
    return (uint16_t)(sign | f64_to_f16_bits((double)bits_f32(fb)));
#endif
}
ATLAS_HD uint16_t bf16_bits_to_f16_bits(uint16_t b) {
    return f32_to_f16_bits(bits_f32((uint32_t)b << 16));
}

// ---- order-preserving integer keys ---------------------------------------------------
// fp16 bits -> 16-bit key, larger key <=> larger value; -0 == +0 (a tie, broken by row)
ATLAS_HD uint16_t f16_order_key(uint16_t h) {
    if ((h & 0x7fff) == 0) h = 0;
    return (h & 0x8000) ? (uint16_t)~h : (uint16_t)(h | 0x8000);
}
ATLAS_HD uint16_t f16_from_order_key(uint16_t k) {
    return (k & 0x8000) ? (uint16_t)(k & 0x7fff) : (uint16_t)~k;
}
// fp32 -> 32-bit key, larger key <=> larger value (used for the approximate MFMA scores)
ATLAS_HD uint32_t f32_order_key(float f) {
    const uint32_t u = f32_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
ATLAS_HD float f32_from_order_key(uint32_t k) {
    return bits_f32((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// canonical 64-bit candidate key inside one shard: (score desc, row asc) == key desc
ATLAS_HD uint64_t local_key(uint16_t h, uint32_t row) {
    return ((uint64_t)f16_order_key(h) << 32) | (uint64_t)(0xffffffffu - row);
}
// cross-shard packed candidate (atlas_hip.h): 16-bit score key, 47-bit inverted global id
#define ATLAS_GID_BITS 47
#define ATLAS_GID_MASK ((1ull << ATLAS_GID_BITS) - 1)
ATLAS_HD uint64_t pack_candidate(uint16_t h, uint64_t gid) {
    return ((uint64_t)f16_order_key(h) << ATLAS_GID_BITS) | (ATLAS_GID_MASK - (gid & ATLAS_GID_MASK));
}

// ---- canonical exact score -------------------------------------------------------------
// 64 interleaved double chains (chain j takes elements j, j+64, ... in order), combined by a fixed
// balanced tree: for m = 1,2,4,..,32: c[j] += c[j+m] for j = 0, 2m, 4m, ...  (on the GPU: one wave per
// row, lane j = chain j, xor-butterfly; IEEE addition is commutative so both give the same bits).
// fp16*fp16 is exact in double, so `c + q*p` rounds once whether or not the compiler contracts it to
// an fma: the result depends only on this order. oracle/oracle.c restates it independently.
#define ATLAS_NCHAIN 64
ATLAS_HD double exact_dot_f16(const uint16_t* q, const uint16_t* p, int d) {
    double c[ATLAS_NCHAIN];
    for (int j = 0; j < ATLAS_NCHAIN; ++j) c[j] = 0.0;
    for (int i = 0; i < d; ++i) c[i % ATLAS_NCHAIN] += f16_bits_to_f64(q[i]) * f16_bits_to_f64(p[i]);
    for (int m = 1; m < ATLAS_NCHAIN; m <<= 1)
        for (int j = 0; j < ATLAS_NCHAIN; j += 2 * m) c[j] = c[j] + c[j + m];
    return c[0];
}

// ---- certified pruning margin ---------------------------------------------------------
// The MFMA scan produces s~ with |s~ - s| <= eps (eps = GAMMA*|q|*pmax).
// Let T be the k-th largest s~ over any set of rows already seen. Then k rows have exact
// score >= T-eps, so the final k-th best fp16 score is >= RNE16(T-eps) >= T - eps - u/2,
// where u bounds the fp16 spacing around T. A row i with
//         s~_i  <=  theta = clamp(T) - 2*eps - 2*u
// has fp16 score RNE16(s_i) <= RNE16(s~_i + eps) < RNE16(T - eps) (strictly: the 2*u covers
// the half-ulp of each rounding plus the doubled grid spacing just below a negative power of
// two), so it cannot enter the canonical top-k and may be dropped. Rows above theta are kept
// and rescored exactly.  u = fp16 spacing of the binade of |clamp(T)| + 2*eps.
#define ATLAS_GAMMA 3.0517578125e-05f   /* 2^-15 */
#define ATLAS_F16_MAX_ROUND 65520.0f    /* values >= this round to fp16 inf */
ATLAS_HD float ulp16_at(float a) {      // fp16 spacing of the binade containing a (a >= 0)
    const uint32_t e = (f32_bits(a) >> 23) & 0xff;     // biased fp32 exponent
    int ex = (int)e - 127;
    if (ex < -14) ex = -14;
    if (ex > 15) ex = 15;
    return bits_f32((uint32_t)(ex - 10 + 127) << 23);
}
ATLAS_HD float prune_threshold(float T, float eps) {
    // k-th best may itself round to -inf (or T is NaN): every row ties with it, nothing is prunable
    if (!(T - eps - 0.0625f > -ATLAS_F16_MAX_ROUND)) return bits_f32(0xff800000u);
    if (T > ATLAS_F16_MAX_ROUND) T = ATLAS_F16_MAX_ROUND;            // everything above ties at +inf
    const float a = (T < 0 ? -T : T) + 2.0f * eps;
    // the last term absorbs the fp32 rounding of this expression itself
    return (T - 2.0f * eps - 2.0f * ulp16_at(a)) - 4.0f * (a * 1.1920929e-07f);
}


// ---- GELU (exact-erf form, modeling_bert.py:448-451 via ACT2FN["gelu"]) for the 16-bit encoder paths ----
// gelu(v) = 0.5 v (1 + erf(v / sqrt 2)). With z = |v| / sqrt 2 and e = erfc(z) = 2^P(z):
//     v >= 0: gelu = v - h,   v < 0: gelu = h,   h = 0.5 v e
// which has no cancellation for negative v (the textbook form computes 1 + erf(..) ~ 1e-5 from two O(1) numbers in fp32).
// P is a degree-8 polynomial with P(0) = 0, a weighted minimax fit of log2(erfc) on [0, 4] whose leading coefficient is
// negative, so beyond the fit range e underflows to 0 like erfc does. |erf error| <= 1.1e-7 in fp32 arithmetic (the fp32
// erff behind torch's GELU is of the same order). Over all 63 488 finite fp16 inputs the fp16-rounded result differs from
// the correctly rounded GELU on 187 inputs by 1 ulp; the reference formula with an ideal fp32 erf on 331 inputs by up to
// 2 ulp (tests/test_host_helpers.py::test_gelu_poly). One v_exp_f32 + 12 VALU instead of libm erff (~30).
ATLAS_HD float gelu_erf_poly(float v) {
    const float z = fabsf(v) * 0.70710678118654752f;
    float p = -4.536094274953939e-05f;
    p = p * z + 0.0004455238813534379f;
    p = p * z + -0.0014894854975864291f;
    p = p * z + -0.0007745709153823555f;
    p = p * z + 0.028253639116883278f;
    p = p * z + -0.1484816074371338f;
    p = p * z + -0.9184163808822632f;
    p = p * z + -1.6279085874557495f;
    p = p * z;
#if defined(__HIP_DEVICE_COMPILE__)
    const float e = __builtin_amdgcn_exp2f(p);
#else
    const float e = exp2f(p);
#endif
    const float h = (0.5f * v) * e;
    return (v >= 0.0f) ? v - h : h;     // NaN -> h = NaN
}

#if defined(__HIPCC__)
// The same function on two values at once with the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): the same
// operations in the same order per element, so the results are bit-identical to gelu_erf_poly on the device (where the Horner steps are
// fused multiply-adds in both). The select is replaced by max(v, 0) - |h| (identical: v - h for v >= 0, 0 - |h| = h for v < 0).
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f2 gelu_erf_poly2(const gelu_f2 v) {
    const gelu_f2 z = {fabsf(v.x) * 0.70710678118654752f, fabsf(v.y) * 0.70710678118654752f};
    gelu_f2 p = {-4.536094274953939e-05f, -4.536094274953939e-05f};
    p = __builtin_elementwise_fma(p, z, (gelu_f2){0.0004455238813534379f, 0.0004455238813534379f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){-0.0014894854975864291f, -0.0014894854975864291f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){-0.0007745709153823555f, -0.0007745709153823555f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){0.028253639116883278f, 0.028253639116883278f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){-0.1484816074371338f, -0.1484816074371338f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){-0.9184163808822632f, -0.9184163808822632f});
    p = __builtin_elementwise_fma(p, z, (gelu_f2){-1.6279085874557495f, -1.6279085874557495f});
    p = p * z;
#if defined(__HIP_DEVICE_COMPILE__)
    const gelu_f2 e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
#else
    const gelu_f2 e = {exp2f(p.x), exp2f(p.y)};     // (host pass of hipcc only; never called there)
#endif
    const gelu_f2 a = {0.5f * fabsf(v.x), 0.5f * fabsf(v.y)};
    const gelu_f2 habs = a * e;
    const gelu_f2 relu = {__builtin_fmaxf(v.x, 0.0f), __builtin_fmaxf(v.y, 0.0f)};
    return relu - habs;
}
// The same function on FOUR pairs at once, breadth first: every step of the four independent chains is issued before the next step of any of
// them (scheduling barriers keep hipcc from going depth first). Written pair by pair, hipcc emitted each pair's Horner scheme as seven
// back-to-back DEPENDENT v_pk_fma_f32 on one register pair -- an instruction-level parallelism of one in the epilogue of the largest GEMM of a
// layer (FFN-1: 64 pairs per wave and tile). Same operations in the same order per element: bit-identical to gelu_erf_poly2 / gelu_erf_poly.
__device__ __forceinline__ void gelu_erf_poly2x4(const gelu_f2 (&v)[4], gelu_f2 (&out)[4]) {
    gelu_f2 z[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        z[i] = (gelu_f2){fabsf(v[i].x) * 0.70710678118654752f, fabsf(v[i].y) * 0.70710678118654752f};
        p[i] = __builtin_elementwise_fma((gelu_f2){-4.536094274953939e-05f, -4.536094274953939e-05f}, z[i], (gelu_f2){0.0004455238813534379f, 0.0004455238813534379f});
    }
    constexpr float c[6] = {-0.0014894854975864291f, -0.0007745709153823555f, 0.028253639116883278f, -0.1484816074371338f, -0.9184163808822632f,
                            -1.6279085874557495f};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], z[i], (gelu_f2){c[s], c[s]});
    }
    __builtin_amdgcn_sched_barrier(0);
    gelu_f2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = p[i] * z[i];
        e[i] = (gelu_f2){__builtin_amdgcn_exp2f(p[i].x), __builtin_amdgcn_exp2f(p[i].y)};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const gelu_f2 a = {0.5f * fabsf(v[i].x), 0.5f * fabsf(v[i].y)};
        const gelu_f2 habs = a * e[i];
        const gelu_f2 relu = {__builtin_fmaxf(v[i].x, 0.0f), __builtin_fmaxf(v[i].y, 0.0f)};
        out[i] = relu - habs;
    }
}
#endif

}  // namespace atlas
