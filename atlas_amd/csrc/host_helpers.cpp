// host_helpers.cpp — exposes the host+device arithmetic helpers of common.h to CPU unit tests
// (tests/test_host_helpers.py compares each against the independent oracle/oracle.c).
#include "common.h"
using namespace atlas;
extern "C" {
double h_f16_to_f64(uint16_t h) { return f16_bits_to_f64(h); }
uint16_t h_f64_to_f16(double x) { return f64_to_f16_bits(x); }
uint16_t h_f32_to_f16(float x) { return f32_to_f16_bits(x); }
uint16_t h_f64_to_f16_rto(double x) { return f64_to_f16_bits_rto(x); }
float h_gelu_erf_poly(float v) { return gelu_erf_poly(v); }
uint16_t h_bf16_to_f16(uint16_t b) { return bf16_bits_to_f16_bits(b); }
uint16_t h_f16_order_key(uint16_t h) { return f16_order_key(h); }
uint16_t h_f16_from_order_key(uint16_t k) { return f16_from_order_key(k); }
uint32_t h_f32_order_key(float f) { return f32_order_key(f); }
float h_f32_from_order_key(uint32_t k) { return f32_from_order_key(k); }
uint64_t h_local_key(uint16_t h, uint32_t row) { return local_key(h, row); }
uint64_t h_pack_candidate(uint16_t h, uint64_t gid) { return pack_candidate(h, gid); }
double h_exact_dot(const uint16_t* q, const uint16_t* p, int d) { return exact_dot_f16(q, p, d); }
float h_ulp16_at(float a) { return ulp16_at(a); }
float h_prune_threshold(float T, float eps) { return prune_threshold(T, eps); }
float h_gamma() { return ATLAS_GAMMA; }
}
