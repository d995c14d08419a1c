// K8B3: the activation layout of the fp32-equivalent split-bf16 GEMM path (conv_bf3.hip).
//
//   tensor [B][C][T] (C % 8 == 0)  ->  [B][C/8][3][T+2][8] bf16
//   channel c = 8q + j, frame t, plane p  ->  ((((b*(C/8) + q)*3 + p)*(T+2) + (t+1))*8 + j      (bf16 elements)
//
// * Every fp32 value v is stored as three bf16 terms v = v1 + v2 + v3 (v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2),
//   round-to-nearest-even).  Three 8-bit significands cover the 24 bits of an fp32 significand, and each residual is exactly
//   representable, so the triple is LOSSLESS (for |v| >= 2^-110; smaller values lose low bits to bf16's exponent range): a reader
//   gets the fp32 value back bit for bit as (v1 + v2) + v3 (a -0.0 returns as +0.0).  6 bytes per element instead of 4.
// * v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive k of A / B: lanes 0-31 hold k 0-7, lanes 32-63 k 8-15.  One 16-byte
//   entry (8 channels of one plane at one frame) is exactly that fragment, so an LDS copy of a row segment serves ds_read_b128
//   operand loads with no transpose and no VALU -- the same property K4P has for the fp32 MFMA (k4p.h).
// * frames -1 and T of every row are zero (the k3 convolutions' padding); channel concatenation = concatenation of 8-channel blocks.
// * A product a*b of two fp32 values is evaluated as the six bf16 products of magnitude >= 2^-24 |ab|:
//       a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2),   each exact in the MFMA's fp32 datapath, accumulated in fp32;
//   the three dropped terms (a2b3, a3b2, a3b3) are below 2^-23 |ab|.
//
// Second format of the same family, FMT_F16X2 ("K8H2"): two fp16 terms per value, [B][C/8][2][T+2][8], 4 bytes per element (what fp32
// costs).  v1 = fp16(v), v2 = fp16(v - v1): 11 + 11 significand bits, i.e. |v - (v1 + v2)| <= 2^-22 |v| -- NOT lossless: a quarter of
// the operand bits of the bf16 triple are traded for a third fewer bytes and half the matrix instructions (a1b1 + a1b2 + a2b1
// [+ a2b2]).  fp16's exponent range is the catch: |v| must stay below 65504, and below 2^-3 the second term is subnormal, so the
// absolute error has a floor of 2^-25.  Weights are therefore stored times a per-layer power of two that puts their largest value near
// 2^14 (undone exactly in the epilogue); activations are stored as they are (unit scale).  Probe only so far (tools/split_f16_probe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lds {

enum { FMT_BF16X3 = 0, FMT_F16X2 = 1 };
__host__ __device__ constexpr int fmt_planes(int fmt) { return fmt == FMT_F16X2 ? 2 : 3; }

__host__ __device__ inline long long k8b3_index(int C, int T, int b, int c, int t, int plane) {
    return ((((long long)b * (C >> 3) + (c >> 3)) * 3 + plane) * (T + 2) + (t + 1)) * 8 + (c & 7);
}
// size of a K8B3 tensor in floats of workspace (6 bytes per element); split_floats: of either format (2 bytes per plane and element)
__host__ __device__ inline size_t k8b3_floats(int C, int T) { return ((size_t)C * (T + 2) * 3 + 1) / 2; }
__host__ __device__ inline size_t split_floats(int fmt, int C, int T) { return ((size_t)C * (T + 2) * fmt_planes(fmt) + 1) / 2; }

// host: fp32 -> bf16 bits, round to nearest even (finite inputs)
inline uint16_t bf16_rn_host(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)u;
}
inline float bf16_to_float_host(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
inline void split3_host(float v, uint16_t out[3]) {
    out[0] = bf16_rn_host(v);
    const float r1 = v - bf16_to_float_host(out[0]);
    out[1] = bf16_rn_host(r1);
    const float r2 = r1 - bf16_to_float_host(out[1]);
    out[2] = bf16_rn_host(r2);
}

// host: fp32 -> fp16 bits, round to nearest even (overflow -> infinity, subnormals kept)
inline uint16_t f16_rn_host(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
inline float f16_to_float_host(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
inline void split2h_host(float v, uint16_t out[2]) {
    out[0] = f16_rn_host(v);
    out[1] = f16_rn_host(v - f16_to_float_host(out[0]));
}

#if defined(__HIPCC__)
typedef __bf16 k8_bf16x2 __attribute__((ext_vector_type(2)));
typedef float k8_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int k8_u32x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> their three bf16 planes, each packed (lo = a, hi = b): v_cvt_pk_bf16_f32 + shift / mask + subtract
static __device__ __forceinline__ void k8_split_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(k8_f32x2{a, b}, k8_bf16x2));
    const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(k8_f32x2{ra, rb}, k8_bf16x2));
    const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(k8_f32x2{sa, sb}, k8_bf16x2));
}
// the inverse: (v1 + v2) + v3, exact
static __device__ __forceinline__ void k8_join_pair(unsigned p1, unsigned p2, unsigned p3, float& a, float& b) {
    a = (__builtin_bit_cast(float, p1 << 16) + __builtin_bit_cast(float, p2 << 16)) + __builtin_bit_cast(float, p3 << 16);
    b = (__builtin_bit_cast(float, p1 & 0xffff0000u) + __builtin_bit_cast(float, p2 & 0xffff0000u)) + __builtin_bit_cast(float, p3 & 0xffff0000u);
}
typedef _Float16 k8_f16x2 __attribute__((ext_vector_type(2)));
// FMT_F16X2: two fp32 values -> their two fp16 planes, packed (lo = a, hi = b): v_cvt_pk_f16_f32, v_cvt_f32_f16, subtract
static __device__ __forceinline__ void k8h_split_pair(float a, float b, unsigned& p1, unsigned& p2) {
    const k8_f16x2 h1 = __builtin_convertvector(k8_f32x2{a, b}, k8_f16x2);
    const k8_f32x2 f1 = __builtin_convertvector(h1, k8_f32x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(k8_f32x2{a - f1[0], b - f1[1]}, k8_f16x2));
}
static __device__ __forceinline__ void k8h_join_pair(unsigned p1, unsigned p2, float& a, float& b) {
    const k8_f32x2 f1 = __builtin_convertvector(__builtin_bit_cast(k8_f16x2, p1), k8_f32x2), f2 = __builtin_convertvector(__builtin_bit_cast(k8_f16x2, p2), k8_f32x2);
    a = f1[0] + f2[0];
    b = f1[1] + f2[1];
}
// format-generic forms: p[] has fmt_planes(FMT) entries
template <int FMT>
static __device__ __forceinline__ void sp_split_pair(float a, float b, unsigned* p) {
    if constexpr (FMT == FMT_F16X2) k8h_split_pair(a, b, p[0], p[1]);
    else k8_split_pair(a, b, p[0], p[1], p[2]);
}
template <int FMT>
static __device__ __forceinline__ void sp_join_pair(const unsigned* p, float& a, float& b) {
    if constexpr (FMT == FMT_F16X2) k8h_join_pair(p[0], p[1], a, b);
    else k8_join_pair(p[0], p[1], p[2], a, b);
}
// 8-byte write-through store (see k4p.h, k4p_store_wt)
static __device__ __forceinline__ void k8_store_wt(void* p, k8_u32x2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

}  // namespace lds
