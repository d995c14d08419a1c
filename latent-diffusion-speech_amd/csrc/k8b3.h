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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lds {

__host__ __device__ inline long long k8b3_index(int C, int T, int b, int c, int t, int plane) {
    return ((((long long)b * (C >> 3) + (c >> 3)) * 3 + plane) * (T + 2) + (t + 1)) * 8 + (c & 7);
}
// size of a K8B3 tensor in floats of workspace (6 bytes per element)
__host__ __device__ inline size_t k8b3_floats(int C, int T) { return ((size_t)C * (T + 2) * 3 + 1) / 2; }

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
// 8-byte write-through store (see k4p.h, k4p_store_wt)
static __device__ __forceinline__ void k8_store_wt(void* p, k8_u32x2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

}  // namespace lds
