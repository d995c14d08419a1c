// Flash-style self-attention over the frame axis on the exact-fp32 matrix core
// (reference Attention + AttnProcessor2_0, diffusion/unet1d/attention_processor.py:980-1052:
// softmax(Q K^T / sqrt(d)) V, 8 heads, no mask; every attention on the hot path is
// self-attention).  Layout stays channel-major: qkv [B][3C][T], head h owns rows
// h*d..h*d+d-1 of each of the three C-row slabs, so Q/K/V tiles are [d][frames] with frames
// contiguous (coalesced loads, conflict-free LDS reads).
//
// Per wave: 32 queries.  S^T = K^T-tile x Q is computed "swapped" so that each lane owns one
// query column and 16 of the 32 keys in its accumulator registers: the row softmax needs only
// register reductions plus one lane<->lane+32 exchange.  The probability tile never leaves
// registers: accumulator register r of S^T is fed directly as the B operand of the r-th
// P.V MFMA (the k index of v_mfma_f32_32x32x2 is the lane half, which is exactly how the
// accumulator rows are split), with V read from LDS at the matching key.
#include "kernels.h"

#include <math.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D, int NW>
__global__ void __launch_bounds__(NW * 64) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int T, float scale) {
    constexpr int DT = (D + 31) / 32;
    constexpr int DP = DT * 32;
    constexpr int KB = 64;
    constexpr int VP = KB + 1;
    __shared__ __attribute__((aligned(16))) float Ks[D * KB];
    __shared__ __attribute__((aligned(16))) float Vs[DP * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int tq = blockIdx.x * (NW * 32) + wave * 32 + c;
    const float* Q = qkv + ((long long)b * 3 * C + (long long)hd * D) * T;
    const float* K = Q + (long long)C * T;
    const float* V = Q + 2ll * C * T;

    float qreg[D / 2];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) qreg[i] = (tq < T) ? Q[(long long)(2 * i + h) * T + tq] : 0.f;

    for (int i = tid; i < (DP - D) * VP; i += NW * 64) Vs[D * VP + i] = 0.f;   // zero the padded head-dim rows once

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bool vec = (T & 3) == 0;

    for (int k0 = 0; k0 < T; k0 += KB) {
        __syncthreads();
        for (int idx = tid; idx < D * (KB / 4); idx += NW * 64) {
            const int d = idx / (KB / 4), j = (idx - d * (KB / 4)) * 4;
            float kv[4], vv[4];
            if (vec && k0 + j + 3 < T) {
                float4 a = *reinterpret_cast<const float4*>(K + (long long)d * T + k0 + j);
                float4 e = *reinterpret_cast<const float4*>(V + (long long)d * T + k0 + j);
                kv[0] = a.x; kv[1] = a.y; kv[2] = a.z; kv[3] = a.w;
                vv[0] = e.x; vv[1] = e.y; vv[2] = e.z; vv[3] = e.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = k0 + j + e < T;
                    kv[e] = ok ? K[(long long)d * T + k0 + j + e] : 0.f;
                    vv[e] = ok ? V[(long long)d * T + k0 + j + e] : 0.f;
                }
            }
            *reinterpret_cast<float4*>(Ks + d * KB + j) = make_float4(kv[0], kv[1], kv[2], kv[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) Vs[d * VP + j + e] = vv[e];
        }
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < KB / 32; ++kt) {
            const int kbase = k0 + kt * 32;
            if (kbase >= T) break;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < D / 2; ++i)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * i + h) * KB + kt * 32 + c], qreg[i], s, 0, 0, 0);
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                s[r] = (key < T) ? s[r] * scale : -INFINITY;
                mt = fmaxf(mt, s[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = expf(m_run - m_new);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); ls += s[r]; }
            l_run = l_run * alpha + ls;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int i = 0; i < DT; ++i)
                    o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(i * 32 + c) * VP + kk], s[r], o[i], 0, 0, 0);
            }
        }
    }
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    if (tq < T) {
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (d < D) out[((long long)b * C + (long long)hd * D + d) * T + tq] = o[i][r] / l;
            }
    }
}

template <int D>
static hipError_t launch_d(const float* qkv, float* out, int B, int C, int T, int heads, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)D);
    if (T > 64) {
        hipLaunchKernelGGL((attention_kernel<D, 4>), dim3((T + 127) / 128, heads, B), dim3(256), 0, s, qkv, out, C, T, scale);
    } else {
        hipLaunchKernelGGL((attention_kernel<D, 2>), dim3((T + 63) / 64, heads, B), dim3(128), 0, s, qkv, out, C, T, scale);
    }
    return hipGetLastError();
}

hipError_t launch_attention(const float* qkv, float* out, int B, int C, int T, int heads, hipStream_t s) {
    if (C % heads) return hipErrorInvalidValue;
    switch (C / heads) {
        case 32: return launch_d<32>(qkv, out, B, C, T, heads, s);
        case 48: return launch_d<48>(qkv, out, B, C, T, heads, s);
        case 64: return launch_d<64>(qkv, out, B, C, T, heads, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lds
