// Normalisation statistics for the normalise-on-load path of conv_gemm.
//   GroupNorm (reference nn.GroupNorm in resnet.py:536,557, transformer_1d.py:134,
//   unet_1d_condition.py:546-548): one workgroup per (batch, group); two passes over the group
//   (mean, then centred second moment) with wavefront-shuffle + LDS reductions; the group may
//   straddle the two sources of a skip-concat.  Emits per-(b,channel) coefficients that already
//   fold gamma/beta and the resnet's time scale/shift (resnet.py:627-629).
//   LayerNorm over channels (reference attention.py:83,102,118) in [B,C,T] layout: threads run
//   along the frame axis (coalesced), channel slices are combined through LDS.
#include "kernels.h"

namespace lds {

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NT>
static __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

template <int NT>
__global__ void __launch_bounds__(NT) gn_coef_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1, int C2, int T,
                                                     long long xb1, long long xb2, int groups, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ ss, int ss_stride, int ss_off, float4* __restrict__ coef) {
    __shared__ float red[NT / 64];
    const int g = blockIdx.x, b = blockIdx.y;
    const int C = C1 + C2, cpg = C / groups;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int c_lo = g * cpg;
    auto row = [&](int ci) -> const float* {
        return (ci < C1) ? (x1 + (long long)b * xb1 + (long long)ci * T) : (x2 + (long long)b * xb2 + (long long)(ci - C1) * T);
    };
    const bool vec = (T & 3) == 0;
    float s = 0.f;
    for (int cc = w; cc < cpg; cc += NT / 64) {
        const float* r = row(c_lo + cc);
        if (vec) {
            for (int t = l * 4; t < T; t += 256) { float4 v = *reinterpret_cast<const float4*>(r + t); s += (v.x + v.y) + (v.z + v.w); }
        } else {
            for (int t = l; t < T; t += 64) s += r[t];
        }
    }
    const float n = (float)cpg * (float)T;
    const float mean = block_sum<NT>(s, red) / n;
    float q = 0.f;
    for (int cc = w; cc < cpg; cc += NT / 64) {
        const float* r = row(c_lo + cc);
        if (vec) {
            for (int t = l * 4; t < T; t += 256) {
                float4 v = *reinterpret_cast<const float4*>(r + t);
                float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        } else {
            for (int t = l; t < T; t += 64) { float a0 = r[t] - mean; q += a0 * a0; }
        }
    }
    const float var = block_sum<NT>(q, red) / n;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int cc = threadIdx.x; cc < cpg; cc += NT) {
        const int ci = c_lo + cc;
        float a = rstd * gamma[ci], bb = beta[ci];
        if (ss) {
            const float sc = 1.0f + ss[(long long)b * ss_stride + ss_off + ci];
            const float sh = ss[(long long)b * ss_stride + ss_off + C + ci];
            a *= sc;
            bb = bb * sc + sh;
        }
        coef[(long long)b * C + ci] = make_float4(mean, a, bb, 0.f);
    }
}

hipError_t launch_gn_coef(const float* x1, const float* x2, int C1, int C2, int T, long long xb1, long long xb2, int groups,
                          float eps, const float* gamma, const float* beta, const float* ss, int ss_stride, int ss_off,
                          float4* coef, int B, hipStream_t s) {
    if ((C1 + C2) % groups != 0) return hipErrorInvalidValue;
    ProfScope ps(s, "gn_coef", 0.0, 4.0 * 2.0 * B * (double)(C1 + C2) * T);
    hipLaunchKernelGGL(gn_coef_kernel<512>, dim3(groups, B), dim3(512), 0, s, x1, x2, C1, C2, T, xb1, xb2, groups, eps, gamma,
                       beta, ss, ss_stride, ss_off, coef);
    return hipGetLastError();
}

// block = 256 threads = 32 frames x 8 channel slices
__global__ void __launch_bounds__(256) ln_stats_kernel(const float* __restrict__ x, int C, int T, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float red[8][33];
    const int b = blockIdx.y;
    const int ts = threadIdx.x & 31, cs = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + ts;
    const float* xb = x + (long long)b * C * T;
    float s = 0.f;
    if (t < T)
        for (int c = cs; c < C; c += 8) s += xb[(long long)c * T + t];
    red[cs][ts] = s;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m += red[i][ts];
    m /= (float)C;
    __syncthreads();
    float q = 0.f;
    if (t < T)
        for (int c = cs; c < C; c += 8) { float a = xb[(long long)c * T + t] - m; q += a * a; }
    red[cs][ts] = q;
    __syncthreads();
    if (cs == 0 && t < T) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += red[i][ts];
        mean[(long long)b * T + t] = m;
        rstd[(long long)b * T + t] = 1.0f / sqrtf(v / (float)C + eps);
    }
}

hipError_t launch_ln_stats(const float* x, int C, int T, float eps, float* mean, float* rstd, int B, hipStream_t s) {
    ProfScope ps(s, "ln_stats", 0.0, 4.0 * 2.0 * B * (double)C * T);
    hipLaunchKernelGGL(ln_stats_kernel, dim3((T + 31) / 32, B), dim3(256), 0, s, x, C, T, eps, mean, rstd);
    return hipGetLastError();
}

}  // namespace lds
