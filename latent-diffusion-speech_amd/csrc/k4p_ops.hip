// Memory-bound companions of conv_dma over K4P activations (k4p.h): layout conversion at the UNet boundary,
// GroupNorm(+scale/shift)(+SiLU) and LayerNorm materialised ONCE per tensor (reference nn.GroupNorm / SiLU in
// resnet.py:591-641, transformer_1d.py:134,262; nn.LayerNorm in attention.py:83,102,118), nearest resampling.
// An 8-channel block of a K4P tensor is one contiguous run of 2*(T+2)*4 floats, so all of these stream 16-byte
// entries with unit stride; the GroupNorm statistics are per-block partials (one workgroup per block) combined
// with Chan's formula, never atomics, so results do not depend on scheduling.
#include "k4p.h"
#include "kernels.h"

#include <math.h>
#include <stdlib.h>

namespace lds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NT>
static __device__ __forceinline__ float bsum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// one workgroup per (b, 8-channel block)
__global__ void __launch_bounds__(256) to_k4p_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T, int Ctot, int c_off) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2;
    float* ob = out + (((long long)b * (Ctot >> 3) + (c_off >> 3) + q) * 2) * Tp * 4;
    const float* ib = in + ((long long)b * C + q * 8) * T;
    for (int idx = threadIdx.x; idx < 2 * Tp; idx += 256) {
        const int hh = idx / Tp, e = idx - hh * Tp, t = e - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ib[(long long)(2 * j + hh) * T + t];
        }
        *reinterpret_cast<f32x4*>(ob + (long long)idx * 4) = v;
    }
}
hipError_t launch_to_k4p(const float* in, float* out, int B, int C, int T, int Ctot, int c_off, hipStream_t s) {
    if ((C & 7) || (Ctot & 7) || (c_off & 7)) return hipErrorInvalidValue;
    ProfScope ps(s, "to_k4p", 0.0, 8.0 * B * (double)C * T);
    hipLaunchKernelGGL(to_k4p_kernel, dim3(C / 8, B), dim3(256), 0, s, in, out, C, T, Ctot, c_off);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) from_k4p_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2;
    const float* ib = in + (((long long)b * (C >> 3) + q) * 2) * Tp * 4;
    float* ob = out + ((long long)b * C + q * 8) * T;
    for (int idx = threadIdx.x; idx < 2 * T; idx += 256) {
        const int hh = idx / T, t = idx - hh * T;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ib + ((long long)hh * Tp + t + 1) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[(long long)(2 * j + hh) * T + t] = v[j];
    }
}
hipError_t launch_from_k4p(const float* in, float* out, int B, int C, int T, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(from_k4p_kernel, dim3(C / 8, B), dim3(256), 0, s, in, out, C, T);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) plain_to_vt_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T, int D) {
    const int ch = blockIdx.x, b = blockIdx.y;
    const int T4 = (T + 3) & ~3, head = ch / D, d = ch - head * D;
    const float* ib = in + ((long long)b * C + ch) * T;
    float* ob = out + ((long long)b * C + (long long)head * D) * T4 + d * 4;
    for (int n = threadIdx.x; n < T4; n += 256) ob[(long long)(n >> 2) * D * 4 + (n & 3)] = (n < T) ? ib[n] : 0.f;
}
hipError_t launch_plain_to_vt(const float* in, float* out, int B, int C, int T, int D, hipStream_t s) {
    if (D <= 0 || C % D) return hipErrorInvalidValue;
    hipLaunchKernelGGL(plain_to_vt_kernel, dim3(C, B), dim3(256), 0, s, in, out, C, T, D);
    return hipGetLastError();
}

// ---- GroupNorm: pass 1, per 8-channel block (mean, M2) over its 8*T real elements.  One read: sums of (x - k) and
//      (x - k)^2 with k = the block's first element (a shift close to the mean removes the cancellation of the naive
//      sum / sum-of-squares form); blocks are combined in pass 2 with Chan's formula ----
__global__ void __launch_bounds__(256) gn_part_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1, int C2, int T,
                                                      float4* __restrict__ part) {
    __shared__ float red[4];
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2, nq1 = C1 >> 3;
    const float* xb = (q < nq1) ? x1 + (((long long)b * nq1 + q) * 2) * Tp * 4 : x2 + (((long long)b * (C2 >> 3) + (q - nq1)) * 2) * Tp * 4;
    const float k = xb[4];                     // element (h=0, t=0, j=0)
    float s1 = 0.f, s2 = 0.f;
    for (int idx = threadIdx.x; idx < 2 * T; idx += 256) {
        const int hh = idx / T, t = idx - hh * T;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long long)hh * Tp + t + 1) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[j] - k; s1 += d; s2 += d * d; }
    }
    s1 = bsum<256>(s1, red);
    s2 = bsum<256>(s2, red);
    const float n = 8.0f * (float)T;
    if (threadIdx.x == 0) part[(long long)b * ((C1 + C2) >> 3) + q] = make_float4(k + s1 / n, fmaxf(s2 - s1 * s1 / n, 0.f), n, 0.f);
}

// ---- GroupNorm: pass 2, combine the group's block partials, normalise + affine (+scale/shift) (+SiLU), write K4P ----
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1, int C2, int T,
                                                       int groups, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ ss, int ss_stride, int ss_off, int silu,
                                                       const float4* __restrict__ part, float* __restrict__ y) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int C = C1 + C2, Tp = T + 2, nq1 = C1 >> 3, nq = C >> 3;
    const int bpg = nq / groups;                     // 8-channel blocks per group
    const int g0 = (q / bpg) * bpg;
    float mean = 0.f, m2 = 0.f, n = 0.f;
    for (int i0 = 0; i0 < bpg; i0 += 8) {            // Chan's parallel combination, fixed order; partials fetched 8 at a time
        float4 pp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pp[e] = part[(long long)b * nq + g0 + ((i0 + e < bpg) ? i0 + e : i0)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (i0 + e < bpg) {
                const float d = pp[e].x - mean, nn = n + pp[e].z;
                mean += d * (pp[e].z / nn);
                m2 += pp[e].y + d * d * (n * pp[e].z / nn);
                n = nn;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(m2 / n + eps);
    // per-channel coefficients of this block: v = (x - mean) * a + bb
    float a[2][4], bb[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = q * 8 + 2 * j + hh;
            float aa = rstd * gamma[ci], be = beta[ci];
            if (ss) {
                const float sc = 1.0f + ss[(long long)b * ss_stride + ss_off + ci];
                const float sh = ss[(long long)b * ss_stride + ss_off + C + ci];
                aa *= sc;
                be = be * sc + sh;
            }
            a[hh][j] = aa; bb[hh][j] = be;
        }
    const float* xb = (q < nq1) ? x1 + (((long long)b * nq1 + q) * 2) * Tp * 4 : x2 + (((long long)b * (C2 >> 3) + (q - nq1)) * 2) * Tp * 4;
    float* yb = y + (((long long)b * nq + q) * 2) * Tp * 4;
    for (int idx = threadIdx.x; idx < 2 * Tp; idx += 256) {
        const int hh = idx / Tp, e = idx - hh * Tp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e >= 1 && e <= T) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (long long)idx * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = (xv[j] - mean) * a[hh][j] + bb[hh][j];
                if (silu) r = r * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r * -1.4426950408889634f));
                v[j] = r;
            }
        }
        *reinterpret_cast<f32x4*>(yb + (long long)idx * 4) = v;      // pad frames written as zeros
    }
}

// ---- GroupNorm in ONE pass over memory: one 1024-thread workgroup per (batch, group) keeps the whole group
//      ((C/groups) channels x T frames, <= MAXI 16-byte entries per thread) in registers: exact two-pass statistics
//      (mean, then sum of squared deviations) from the registers, then normalise + affine (+scale/shift) (+SiLU) and
//      write.  All of a thread's loads are issued before the first use.  Sums are combined in a fixed order. ----
template <int MAXI>
__global__ void __launch_bounds__(1024) gn_fused_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1, int C2, int T,
                                                        int groups, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ ss, int ss_stride,
                                                        int ss_off, int silu, float* __restrict__ y) {
    constexpr int NT = 1024, MAXR = 64;
    __shared__ float red[NT / 64];
    __shared__ __attribute__((aligned(16))) float cA[MAXR * 4], cB[MAXR * 4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = C1 + C2, Tp = T + 2, nq1 = C1 >> 3, nq = C >> 3;
    const int bpg = nq / groups, R = 2 * bpg;      // rows (block, hh) of Tp entries each
    const int dq = NT / Tp, dr = NT - dq * Tp;
    // per-channel affine terms of row tid>>2, element tid&3 (fetched early; finished once the statistics are known)
    float ga = 0.f, be = 0.f, sc = 1.f, sh = 0.f;
    if (tid < R * 4) {
        const int row = tid >> 2, ci = (g * bpg + (row >> 1)) * 8 + 2 * (tid & 3) + (row & 1);
        ga = gamma[ci]; be = beta[ci];
        if (ss) {
            sc = 1.0f + ss[(long long)b * ss_stride + ss_off + ci];
            sh = ss[(long long)b * ss_stride + ss_off + C + ci];
        }
    }
    f32x4 v[MAXI];
    {
        int row = tid / Tp, e = tid - row * Tp;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < R && e >= 1 && e <= T) {
                const int q = g * bpg + (row >> 1);
                const float* src = (q < nq1) ? x1 + (((long long)b * nq1 + q) * 2 + (row & 1)) * Tp * 4
                                             : x2 + (((long long)b * (C2 >> 3) + (q - nq1)) * 2 + (row & 1)) * Tp * 4;
                v[i] = *reinterpret_cast<const f32x4*>(src + (long long)e * 4);
            }
            row += dq; e += dr;
            if (e >= Tp) { e -= Tp; ++row; }
        }
    }
    const float n = (float)(C / groups) * (float)T;
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);      // pad / idle entries hold zeros
    const float mean = bsum<NT>(s1, red) / n;
    float s2 = 0.f;
    {
        int row = tid / Tp, e = tid - row * Tp;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            if (row < R && e >= 1 && e <= T) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; s2 += d * d; }
            }
            row += dq; e += dr;
            if (e >= Tp) { e -= Tp; ++row; }
        }
    }
    const float rstd = 1.0f / sqrtf(bsum<NT>(s2, red) / n + eps);
    if (tid < R * 4) {
        const float aa = rstd * ga * sc;
        cA[tid] = aa;
        cB[tid] = be * sc + sh;
    }
    __syncthreads();
    float* yb = y + ((long long)b * nq + (long long)g * bpg) * 2 * Tp * 4;      // the group's rows are contiguous in the output
    {
        int row = tid / Tp, e = tid - row * Tp, idx = tid;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            if (row < R) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};                              // pad frames written as zeros
                if (e >= 1 && e <= T) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(cA + row * 4), b4 = *reinterpret_cast<const f32x4*>(cB + row * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float r = (v[i][j] - mean) * a4[j] + b4[j];
                        if (silu) r = r * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r * -1.4426950408889634f));
                        o[j] = r;
                    }
                }
                *reinterpret_cast<f32x4*>(yb + (long long)idx * 4) = o;
            }
            row += dq; e += dr; idx += NT;
            if (e >= Tp) { e -= Tp; ++row; }
        }
    }
}

template <int MAXI>
static hipError_t launch_gn_fused(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                                  const float* beta, const float* ss, int ss_stride, int ss_off, int silu, float* y, int B, hipStream_t s) {
    hipLaunchKernelGGL(gn_fused_kernel<MAXI>, dim3(groups, B), dim3(1024), 0, s, x1, x2 ? x2 : x1, C1, C2, T, groups, eps, gamma, beta, ss,
                       ss_stride, ss_off, silu, y);
    return hipGetLastError();
}

hipError_t launch_gn_apply(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                           const float* beta, const float* ss, int ss_stride, int ss_off, int silu, float4* part, float* y, int B,
                           hipStream_t s) {
    const int C = C1 + C2;
    if ((C1 & 7) || (C2 & 7) || (C / 8) % groups) return hipErrorInvalidValue;
    static const bool two_kernels = getenv("LDS_GN_SPLIT") != nullptr;      // experiments only
    const long long entries = (long long)(C / 8 / groups) * 2 * (T + 2);
    const int need = (int)((entries + 1023) / 1024);
    if (!two_kernels && need <= 12 && 2 * (C / 8 / groups) <= 64) {          // the group fits one workgroup's registers
        ProfScope ps(s, "gn_fused", 0.0, 4.0 * 2.0 * B * (double)C * T);
        if (need <= 2) return launch_gn_fused<2>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, y, B, s);
        if (need <= 4) return launch_gn_fused<4>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, y, B, s);
        if (need <= 9) return launch_gn_fused<9>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, y, B, s);
        return launch_gn_fused<12>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, y, B, s);
    }
    {
        ProfScope ps(s, "gn_part", 0.0, 4.0 * 2.0 * B * (double)C * T);
        hipLaunchKernelGGL(gn_part_kernel, dim3(C / 8, B), dim3(256), 0, s, x1, x2 ? x2 : x1, C1, C2, T, part);
    }
    ProfScope ps(s, "gn_apply", 0.0, 4.0 * 2.0 * B * (double)C * T);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(C / 8, B), dim3(256), 0, s, x1, x2 ? x2 : x1, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride,
                       ss_off, silu, part, y);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) resample_k4p_kernel(const float* __restrict__ in, float* __restrict__ out, int Tin, int Tout) {
    const long long row = blockIdx.x;                // (b, q, hh) flattened
    const float sc = (float)Tin / (float)Tout;
    const float* ib = in + row * (Tin + 2) * 4;
    float* ob = out + row * (Tout + 2) * 4;
    for (int e = threadIdx.x; e < Tout + 2; e += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e >= 1 && e <= Tout) {
            int src = (int)floorf((float)(e - 1) * sc);
            if (src > Tin - 1) src = Tin - 1;
            v = *reinterpret_cast<const f32x4*>(ib + (long long)(src + 1) * 4);
        }
        *reinterpret_cast<f32x4*>(ob + (long long)e * 4) = v;
    }
}
hipError_t launch_resample_k4p(const float* in, float* out, int B, int C, int Tin, int Tout, hipStream_t s) {
    ProfScope ps(s, "resample", 0.0, 4.0 * B * (double)C * (Tin + Tout));
    hipLaunchKernelGGL(resample_k4p_kernel, dim3((unsigned)((long long)B * C / 4)), dim3(256), 0, s, in, out, Tin, Tout);
    return hipGetLastError();
}

}  // namespace lds
