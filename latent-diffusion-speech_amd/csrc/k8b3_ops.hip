// Memory-bound companions of conv_bf3 over K8B3 activations (k8b3.h): layout conversion at the UNet boundary, GroupNorm(+scale/shift)
// (+SiLU) materialised once per tensor (reference nn.GroupNorm / SiLU in resnet.py:591-641, transformer_1d.py:134,262), nearest
// resampling.  Same structure and statistics path as k4p_ops.hip: an 8-channel block is one contiguous run of 3*(T+2) 16-byte
// entries; a thread owns one frame of the block (three 16-byte loads -> 8 fp32 values, exact -> ... -> three 16-byte stores).
#include "k4p.h"
#include "k8b3.h"
#include "kernels.h"

#include <hip/hip_ext.h>

#include <math.h>

namespace lds {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pl[] = the fmt_planes(FMT) 16-byte entries of one frame of an 8-channel block <-> its 8 fp32 values
template <int FMT>
static __device__ __forceinline__ void sp_join8(const u32x4* pl, float v[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned w[3];
#pragma unroll
        for (int q = 0; q < fmt_planes(FMT); ++q) w[q] = pl[q][i];
        sp_join_pair<FMT>(w, v[2 * i], v[2 * i + 1]);
    }
}
template <int FMT>
static __device__ __forceinline__ void sp_split8(const float v[8], u32x4* pl) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned w[3];
        sp_split_pair<FMT>(v[2 * i], v[2 * i + 1], w);
#pragma unroll
        for (int q = 0; q < fmt_planes(FMT); ++q) pl[q][i] = w[q];
    }
}
template <int FMT>
static __device__ __forceinline__ void sp_load(const char* base, int Tp, int entry, u32x4* pl) {      // base = the block's first plane row
#pragma unroll
    for (int q = 0; q < fmt_planes(FMT); ++q) pl[q] = *reinterpret_cast<const u32x4*>(base + ((long long)q * Tp + entry) * 16);
}

// one workgroup per (b, 8-channel block)
template <int FMT>
__global__ void __launch_bounds__(256) to_k8b3_kernel(const float* __restrict__ in, char* __restrict__ out, int C, int T, int Ctot, int c_off, const int* __restrict__ lens) {
    constexpr int NPL = fmt_planes(FMT);
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2, Tv = ragged_len(lens, b, 0, T);      // (ragged batch: zeros from the utterance's length on)
    char* ob = out + (((long long)b * (Ctot >> 3) + (c_off >> 3) + q) * NPL) * Tp * 16;
    const float* ib = in + ((long long)b * C + q * 8) * T;
    for (int e = threadIdx.x; e < Tp; e += 256) {
        const int t = e - 1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (t >= 0 && t < Tv) ? ib[(long long)j * T + t] : 0.f;
        u32x4 pl[NPL];
        sp_split8<FMT>(v, pl);
#pragma unroll
        for (int w = 0; w < NPL; ++w) *reinterpret_cast<u32x4*>(ob + ((long long)w * Tp + e) * 16) = pl[w];
    }
}
hipError_t launch_to_k8b3(const float* in, void* out, int B, int C, int T, int Ctot, int c_off, hipStream_t s, int fmt, const int* lens) {
    if ((C & 7) || (Ctot & 7) || (c_off & 7)) return hipErrorInvalidValue;
    ProfScope ps(s, "to_k8b3", 0.0, (4.0 + 2.0 * fmt_planes(fmt)) * B * (double)C * T);
    if (fmt == FMT_F16X2) hipLaunchKernelGGL(to_k8b3_kernel<FMT_F16X2>, dim3(C / 8, B), dim3(256), 0, s, in, (char*)out, C, T, Ctot, c_off, lens);
    else hipLaunchKernelGGL(to_k8b3_kernel<FMT_BF16X3>, dim3(C / 8, B), dim3(256), 0, s, in, (char*)out, C, T, Ctot, c_off, lens);
    return hipGetLastError();
}

template <int FMT>
__global__ void __launch_bounds__(256) from_k8b3_kernel(const char* __restrict__ in, float* __restrict__ out, int C, int T) {
    constexpr int NPL = fmt_planes(FMT);
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2;
    const char* ib = in + (((long long)b * (C >> 3) + q) * NPL) * Tp * 16;
    float* ob = out + ((long long)b * C + q * 8) * T;
    for (int t = threadIdx.x; t < T; t += 256) {
        u32x4 pl[NPL];
        sp_load<FMT>(ib, Tp, t + 1, pl);
        float v[8];
        sp_join8<FMT>(pl, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[(long long)j * T + t] = v[j];
    }
}
hipError_t launch_from_k8b3(const void* in, float* out, int B, int C, int T, hipStream_t s, int fmt) {
    if (C & 7) return hipErrorInvalidValue;
    if (fmt == FMT_F16X2) hipLaunchKernelGGL(from_k8b3_kernel<FMT_F16X2>, dim3(C / 8, B), dim3(256), 0, s, (const char*)in, out, C, T);
    else hipLaunchKernelGGL(from_k8b3_kernel<FMT_BF16X3>, dim3(C / 8, B), dim3(256), 0, s, (const char*)in, out, C, T);
    return hipGetLastError();
}

// ---- GroupNorm (statistics as (mean, M2) partials of (16 channels x 32 frames) blocks, k4p_ops.hip) --------------------------------
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ float dpp_get8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
static __device__ __forceinline__ float wave_sum_to_lane63_8(float v) {
    v += dpp_get8<0x111, 0xf, true>(v);
    v += dpp_get8<0x112, 0xf, true>(v);
    v += dpp_get8<0x114, 0xf, true>(v);
    v += dpp_get8<0x118, 0xf, true>(v);
    v += dpp_get8<0x142, 0xa, false>(v);
    v += dpp_get8<0x143, 0xc, false>(v);
    return v;
}
static __device__ __forceinline__ void chan8(float& n, float& mean, float& m2, float nb, float mb, float qb) {
    const float nn = n + nb;
    const float r = (nn > 0.f) ? __builtin_amdgcn_rcpf(nn) : 0.f;
    const float d = mb - mean;
    mean += d * (nb * r);
    m2 += qb + d * d * (n * nb * r);
    n = nn;
}
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ void chan_step8(float& n, float& mean, float& m2) {
    asm volatile("s_nop 1" : "+v"(n), "+v"(mean), "+v"(m2));      // as gn_chan.h chan_step: complete values, then the cross-lane reads
    const float nb = dpp_get8<CTRL, ROW_MASK, BOUND>(n), mb = dpp_get8<CTRL, ROW_MASK, BOUND>(mean), qb = dpp_get8<CTRL, ROW_MASK, BOUND>(m2);
    chan8(n, mean, m2, nb, mb, qb);
}

// one wave per (16-channel block, 32-frame block): lane = (frame, which 8-channel block)
template <int FMT>
__global__ void __launch_bounds__(64) gn_partials_bf3_kernel(const char* __restrict__ x, int C, int T, float2* __restrict__ gp) {
    constexpr int NPL = fmt_planes(FMT);
    const int nT = (T + 31) >> 5, kb = blockIdx.x / nT, tb = blockIdx.x - kb * nT, b = blockIdx.y, lane = threadIdx.x;
    const int Tp = T + 2;
    const int t = tb * 32 + (lane & 31), half = lane >> 5;
    const bool ok = t < T;
    const char* xb = x + (((long long)b * (C >> 3) + 2 * kb) * NPL) * Tp * 16;      // first block of the pair
    float k;
    {
        u32x4 pl[NPL];
        sp_load<FMT>(xb, Tp, tb * 32 + 1, pl);
        float v[8];
        sp_join8<FMT>(pl, v);
        k = v[0];                                       // channel 16 kb, first frame of the block
    }
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        const char* xr = xb + (long long)half * NPL * Tp * 16;
        u32x4 pl[NPL];
        sp_load<FMT>(xr, Tp, t + 1, pl);
        float v[8];
        sp_join8<FMT>(pl, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - k; s1 += d; s2 = fmaf(d, d, s2); }
    }
    s1 = wave_sum_to_lane63_8(s1);
    s2 = wave_sum_to_lane63_8(s2);
    const int nv = (T - tb * 32 < 32) ? T - tb * 32 : 32;
    if (lane == 63) {
        const float cnt = 16.0f * (float)nv, rc = 1.0f / cnt;
        gp[((long long)b * (C >> 4) + kb) * nT + tb] = make_float2(k + s1 * rc, fmaxf(s2 - s1 * s1 * rc, 0.f));
    }
}
hipError_t launch_gn_partials_bf3(const void* x, int C, int T, float2* gp, int B, hipStream_t s, int fmt) {
    if (C & 15) return hipErrorInvalidValue;
    if (fmt == FMT_F16X2) hipLaunchKernelGGL(gn_partials_bf3_kernel<FMT_F16X2>, dim3((C / 16) * ((T + 31) / 32), B), dim3(64), 0, s, (const char*)x, C, T, gp);
    else hipLaunchKernelGGL(gn_partials_bf3_kernel<FMT_BF16X3>, dim3((C / 16) * ((T + 31) / 32), B), dim3(64), 0, s, (const char*)x, C, T, gp);
    return hipGetLastError();
}

// One workgroup per (batch, 8-channel block); a thread owns E frames.  The loads are requested first, the group's statistics are
// combined from the partials while they are in flight (every wave redundantly: no LDS, no barrier), then normalise + affine
// (+scale/shift) (+SiLU), split into the three planes and store; the pad frames are written as zeros.
template <int E, int FMT>
__global__ void __launch_bounds__(256) gn_stream_bf3_kernel(const char* __restrict__ x1, const char* __restrict__ x2, int C1, int C2, int T, int groups,
                                                            float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ ss, int ss_stride, int ss_off, int silu,
                                                            const float2* __restrict__ gp1, const float2* __restrict__ gp2, char* __restrict__ y,
                                                            const int* __restrict__ lens, int lvl) {
    constexpr int NPL = fmt_planes(FMT);
    const int q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int Tv = ragged_len(lens, b, lvl, T);      // ragged batch: statistics over, and output for, the utterance's own frames; zeros beyond
    const int C = C1 + C2, Tp = T + 2, nq1 = C1 >> 3, nq = C >> 3;
    const char* xb = (q < nq1) ? x1 + (((long long)b * nq1 + q) * NPL) * Tp * 16 : x2 + (((long long)b * (C2 >> 3) + (q - nq1)) * NPL) * Tp * 16;
    char* yb = y + (((long long)b * nq + q) * NPL) * Tp * 16;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, NPL * Tp * 16, 0x00020000);
    // ---- 1. request the first chunk ----
    u32x4 vin[E][NPL];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int t = tid + 256 * i;
        if (t < T) sp_load<FMT>(xb, Tp, t + 1, vin[i]);
    }
    // ---- 2. per-channel affine terms ----
    float ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = q * 8 + j;
        float g_ = gamma[ci], b_ = beta[ci];
        if (ss) {
            const float sc = 1.0f + ss[(long long)b * ss_stride + ss_off + ci];
            const float sh = ss[(long long)b * ss_stride + ss_off + C + ci];
            g_ *= sc;
            b_ = b_ * sc + sh;
        }
        ga[j] = g_; be[j] = b_;
    }
    // ---- 3. group statistics from the partials (k4p_ops.hip gn_stream_kernel, same order) ----
    const int cg16 = (C / groups) >> 4;
    const int g = (q * 8) / (C / groups);
    const int nT = (T + 31) >> 5, P = cg16 * nT, nk1 = C1 >> 4;
    const float inv_nT = 1.0f / (float)nT;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int pi = p0 + lane;
        float nb = 0.f, mb = 0.f, qb = 0.f;
        if (pi < P) {
            const int kk = (int)(((float)pi + 0.5f) * inv_nT), tb = pi - kk * nT, kb = g * cg16 + kk;
            const int nv = (Tv - tb * 32 < 32) ? Tv - tb * 32 : 32;
            if (nv > 0) {
                const float2 pr = (kb < nk1) ? gp1[((long long)b * nk1 + kb) * nT + tb] : gp2[((long long)b * (C2 >> 4) + (kb - nk1)) * nT + tb];
                nb = 16.0f * (float)nv; mb = pr.x; qb = pr.y;
            }
        }
        chan8(n, mean, m2, nb, mb, qb);
    }
    chan_step8<0x111, 0xf, true>(n, mean, m2);
    chan_step8<0x112, 0xf, true>(n, mean, m2);
    chan_step8<0x114, 0xf, true>(n, mean, m2);
    chan_step8<0x118, 0xf, true>(n, mean, m2);
    chan_step8<0x142, 0xa, false>(n, mean, m2);
    chan_step8<0x143, 0xc, false>(n, mean, m2);
    const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mean), 63));
    const float var = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m2), 63)) /
                      __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n), 63));
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] *= rstd;
    // ---- 4. normalise and store ----
    if (tid < 2 * NPL) {                              // the pad entries (frames -1 and T of every plane)
        const int pl = tid >> 1, e = (tid & 1) ? T + 1 : 0;
        *reinterpret_cast<u32x4*>(yb + ((long long)pl * Tp + e) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int base = 0; base < T; base += 256 * E) {
        if (base > 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int t = base + tid + 256 * i;
                if (t < T) sp_load<FMT>(xb, Tp, t + 1, vin[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int t = base + tid + 256 * i;
            if (t < T) {
                float v[8];
                sp_join8<FMT>(vin[i], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float r = (v[j] - mu) * ga[j] + be[j];
                    if (silu) r = r * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r * -1.4426950408889634f));
                    v[j] = (t < Tv) ? r : 0.f;
                }
                u32x4 o[NPL];
                sp_split8<FMT>(v, o);
                // write-through 16-byte stores (sc1): no dirty lines left for the end-of-kernel write-back
#pragma unroll
                for (int w = 0; w < NPL; ++w) __builtin_amdgcn_raw_buffer_store_b128(o[w], ry, (int)(((long long)w * Tp + t + 1) * 16), 0, 16);
            }
        }
    }
}

template <int FMT>
static hipError_t gn_stream_launch(const void* x1, const void* x2, int C1, int C2, int T, int groups, float eps, const float* gamma, const float* beta,
                                   const float* ss, int ss_stride, int ss_off, int silu, const float2* gp1, const float2* gp2, void* y, int B, hipStream_t s,
                                   const int* lens, int lvl) {
    const int C = C1 + C2;
    ProfScope ps(s, "gn_stream_bf3", 0.0, 2.0 * fmt_planes(FMT) * 2.0 * B * (double)C * T, true);
    const dim3 grid(C / 8, B), blk(256);
    const int need = (T + 255) / 256;
#define GN_ARGS (const char*)x1, (const char*)(x2 ? x2 : x1), C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, gp1, gp2 ? gp2 : gp1, (char*)y, lens, lvl
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) {
        if (need <= 1) hipExtLaunchKernelGGL((gn_stream_bf3_kernel<1, FMT>), grid, blk, 0, s, e0, e1, 0, GN_ARGS);
        else hipExtLaunchKernelGGL((gn_stream_bf3_kernel<2, FMT>), grid, blk, 0, s, e0, e1, 0, GN_ARGS);
    } else if (need <= 1) hipLaunchKernelGGL((gn_stream_bf3_kernel<1, FMT>), grid, blk, 0, s, GN_ARGS);
    else hipLaunchKernelGGL((gn_stream_bf3_kernel<2, FMT>), grid, blk, 0, s, GN_ARGS);
#undef GN_ARGS
    return hipGetLastError();
}
hipError_t launch_gn_stream_bf3(const void* x1, const void* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                                const float* beta, const float* ss, int ss_stride, int ss_off, int silu, const float2* gp1, const float2* gp2,
                                void* y, int B, hipStream_t s, int fmt, const int* lens, int lvl) {
    const int C = C1 + C2;
    if ((C1 & 15) || (C2 & 15) || groups <= 0 || C % groups || (C / groups) % 16 || !gp1 || (C2 && !gp2)) return hipErrorInvalidValue;
    return fmt == FMT_F16X2 ? gn_stream_launch<FMT_F16X2>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, gp1, gp2, y, B, s, lens, lvl)
                            : gn_stream_launch<FMT_BF16X3>(x1, x2, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, gp1, gp2, y, B, s, lens, lvl);
}

// nearest-neighbour resample along frames (K8B3 -> K8B3), reference F.interpolate(size=Tout); one workgroup per (b, block, plane) row
__global__ void __launch_bounds__(256) resample_k8b3_kernel(const char* __restrict__ in, char* __restrict__ out, int Tin_, int Tout_, int rows_per_b,
                                                            const int* __restrict__ lens, int lvl_in, int lvl_out) {
    const long long row = blockIdx.x;
    const int b = (int)(row / rows_per_b);
    const int Tin = ragged_len(lens, b, lvl_in, Tin_), Tout = ragged_len(lens, b, lvl_out, Tout_);      // ragged batch: the utterance's own lengths
    const float sc = (float)Tin / (float)Tout;
    const char* ib = in + row * (Tin_ + 2) * 16;
    char* ob = out + row * (Tout_ + 2) * 16;
    for (int e = threadIdx.x; e < Tout_ + 2; e += 256) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (e >= 1 && e <= Tout) {
            int src = (int)floorf((float)(e - 1) * sc);
            if (src > Tin - 1) src = Tin - 1;
            v = *reinterpret_cast<const u32x4*>(ib + (long long)(src + 1) * 16);
        }
        *reinterpret_cast<u32x4*>(ob + (long long)e * 16) = v;
    }
}
hipError_t launch_resample_k8b3(const void* in, void* out, int B, int C, int Tin, int Tout, hipStream_t s, int fmt, const int* lens, int lvl_in, int lvl_out) {
    ProfScope ps(s, "resample", 0.0, 2.0 * fmt_planes(fmt) * B * (double)C * (Tin + Tout));
    hipLaunchKernelGGL(resample_k8b3_kernel, dim3((unsigned)((long long)B * (C / 8) * fmt_planes(fmt))), dim3(256), 0, s, (const char*)in, (char*)out, Tin, Tout,
                       (C / 8) * fmt_planes(fmt), lens, lvl_in, lvl_out);
    return hipGetLastError();
}

}  // namespace lds
