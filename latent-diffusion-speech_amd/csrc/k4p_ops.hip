// Memory-bound companions of conv_dma over K4P activations (k4p.h): layout conversion at the UNet boundary,
// GroupNorm(+scale/shift)(+SiLU) and LayerNorm materialised ONCE per tensor (reference nn.GroupNorm / SiLU in
// resnet.py:591-641, transformer_1d.py:134,262; nn.LayerNorm in attention.py:83,102,118), nearest resampling.
// An 8-channel block of a K4P tensor is one contiguous run of 2*(T+2)*4 floats, so all of these stream 16-byte
// entries with unit stride; the GroupNorm statistics are per-block partials written by the producing convolution's
// epilogue and combined with Chan's formula in a fixed order, never atomics, so results do not depend on scheduling.
#include "k4p.h"
#include "gn_chan.h"
#include "kernels.h"

#include <hip/hip_ext.h>

#include <math.h>
#include <stdlib.h>

namespace lds {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one workgroup per (b, 8-channel block)
__global__ void __launch_bounds__(256) to_k4p_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T, int Ctot, int c_off, const int* __restrict__ lens) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2, Tv = ragged_len(lens, b, 0, T);      // (ragged batch: zeros from the utterance's length on)
    float* ob = out + (((long long)b * (Ctot >> 3) + (c_off >> 3) + q) * 2) * Tp * 4;
    const float* ib = in + ((long long)b * C + q * 8) * T;
    for (int idx = threadIdx.x; idx < 2 * Tp; idx += 256) {
        const int hh = idx / Tp, e = idx - hh * Tp, t = e - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < Tv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ib[(long long)(2 * j + hh) * T + t];
        }
        *reinterpret_cast<f32x4*>(ob + (long long)idx * 4) = v;
    }
}
hipError_t launch_to_k4p(const float* in, float* out, int B, int C, int T, int Ctot, int c_off, hipStream_t s, const int* lens) {
    if ((C & 7) || (Ctot & 7) || (c_off & 7)) return hipErrorInvalidValue;
    ProfScope ps(s, "to_k4p", 0.0, 8.0 * B * (double)C * T);
    hipLaunchKernelGGL(to_k4p_kernel, dim3(C / 8, B), dim3(256), 0, s, in, out, C, T, Ctot, c_off, lens);
    return hipGetLastError();
}

// Vocoder entry into K4P: plain [B][C][T] -> K4P rows with `pad` zero frames per side (the dilated k 7 / 11 convolutions pad up to 25
// frames), the raw tensor (residual operand) and its LeakyReLU (what the convolutions read: reference models.py:186-192 applies
// F.leaky_relu before every conv, here once per tensor).  One workgroup per (8-channel block, 1024-frame slab, batch element).
__global__ void __launch_bounds__(256) to_k4p_act_kernel(const float* __restrict__ in, float* __restrict__ raw, float* __restrict__ act,
                                                         float slope, int C, int T, int pad, const int* __restrict__ vlen) {
    const int q = blockIdx.x, t0 = blockIdx.y * 1024, b = blockIdx.z;
    const int Tv = vlen ? vlen[b] : T;      // (ragged batch: zeros from the utterance's length on)
    const int Tp = T + 2 * pad;
    const long long ro = (((long long)b * (C >> 3) + q) * 2) * Tp * 4;
    const float* ib = in + ((long long)b * C + q * 8) * T;
    for (int idx = threadIdx.x; idx < 2 * 1024; idx += 256) {
        const int hh = idx >> 10, t = t0 + (idx & 1023);
        if (t >= T) continue;
        f32x4 v, a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = (t < Tv) ? ib[(long long)(2 * j + hh) * T + t] : 0.f; a[j] = (v[j] >= 0.f) ? v[j] : v[j] * slope; }
        const long long o = ro + ((long long)hh * Tp + t + pad) * 4;
        if (raw) *reinterpret_cast<f32x4*>(raw + o) = v;
        if (act) *reinterpret_cast<f32x4*>(act + o) = a;
    }
}
hipError_t launch_to_k4p_act(const float* in, float* raw, float* act, float slope, int B, int C, int T, int pad, hipStream_t s, const int* vlen) {
    if ((C & 7) || pad < 1) return hipErrorInvalidValue;
    ProfScope ps(s, "to_k4p_act", 0.0, 4.0 * B * (double)C * T * (1.0 + (raw ? 1.0 : 0.0) + (act ? 1.0 : 0.0)));
    hipLaunchKernelGGL(to_k4p_act_kernel, dim3(C / 8, (T + 1023) / 1024, B), dim3(256), 0, s, in, raw, act, slope, C, T, pad, vlen);
    return hipGetLastError();
}

// one workgroup per (b, 8-channel block): both rows' left and right pads
__global__ void __launch_bounds__(256) k4p_zero_pads_kernel(float* __restrict__ x, int T, int pad) {
    const int Tp = T + 2 * pad;
    float* xb = x + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * Tp * 4;
    for (int idx = threadIdx.x; idx < 4 * pad; idx += 256) {
        const int hh = idx / (2 * pad), e = idx - hh * 2 * pad;
        const int entry = (e < pad) ? e : T + e;                      // left pad entries 0..pad-1, right pad entries T+pad..T+2pad-1
        *reinterpret_cast<f32x4*>(xb + ((long long)hh * Tp + entry) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
hipError_t launch_k4p_zero_pads(float* x, int B, int C, int T, int pad, hipStream_t s) {
    if ((C & 7) || pad < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k4p_zero_pads_kernel, dim3(C / 8, B), dim3(256), 0, s, x, T, pad);
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
// K4P with `pad` frames per side -> plain [B][C][T] (test entry points of the vocoder path)
__global__ void __launch_bounds__(256) from_k4p_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T, int pad) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int Tp = T + 2 * pad;
    const float* ib = in + (((long long)b * (C >> 3) + q) * 2) * Tp * 4;
    float* ob = out + ((long long)b * C + q * 8) * T;
    for (int idx = threadIdx.x; idx < 2 * T; idx += 256) {
        const int hh = idx / T, t = idx - hh * T;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ib + ((long long)hh * Tp + t + pad) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[(long long)(2 * j + hh) * T + t] = v[j];
    }
}
hipError_t launch_from_k4p_pad(const float* in, float* out, int B, int C, int T, int pad, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(from_k4p_pad_kernel, dim3(C / 8, B), dim3(256), 0, s, in, out, C, T, pad);
    return hipGetLastError();
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

// ---- GroupNorm -------------------------------------------------------------------------------------------------
// Statistics travel as (mean, M2) partials of (16 channels x 32 frames) blocks, [B][C/16][ceil(T/32)] float2.  conv_dma
// writes them from its epilogue for every tensor that feeds a GroupNorm (DmaConvArgs::gnpart_out); gn_partials_kernel
// computes the same from a K4P tensor (stand-alone entry points).  gn_stream_kernel is the only pass over the tensor.

// Wave64 reduction helpers in a fixed order (DPP row shifts + row broadcasts): the result is valid in lane 63.
static __device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_get<0x111, 0xf, true>(v);
    v += dpp_get<0x112, 0xf, true>(v);
    v += dpp_get<0x114, 0xf, true>(v);
    v += dpp_get<0x118, 0xf, true>(v);
    v += dpp_get<0x142, 0xa, false>(v);
    v += dpp_get<0x143, 0xc, false>(v);
    return v;
}
// one wave per (16-channel block, 32-frame block)
__global__ void __launch_bounds__(64) gn_partials_kernel(const float* __restrict__ x, int C, int T, float2* __restrict__ gp) {
    const int nT = (T + 31) >> 5, kb = blockIdx.x / nT, tb = blockIdx.x - kb * nT, b = blockIdx.y, lane = threadIdx.x;
    const int Tp = T + 2;
    const float* xb = x + (((long long)b * (C >> 3) + 2 * kb) * 2) * Tp * 4;      // 4 rows: (q = 2kb, 2kb+1) x (hh = 0, 1)
    const int t = tb * 32 + (lane & 31), half = lane >> 5;                          // each lane: two rows of one frame
    const bool ok = t < T;
    const float k = xb[(long long)(tb * 32 + 1) * 4];                               // element (row 0, first frame of the block, j = 0)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = half * 2 + rr;
        f32x4 v = {k, k, k, k};
        if (ok) v = *reinterpret_cast<const f32x4*>(xb + ((long long)row * Tp + t + 1) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[j] - k; s1 += d; s2 = fmaf(d, d, s2); }
    }
    s1 = wave_sum_to_lane63(s1);
    s2 = wave_sum_to_lane63(s2);
    const int nv = (T - tb * 32 < 32) ? T - tb * 32 : 32;
    if (lane == 63) {
        const float cnt = 16.0f * (float)nv, rc = 1.0f / cnt;
        gp[((long long)b * (C >> 4) + kb) * nT + tb] = make_float2(k + s1 * rc, fmaxf(s2 - s1 * s1 * rc, 0.f));
    }
}
hipError_t launch_gn_partials(const float* x, int C, int T, float2* gp, int B, hipStream_t s) {
    if (C & 15) return hipErrorInvalidValue;
    ProfScope ps(s, "gn_partials", 0.0, 4.0 * B * (double)C * T);
    hipLaunchKernelGGL(gn_partials_kernel, dim3((C / 16) * ((T + 31) / 32), B), dim3(64), 0, s, x, C, T, gp);
    return hipGetLastError();
}

// One workgroup per (batch, 8-channel block): E x 256 entries of 16 bytes are requested first, the group's statistics are
// combined from the partials while those loads are in flight (every wave does it redundantly: no LDS, no barrier), then
// normalise + affine (+scale/shift) (+SiLU) and store; the pad frames are written as zeros.
template <int E>
__global__ void __launch_bounds__(256) gn_stream_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1, int C2, int T,
                                                        int groups, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ ss, int ss_stride, int ss_off, int silu,
                                                        const float2* __restrict__ gp1, const float2* __restrict__ gp2, float* __restrict__ y,
                                                        const int* __restrict__ lens, int lvl) {
    const int q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int Tv = ragged_len(lens, b, lvl, T);      // ragged batch: statistics over, and output for, the utterance's own frames; zeros beyond
    const int C = C1 + C2, Tp = T + 2, nq1 = C1 >> 3, nq = C >> 3;
    const float* xb = (q < nq1) ? x1 + (((long long)b * nq1 + q) * 2) * Tp * 4 : x2 + (((long long)b * (C2 >> 3) + (q - nq1)) * 2) * Tp * 4;
    float* yb = y + (((long long)b * nq + q) * 2) * Tp * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, 2 * Tp * 16, 0x00020000);
    const int total = 2 * T;                          // real entries of this block: (row hh, frame t) -> idx = hh * T + t
    // ---- 1. request the first chunk ----
    f32x4 v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int idx = tid + 256 * i;
        if (idx < total) {
            const int hh = idx >= T, t = idx - hh * T;
            v[i] = *reinterpret_cast<const f32x4*>(xb + ((long long)hh * Tp + t + 1) * 4);
        }
    }
    // ---- 2. per-channel affine terms ----
    float ga[2][4], be[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = q * 8 + 2 * j + hh;
            float g_ = gamma[ci], b_ = beta[ci];
            if (ss) {
                const float sc = 1.0f + ss[(long long)b * ss_stride + ss_off + ci];
                const float sh = ss[(long long)b * ss_stride + ss_off + C + ci];
                g_ *= sc;
                b_ = b_ * sc + sh;
            }
            ga[hh][j] = g_; be[hh][j] = b_;
        }
    // ---- 3. group statistics from the partials ----
    const int cg16 = (C / groups) >> 4;               // 16-channel blocks per group
    const int g = (q * 8) / (C / groups);
    const int nT = (T + 31) >> 5, P = cg16 * nT, nk1 = C1 >> 4;
    const float inv_nT = 1.0f / (float)nT;
    // two plain sums in a fixed order (gn_chan.h gnf_wave_sum): mean = sum(n_i mean_i) / N, var = sum(M2_i + n_i (mean_i - mean)^2) / N with
    // N = 16 cg16 Tv known outright -- a dependent chain of 2 x 6 additions where the Chan tree had 6 steps of a dozen operations with a
    // reciprocal each, on the critical path of a launch that is latency-bound (DESIGN.md sections 3.10, 14.10).  The partials of the first
    // two rounds (<= 128: every shape of the UNet at T <= 1024) stay in registers between the sums; further rounds are read twice.
    auto part = [&](int p0, float& nb, float& mb, float& qb) {
        const int pi = p0 + lane;
        nb = 0.f; mb = 0.f; qb = 0.f;
        if (pi < P) {
            const int kk = (int)(((float)pi + 0.5f) * inv_nT), tb = pi - kk * nT, kb = g * cg16 + kk;      // pi / nT without the integer-division sequence
            const int nv = (Tv - tb * 32 < 32) ? Tv - tb * 32 : 32;
            if (nv > 0) {
                const float2 pr = (kb < nk1) ? gp1[((long long)b * nk1 + kb) * nT + tb] : gp2[((long long)b * (C2 >> 4) + (kb - nk1)) * nT + tb];
                nb = 16.0f * (float)nv; mb = pr.x; qb = pr.y;
            }
        }
    };
    float n0, m0, q0, n1 = 0.f, m1 = 0.f, q1 = 0.f;
    part(0, n0, m0, q0);
    if (P > 64) part(64, n1, m1, q1);
    float s1 = fmaf(n1, m1, n0 * m0);
    for (int p0 = 128; p0 < P; p0 += 64) {
        float nb, mb, qb;
        part(p0, nb, mb, qb);
        s1 = fmaf(nb, mb, s1);
    }
    const float rN = __builtin_amdgcn_rcpf(16.0f * (float)cg16 * (float)Tv);
    const float mu = gnf_wave_sum(s1) * rN;
    float d0 = m0 - mu, d1 = m1 - mu;
    float s2 = fmaf(n0 * d0, d0, q0) + fmaf(n1 * d1, d1, q1);      // (an empty partial carries n = 0, mean = 0, M2 = 0)
    for (int p0 = 128; p0 < P; p0 += 64) {
        float nb, mb, qb;
        part(p0, nb, mb, qb);
        const float d = mb - mu;
        s2 += fmaf(nb * d, d, qb);
    }
    const float var = gnf_wave_sum(s2) * rN;
    const float rstd = __builtin_amdgcn_rsqf(var + eps);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[hh][j] *= rstd;
    // ---- 4. normalise and store ----
    if (tid < 4) {                                    // the four pad entries (frames -1 and T of both rows)
        const int hh = tid >> 1, e = (tid & 1) ? T + 1 : 0;
        *reinterpret_cast<f32x4*>(yb + ((long long)hh * Tp + e) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int base = 0; base < total; base += 256 * E) {
        if (base > 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int idx = base + tid + 256 * i;
                if (idx < total) {
                    const int hh = idx >= T, t = idx - hh * T;
                    v[i] = *reinterpret_cast<const f32x4*>(xb + ((long long)hh * Tp + t + 1) * 4);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int idx = base + tid + 256 * i;
            if (idx < total) {
                const int hh = idx >= T, t = idx - hh * T;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float r = (v[i][j] - mu) * (hh ? ga[1][j] : ga[0][j]) + (hh ? be[1][j] : be[0][j]);
                    if (silu) r = r * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r * -1.4426950408889634f));
                    o[j] = (t < Tv) ? r : 0.f;
                }
                // write-through 16-byte store (sc1): no dirty lines left for the end-of-kernel write-back (k4p.h, k4p_store_wt)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, (int)(((long long)hh * Tp + t + 1) * 16), 0, 16);
            }
        }
    }
}

hipError_t launch_gn_stream(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                            const float* beta, const float* ss, int ss_stride, int ss_off, int silu, const float2* gp1, const float2* gp2,
                            float* y, int B, hipStream_t s, const int* lens, int lvl) {
    const int C = C1 + C2;
    if ((C1 & 15) || (C2 & 15) || groups <= 0 || C % groups || (C / groups) % 16 || !gp1 || (C2 && !gp2)) return hipErrorInvalidValue;
    ProfScope ps(s, "gn_stream", 0.0, 4.0 * 2.0 * B * (double)C * T, true);
    const dim3 grid(C / 8, B), blk(256);
    const int need = (2 * T + 255) / 256;
#define GN_ARGS x1, x2 ? x2 : x1, C1, C2, T, groups, eps, gamma, beta, ss, ss_stride, ss_off, silu, gp1, gp2 ? gp2 : gp1, y, lens, lvl
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) {      // bench.py's instrumented step: the events ride in the dispatch
        if (need <= 1) hipExtLaunchKernelGGL(gn_stream_kernel<1>, grid, blk, 0, s, e0, e1, 0, GN_ARGS);
        else if (need <= 2) hipExtLaunchKernelGGL(gn_stream_kernel<2>, grid, blk, 0, s, e0, e1, 0, GN_ARGS);
        else hipExtLaunchKernelGGL(gn_stream_kernel<4>, grid, blk, 0, s, e0, e1, 0, GN_ARGS);
    } else if (need <= 1) hipLaunchKernelGGL(gn_stream_kernel<1>, grid, blk, 0, s, GN_ARGS);
    else if (need <= 2) hipLaunchKernelGGL(gn_stream_kernel<2>, grid, blk, 0, s, GN_ARGS);
    else hipLaunchKernelGGL(gn_stream_kernel<4>, grid, blk, 0, s, GN_ARGS);
#undef GN_ARGS
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) resample_k4p_kernel(const float* __restrict__ in, float* __restrict__ out, int Tin, int Tout, int rows_per_b,
                                                           const int* __restrict__ lens, int lvl_in, int lvl_out) {
    const long long row = blockIdx.x;                // (b, q, hh) flattened
    const int b = (int)(row / rows_per_b);
    // ragged batch: every utterance is resampled from ITS input length to ITS output length (F.interpolate(size=) of the utterance alone)
    const int Ti = ragged_len(lens, b, lvl_in, Tin), To = ragged_len(lens, b, lvl_out, Tout);
    const float sc = (float)Ti / (float)To;
    const float* ib = in + row * (Tin + 2) * 4;
    float* ob = out + row * (Tout + 2) * 4;
    for (int e = threadIdx.x; e < Tout + 2; e += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e >= 1 && e <= To) {
            int src = (int)floorf((float)(e - 1) * sc);
            if (src > Ti - 1) src = Ti - 1;
            v = *reinterpret_cast<const f32x4*>(ib + (long long)(src + 1) * 4);
        }
        *reinterpret_cast<f32x4*>(ob + (long long)e * 4) = v;
    }
}
hipError_t launch_resample_k4p(const float* in, float* out, int B, int C, int Tin, int Tout, hipStream_t s, const int* lens, int lvl_in, int lvl_out) {
    ProfScope ps(s, "resample", 0.0, 4.0 * B * (double)C * (Tin + Tout));
    hipLaunchKernelGGL(resample_k4p_kernel, dim3((unsigned)((long long)B * C / 4)), dim3(256), 0, s, in, out, Tin, Tout, C / 4, lens, lvl_in, lvl_out);
    return hipGetLastError();
}

}  // namespace lds
