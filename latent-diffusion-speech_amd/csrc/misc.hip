// Small kernels around the denoiser: time-embedding MLP (reference embeddings.py:24-64,157-201
// and the 22 time_emb_proj layers, resnet.py:609-612), the samplers' elementwise updates
// (reference dpm_solver_pytorch.py:433-442,547-580,796-831; uni_pc.py:547-567;
// diffusion.py:95-167), layout transposes at the module boundary, speaker-embedding gather.
#include "kernels.h"

#include <math.h>

namespace lds {

// out[b][m] = sum_k W[m][k] * g(in[b][k]) + bias[m]; one wave per output row, lanes split k,
// the weight row is streamed once (16-byte loads) and reused for every batch column.
template <int BT>
__global__ void __launch_bounds__(256) small_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ in, int in_stride, int in_mode,
                                                           const float* __restrict__ freqs, float* __restrict__ out,
                                                           int out_stride, int out_silu, int M, int K, int B) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* wr = W + (long long)m * K;
    for (int b0 = 0; b0 < B; b0 += BT) {
        float acc[BT];
#pragma unroll
        for (int i = 0; i < BT; ++i) acc[i] = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int i = 0; i < BT; ++i) {
                if (b0 + i < B) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v;
                        if (in_mode == IN_SINUSOID) {
                            // Timesteps(flip_sin_to_cos=True): [cos | sin](t * f_i)
                            const int half = K >> 1, kk = k + e;
                            const float arg = in[(long long)(b0 + i) * in_stride] * freqs[kk < half ? kk : kk - half];
                            v = (kk < half) ? cosf(arg) : sinf(arg);
                        } else {
                            v = in[(long long)(b0 + i) * in_stride + k + e];
                            if (in_mode == IN_SILU) v = v / (1.0f + expf(-v));
                        }
                        acc[i] = fmaf(wv[e], v, acc[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BT; ++i) {
            float v = acc[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && b0 + i < B) {
                v += bias ? bias[m] : 0.f;
                if (out_silu) v = v / (1.0f + expf(-v));
                out[(long long)(b0 + i) * out_stride + m] = v;
            }
        }
    }
}

hipError_t launch_small_linear(const float* W, const float* bias, const float* in, int in_stride, int in_mode,
                               const float* freqs, float* out, int out_stride, int out_silu, int M, int K, int B, hipStream_t s) {
    if (K % 4) return hipErrorInvalidValue;
    ProfScope ps(s, "small_linear", 2.0 * M * (double)K * B, 4.0 * M * (double)K);
    hipLaunchKernelGGL(small_linear_kernel<16>, dim3((M + 3) / 4), dim3(256), 0, s, W, bias, in, in_stride, in_mode, freqs, out,
                       out_stride, out_silu, M, K, B);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) ew_kernel(int op, float* __restrict__ out, const float* __restrict__ a,
                                                 const float* __restrict__ b, const float* __restrict__ c,
                                                 const float* __restrict__ d, float c0, float c1, float c2, float c3, float c4,
                                                 long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float r;
        switch (op) {
            case EW_X0: r = (a[i] - c0 * b[i]) / c1; break;
            case EW_AXPBY: r = c0 * a[i] - c1 * b[i]; break;
            case EW_DPM2: { const float bi = b[i]; r = c0 * a[i] - c1 * bi - c2 * (c3 * (bi - c[i])); } break;
            case EW_UNIPC_PRED: r = a[i] - c0 * (c1 * ((c[i] - b[i]) / c2)); break;
            case EW_UNIPC_CORR: { const float bi = b[i]; r = a[i] - c0 * (c1 * ((d[i] - bi) / c2) + c3 * (c[i] - bi)); } break;
            case EW_UNIPC_CORR1: r = a[i] - c0 * (c3 * (c[i] - b[i])); break;
            case EW_DDPM: {
                const float ai = a[i];
                float x0 = c0 * ai - c1 * b[i];
                x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                r = (c2 * x0 + c3 * ai) + c4 * c[i];
            } break;
            case EW_DDIM: r = c0 * (a[i] / c1 + c2 * b[i]); break;
            case EW_PLMS_PRED: { const float ai = a[i]; r = ai + c0 * (c1 * ai - c2 * b[i]); } break;
            case EW_LIN4: {
                float v = c0 * a[i];
                if (b) v += c1 * b[i];
                if (c) v += c2 * c[i];
                if (d) v += c3 * d[i];
                r = v / c4;
            } break;
            default: r = a[i]; break;
        }
        out[i] = r;
    }
}

hipError_t launch_ew(int op, float* out, const float* a, const float* b, const float* c, const float* d, float c0, float c1,
                     float c2, float c3, float c4, long long n, hipStream_t s) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(s, "ew", 0.0, 4.0 * 3.0 * (double)n);
    hipLaunchKernelGGL(ew_kernel, dim3((unsigned)blocks), dim3(256), 0, s, op, out, a, b, c, d, c0, c1, c2, c3, c4, n);
    return hipGetLastError();
}

// One DPM-Solver++(2M) step in one pass (the EW_X0 launch and the EW_AXPBY / EW_DPM2 launch behind it: the same expressions in the same
// order, the data prediction handed over in a register instead of through memory): m0 = (x - sigma eps) / alpha, then
// x = c4 x - c5 m0 [- c6 (c7 (m0 - m1))]
__global__ void __launch_bounds__(256) dpm_step_kernel(float* __restrict__ x, const float* __restrict__ eps, float* __restrict__ m0, const float* __restrict__ m1,
                                                       float sigma, float alpha, int second, float c4, float c5, float c6, float c7, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xi = x[i];
        const float bi = (xi - sigma * eps[i]) / alpha;
        m0[i] = bi;
        x[i] = second ? c4 * xi - c5 * bi - c6 * (c7 * (bi - m1[i])) : c4 * xi - c5 * bi;
    }
}
hipError_t launch_dpm_step(float* x, const float* eps, float* m0, const float* m1, float sigma, float alpha, int second, float c4, float c5, float c6,
                           float c7, long long n, hipStream_t s) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(s, "ew", 0.0, 4.0 * (second ? 5.0 : 4.0) * (double)n);
    hipLaunchKernelGGL(dpm_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, eps, m0, m1, sigma, alpha, second, c4, c5, c6, c7, n);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
// experiment (lds_debug_set_touch_weights): read one dword of every 64-byte line of a buffer, so that the next launch finds it in the
// memory-side cache / the touching XCDs' L2s -- the upper bound of what a weight prefetcher could give a launch
__global__ void __launch_bounds__(256) touch_lines_kernel(const float* p, long long n_lines, float* sink) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < n_lines) v = p[i * 16];
    if (v == 1.2345678e-30f && sink) sink[0] = v;      // (keeps the load)
}
hipError_t launch_touch_lines(const float* p, long long bytes, hipStream_t s) {
    const long long n_lines = bytes / 64;
    if (n_lines <= 0) return hipSuccess;
    hipLaunchKernelGGL(touch_lines_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, s, p, n_lines, (float*)nullptr);
    return hipGetLastError();
}

hipError_t launch_fill(float* p, float v, long long n, hipStream_t s) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(s, "fill", 0.0, 4.0 * (double)n);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}

// dst[i] = v.f[i], i < n <= 64: a short host list travels in the launch's kernel arguments (no pageable host-to-device copy, whose
// staging would block the calling thread and read host memory after the call returned)
__global__ void __launch_bounds__(64) set_list_kernel(float* __restrict__ dst, FloatList64 v, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.f[threadIdx.x];
}
hipError_t launch_set_list(float* dst, const float* host, int n, hipStream_t s) {
    if (n < 0 || n > 64) return hipErrorInvalidValue;
    FloatList64 v;
    for (int i = 0; i < 64; ++i) v.f[i] = i < n ? host[i] : 0.f;
    hipLaunchKernelGGL(set_list_kernel, dim3(1), dim3(64), 0, s, dst, v, n);
    return hipGetLastError();
}

// out[b][c][r] = in[b][r][c] / scale, 32x32 tiles through LDS (both sides coalesced)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C, float scale) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* ib = in + (long long)b * R * C;
    float* ob = out + (long long)b * R * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, cc = c0 + tx;
        tile[ty + i * 8][tx] = (r < R && cc < C) ? ib[(long long)r * C + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + i * 8, r = r0 + tx;
        if (cc < C && r < R) ob[(long long)cc * R + r] = (scale == 1.0f) ? tile[tx][ty + i * 8] : tile[tx][ty + i * 8] / scale;
    }
}
hipError_t launch_transpose(const float* in, float* out, int B, int R, int C, float scale, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, B), dim3(256), 0, s, in, out, R, C, scale);
    return hipGetLastError();
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int idx_off,
                                   float* __restrict__ out, int C, int nrows) {
    const int b = blockIdx.x;
    // An index outside the table (nn.Embedding raises for it, reference unit2mel.py:82) poisons the row with NaN, so the
    // error surfaces in the output instead of silently selecting another speaker; host-side ids are range-checked before
    // the call (Unit2Mel.forward)
    const long long r = idx[b] + idx_off;
    const bool ok = r >= 0 && r < nrows;
    for (int cc = threadIdx.x; cc < C; cc += blockDim.x) out[(long long)b * C + cc] = ok ? table[r * C + cc] : __builtin_nanf("");
}
hipError_t launch_gather_rows(const float* table, const int64_t* idx, int idx_off, float* out, int B, int C, int nrows, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, s, table, idx, idx_off, out, C, nrows);
    return hipGetLastError();
}

// nearest resampling of frame-major rows: out[b, i, :] = in[b, min((int)floorf(i * step), Tin - 1), :]
// (F.interpolate(mode='nearest') on [B,C,T] seen from the [B,T,C] side: reference tools/tools.py:205-214)
__global__ void resample_frames_kernel(const float* __restrict__ in, float* __restrict__ out, int Tin, int Tout, int C, float step) {
    const int i = blockIdx.x, b = blockIdx.y;
    int src = (int)floorf((float)i * step);
    if (src > Tin - 1) src = Tin - 1;
    const float* ib = in + ((long long)b * Tin + src) * C;
    float* ob = out + ((long long)b * Tout + i) * C;
    for (int cc = threadIdx.x; cc < C; cc += blockDim.x) ob[cc] = ib[cc];
}
hipError_t launch_resample_frames(const float* in, float* out, int B, int Tin, int Tout, int C, float step, hipStream_t s) {
    if (B <= 0 || Tin <= 0 || Tout <= 0 || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resample_frames_kernel, dim3(Tout, B), dim3(256), 0, s, in, out, Tin, Tout, C, step);
    return hipGetLastError();
}

// F.interpolate(mode='nearest', size=Tout): src = min(int(floor(dst * (float)Tin/Tout)), Tin-1)
__global__ void __launch_bounds__(256) resample_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int Tin, int Tout, long long rows) {
    const float sc = (float)Tin / (float)Tout;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * Tout; i += (long long)gridDim.x * 256) {
        const long long r = i / Tout;
        const int t = (int)(i - r * Tout);
        int src = (int)floorf((float)t * sc);
        if (src > Tin - 1) src = Tin - 1;
        out[i] = in[r * Tin + src];
    }
}
hipError_t launch_resample_nearest(const float* in, float* out, int B, int C, int Tin, int Tout, hipStream_t s) {
    const long long rows = (long long)B * C;
    long long blocks = (rows * Tout + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(resample_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, Tin, Tout, rows);
    return hipGetLastError();
}

}  // namespace lds
