// Narrow-channel 1-D convolution for the vocoder's tail (reference encoder/hifi_vaegan/modules/models.py:161-222 at the
// 16-channel stage: 262,144 samples per utterance; the kernel is written for 16 or 32 channels, 16 is what is dispatched).  These layers move 0.27 GB per tensor and batch and carry
// only 18-60 FLOP per byte, so they sit at the HBM / fp32-MFMA balance point; the 32x32 tiles of the generic kernels would waste
// half of the matrix pipe on 16 output channels and re-stage the weights for every tile.  Here
//   * the MFMA is v_mfma_f32_16x16x4_f32 (exact fp32, M = 16 output channels = one tile, no padding rows);
//   * a wave keeps its whole weight slice (16 output channels x C x KT taps = KT*C/4 A-operand registers) in VGPRs for the
//     lifetime of the workgroup, which walks over many frame blocks;
//   * per frame block the input window (C rows x (TB + halo) frames) is staged once into LDS with 16-byte coalesced loads along
//     the frame axis, LeakyReLU applied while staging (once per element, not once per tap); every tap reads the same window at a
//     shifted immediate offset (one ds_read_b32 per MFMA, rows padded so lane groups hit disjoint banks);
//   * bias / residual / running-sum (MRF) / division epilogue on 64-byte row segments of the plain [B][C][T] tensors.
// Per-element summation order: k = (tap, input channel) ascending, one accumulator chain -- independent of batch and tiling.
#include "kernels.h"

#include <stdio.h>

namespace lds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C, int KT, int DIL>
struct SmallCfg {
    static constexpr int TB = 512;                       // frames per block
    static constexpr int HALO = (KT - 1) * DIL;          // window = TB + HALO frames (+ up to 3 of alignment slack)
    static constexpr int WP = TB + 64 + 16;              // LDS row stride in floats: >= TB + HALO + 3, and = 16 (mod 32) so the 4 k-rows of one
                                                         // ds_read_b32 (lanes 0-15 / 16-31 / ...) fall on disjoint bank groups
    static constexpr int NCT = C / 16;                   // output-channel tiles
    static constexpr int NA = KT * C / 4;                // MFMAs (= A registers) per 16 x 16 output tile
    static constexpr int TPW = TB / (4 / NCT);           // frames per wave: waves = NCT channel tiles x (4 / NCT) frame slices
    static constexpr size_t LDS_BYTES = (size_t)C * WP * sizeof(float);
    static_assert(HALO + 3 <= 64 + 16, "window does not fit the padded row");
    static_assert(C == 16 || C == 32, "tail widths");
};

template <int C, int KT, int DIL>
__global__ void __launch_bounds__(256) conv_small_kernel(const ConvArgs p, int n_blocks) {
    using Cfg = SmallCfg<C, KT, DIL>;
    constexpr int TB = Cfg::TB, WP = Cfg::WP, NA = Cfg::NA, NCT = Cfg::NCT, TPW = Cfg::TPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [C][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int cot = wave % NCT, fsl = wave / NCT;         // this wave's output-channel tile and frame slice
    const int T = p.Tsrc;

    // ---- weights of this wave's 16 output channels: A operand of MFMA (tap, kb) = W[co = 16 cot + (l & 15)][ci = 4 kb + (l >> 4)][tap],
    //      read once from the packed layout [tap][Ci/8][2][Mp][4] (model.hip widx) ----
    float wa[NA];
    {
        const int co = cot * 16 + l15;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int tap = i / (C / 4), kb = i % (C / 4);
            const int ci = 4 * kb + l4;
            wa[i] = p.w[((((long long)tap * (C / 8) + (ci >> 3)) * 2 + (ci & 1)) * p.Mp + co) * 4 + ((ci & 7) >> 1)];
        }
    }
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = p.bias ? p.bias[cot * 16 + 4 * l4 + r] : 0.f;

    const bool vec = (T & 3) == 0;
    constexpr int W4 = (TB + Cfg::HALO + 3 + 3) / 4;            // 16-byte chunks per window row
    constexpr int NCH = (C * W4 + 255) / 256;                   // chunks per thread
    const int nb = (T + TB - 1) / TB;
    // window chunk `i` of this thread for block `blk`: all of a block's loads are issued together (no dependent round trips),
    // and the next block's are issued before the current block's MFMAs so their latency hides behind the matrix work
    f32x4 xr[NCH];
    auto fetch = [&](int blk) {
        const int b = blk / nb, t0 = (blk - b * nb) * TB;
        const float* xb = p.x1 + (long long)b * C * T;
        const int s0 = t0 - p.pad;
        const int s_al = (s0 >= 0) ? (s0 & ~3) : -(((-s0) + 3) & ~3);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + 256 * i;
            const int ci = q / W4, c4 = q - ci * W4;
            const int s = s_al + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < C * W4) {
                const int Tv = p.vlen_in ? (p.vlen_in[b] < T ? p.vlen_in[b] : T) : T;      // (ragged batch: the input reads as zeros beyond its length)
                if (vec && s >= 0 && s + 3 < Tv) {
                    v = *reinterpret_cast<const f32x4*>(xb + (long long)ci * T + s);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (s + e >= 0 && s + e < Tv) ? xb[(long long)ci * T + s + e] : 0.f;
                }
            }
            xr[i] = v;
        }
    };
    if (blockIdx.x < n_blocks) fetch(blockIdx.x);
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int b = blk / nb, t0 = (blk - b * nb) * TB;
        const int s0 = t0 - p.pad;                              // first frame the taps touch
        const int s_al = (s0 >= 0) ? (s0 & ~3) : -(((-s0) + 3) & ~3);
        const int off = s0 - s_al;                              // 0..3
        // ---- commit the window: rows = input channels; LeakyReLU once per element; zeros outside [0, T) ----
        __syncthreads();                                        // the previous block's reads are done
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + 256 * i;
            const int ci = q / W4, c4 = q - ci * W4;
            f32x4 v = xr[i];
            if (p.act_in == ACT_LRELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] >= 0.f) ? v[e] : v[e] * p.slope;
            }
            if (q < C * W4) *reinterpret_cast<f32x4*>(smem + ci * WP + 4 * c4) = v;
        }
        __syncthreads();
        if (blk + (int)gridDim.x < n_blocks) fetch(blk + gridDim.x);
        // ---- 16 x 16 output tiles of this wave: out[co][t] = sum_(tap, ci) W * X[ci][t + tap * DIL - pad] ----
        const float* xl = smem + l4 * WP + off + l15;           // B operand of MFMA (tap, kb): xl[(4 kb) * WP + tap * DIL + tile offset]
#pragma unroll 1
        for (int tt = 0; tt < TPW; tt += 32) {                  // two tiles per iteration: two independent accumulator chains
            const int tl = fsl * TPW + tt;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int tap = i / (C / 4), kb = i % (C / 4);
                const float x0 = xl[(4 * kb) * WP + tap * DIL + tl];
                const float x1 = xl[(4 * kb) * WP + tap * DIL + tl + 16];
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], x0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], x1, d1, 0, 0, 0);
            }
            // ---- epilogue: D[row = 4 (l >> 4) + r][col = l & 15] ----
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 d = half ? d1 : d0;
                const int t = t0 + tl + 16 * half + l15;
                if (t < p.To) {
                    const long long o = ((long long)b * C + cot * 16 + 4 * l4) * p.To + t;
                    float rv[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) rv[r] = p.res[o + (long long)r * p.To];
                    }
                    if (p.accum) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) av[r] = p.out[o + (long long)r * p.To];
                    }
                    const bool live = !p.vlen || t < p.vlen[b];      // (ragged batch: zeros beyond the utterance's length)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float y = d[r] + bias4[r];
                        y += rv[r];
                        y += av[r];
                        if (p.out_div != 1.0f) y = y / p.out_div;
                        p.out[o + (long long)r * p.To] = live ? y : 0.f;
                    }
                }
            }
        }
    }
}

static thread_local char g_scfg[64] = "";
const char* conv_small_last_config() { return g_scfg; }

template <int C, int KT, int DIL>
static hipError_t launch_small_cfg(const ConvArgs& a, hipStream_t s) {
    using Cfg = SmallCfg<C, KT, DIL>;
    auto kern = conv_small_kernel<C, KT, DIL>;
    if (Cfg::LDS_BYTES > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    const int n_blocks = a.B * ((a.Tsrc + Cfg::TB - 1) / Cfg::TB);
    const int grid = n_blocks < 2048 ? n_blocks : 2048;      // weights are loaded once per workgroup, which strides over the blocks
    snprintf(g_scfg, sizeof(g_scfg), "C%d KT%d D%d grid %d", C, KT, DIL, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, s, a, n_blocks);
    return hipGetLastError();
}

// ---- one output channel (the generator's conv_post: 16 -> 1 channels, k 7, LeakyReLU in, tanh out; models.py:218-221) ----
// 2 FLOP per input byte: a pure stream.  A thread produces 4 consecutive samples from three aligned 16-byte loads per input
// channel (the window t-4 .. t+7; neighbours' loads overlap and hit L1), weights broadcast from LDS.
// Per-sample summation order: channel ascending, tap ascending, one fmaf chain.
template <int KT>
__global__ void __launch_bounds__(256) conv_mono_kernel(const ConvArgs p) {
    static_assert(KT == 7, "window of 12 samples = 4 outputs + 6 halo + 2 alignment");
    __shared__ float wl[64 * KT];
    const int C = p.Ci, T = p.Tsrc;
    for (int i = threadIdx.x; i < C * KT; i += 256) {
        const int ci = i / KT, tap = i - ci * KT;
        wl[i] = p.w[((((long long)tap * (C / 8) + (ci >> 3)) * 2 + (ci & 1)) * p.Mp + 0) * 4 + ((ci & 7) >> 1)];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const float* xb = p.x1 + (long long)b * C * T;
    const bool fast = (T & 3) == 0 && t >= 4 && t + 8 <= T;
    const float bias = p.bias ? p.bias[0] : 0.f;
    float o[4] = {bias, bias, bias, bias};
    for (int ci = 0; ci < C; ++ci) {
        const float* xr = xb + (long long)ci * T;
        float v[12];
        if (fast) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(xr + t - 4 + 4 * q);
                v[4 * q] = u[0]; v[4 * q + 1] = u[1]; v[4 * q + 2] = u[2]; v[4 * q + 3] = u[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 12; ++e) { const int s = t - 4 + e; v[e] = (s >= 0 && s < T) ? xr[s] : 0.f; }
        }
        if (p.act_in == ACT_LRELU) {
#pragma unroll
            for (int e = 1; e < 11; ++e) v[e] = (v[e] >= 0.f) ? v[e] : v[e] * p.slope;
        }
#pragma unroll
        for (int tap = 0; tap < KT; ++tap) {
            const float w = wl[ci * KT + tap];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(w, v[e + tap + 1], o[e]);      // x[t + e + tap - 3]
        }
    }
    if (p.epi == EPI_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tanhf(o[e]);
    }
    if (p.vlen) {      // ragged batch: zeros beyond the utterance's length
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (t + e < p.vlen[b]) ? o[e] : 0.f;
    }
    float* ob = p.out + (long long)b * T + t;
    if ((T & 3) == 0) *reinterpret_cast<f32x4*>(ob) = f32x4{o[0], o[1], o[2], o[3]};
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (t + e < T) ob[e] = o[e];
    }
}

bool conv_mono_applies(const ConvArgs& a) {
    return a.Co == 1 && a.Cout == 1 && a.KT == 7 && a.pad == 3 && a.dil == 1 && a.stride == 1 && !a.ups && a.phases == 1 && a.C2 == 0 && a.Ci % 8 == 0 &&
           a.Ci <= 64 && (a.epi == EPI_NONE || a.epi == EPI_TANH) && !a.bias_bc && !a.res && !a.accum && a.out_div == 1.0f && a.To == a.Tsrc && a.Tout == a.To &&
           a.xb1 == (long long)a.C1 * a.Tsrc;
}

hipError_t launch_conv_mono(const ConvArgs& a, hipStream_t s) {
    if (!conv_mono_applies(a)) return hipErrorInvalidValue;
    snprintf(g_scfg, sizeof(g_scfg), "C%d KT%d grid %d", a.Ci, a.KT, (a.Tsrc + 1023) / 1024);
    hipLaunchKernelGGL(conv_mono_kernel<7>, dim3((a.Tsrc + 1023) / 1024, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

bool conv_small_applies(const ConvArgs& a) {
    return a.Ci == 16 && a.Co == a.Ci && a.C2 == 0 && a.stride == 1 && !a.ups && a.phases == 1 && a.epi == EPI_NONE && !a.bias_bc &&
           (a.KT == 3 || a.KT == 7 || a.KT == 11) && (a.dil == 1 || a.dil == 3 || a.dil == 5) && a.pad == (a.KT - 1) * a.dil / 2 && a.To == a.Tsrc &&
           a.Tout == a.To && a.Cout == a.Co && a.xb1 == (long long)a.C1 * a.Tsrc;
}

hipError_t launch_conv_small(const ConvArgs& a, hipStream_t s) {
    if (!conv_small_applies(a)) return hipErrorInvalidValue;
#define SCASE(C_, KT_, D_) if (a.Ci == C_ && a.KT == KT_ && a.dil == D_) return launch_small_cfg<C_, KT_, D_>(a, s)
#define SROW(C_, KT_) SCASE(C_, KT_, 1); SCASE(C_, KT_, 3); SCASE(C_, KT_, 5)
    SROW(16, 3); SROW(16, 7); SROW(16, 11);      // 32 channels: two 16-row tiles need 88 weight registers per wave and gain nothing over conv_gemm (measured)
#undef SROW
#undef SCASE
    return hipErrorInvalidValue;
}

}  // namespace lds
