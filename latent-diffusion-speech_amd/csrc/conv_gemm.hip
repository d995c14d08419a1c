// Implicit-GEMM 1-D convolution / 1x1-conv / linear layer for gfx950 on the exact-fp32 matrix
// core (v_mfma_f32_32x32x2_f32).  One kernel family serves every dense contraction of the hot
// path: reference LoRACompatibleConv / LoRACompatibleLinear / nn.Conv1d / ConvTranspose1d calls
// (reference diffusion/unet1d/resnet.py:591-641, transformer_1d.py:256-295, attention.py:130-203,
// encoder/hifi_vaegan/modules/models.py:185-262).
//
// Data layout: activations [B][C][T] (frames contiguous), weights re-packed once to [tap][ci][co]
// (co contiguous) so that BOTH MFMA operands are "k-major, unit stride along the lane axis":
//   A (weights)     lane l reads Ws[k0 + (l>>5)][m0 + (l&31)]
//   B (activations) lane l reads Xs[k0 + (l>>5)][window(n0 + (l&31), tap)]
// i.e. conflict-free ds_read_b32 and fully coalesced 16-byte global loads along the frame axis.
// The activation window (BN*stride + halo frames of BK channels) is staged in LDS once per
// K-step and re-read at shifted offsets by every tap (the "LDS-staged 1-D convolution window").
// While staging, the producer's normalisation is applied on the fly (GroupNorm/LayerNorm
// statistics come from small side kernels), optionally followed by SiLU / LeakyReLU, so
// normalised tensors are never materialised in HBM.  A second source pointer implements the
// UNet's skip-concat on read; `ups` reads a 2x nearest-upsampled view of the source.
#include "kernels.h"

#include <math.h>
#include <stdio.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

static __device__ __forceinline__ int floor4(int s) { return (s >= 0) ? (s & ~3) : -(((-s) + 3) & ~3); }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX>
struct ConvCfg {
    static constexpr int BK = 16;
    static constexpr int WAVES_M = (BM >= 64) ? 2 : 1;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int TM = BM / (32 * WAVES_M);
    static constexpr int TN = BN / (32 * WAVES_N);
    static constexpr int XWMAX = UPS ? (BN / 2 + 2 + 3) : ((BN - 1) * STRIDE + (KT - 1) * DILMAX + 1 + 3);
    static constexpr int XW4MAX = (XWMAX + 3) / 4;
    static constexpr int XCH = (BK * XW4MAX + 255) / 256;
    static constexpr int WCHUNKS = KT * BK * BM / 4;
    static constexpr int WCH = (WCHUNKS + 255) / 256;
    static constexpr size_t lds_bytes(int xw4) { return (size_t)(KT * BK * BM + BK * xw4 * 4) * sizeof(float); }
};

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const ConvArgs p) {
    using Cfg = ConvCfg<BM, BN, KT, STRIDE, UPS, DILMAX>;
    constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN, XCH = Cfg::XCH, WCH = Cfg::WCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                    // [KT][BK][BM]
    float* Xs = smem + KT * BK * BM;     // [BK][xwp]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;

    const int nMb = p.Mp / BM;
    const int mb = blockIdx.x % nMb;          // M fastest: blocks that share an activation tile are neighbours
    const int nb = blockIdx.x / nMb;
    const int b = blockIdx.y;
    const int m0 = mb * BM, t0 = nb * BN;

    // ---- activation window geometry ----
    int s0, s_al, off, width;
    if (UPS) {
        s0 = (t0 - 1) >> 1;
        s_al = floor4(s0);
        off = 0;
        width = ((t0 + BN) >> 1) - s_al + 1;
    } else {
        s0 = t0 * STRIDE - p.pad;
        s_al = floor4(s0);
        off = s0 - s_al;
        width = (BN - 1) * STRIDE + (KT - 1) * p.dil + 1 + off;
    }
    const int xw4 = (width + 3) >> 2;
    const int xwp = xw4 * 4;
    const bool vec_ok = ((p.Tsrc & 3) == 0);

    // per-thread staging assignments (loop invariant)
    int xrow[XCH], xcol[XCH];
    bool xval[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        int q = tid + i * 256;
        xval[i] = q < BK * xw4;
        xrow[i] = q / xw4;
        xcol[i] = (q - xrow[i] * xw4) * 4;
    }
    float4 cm4[XCH], cr4[XCH];
    if (p.norm_mode == NORM_COLSTAT) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            float m_[4], r_[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int s = s_al + xcol[i] + e;
                bool ok = xval[i] && s >= 0 && s < p.Tsrc;
                m_[e] = ok ? p.cmean[(long long)b * p.Tsrc + s] : 0.f;
                r_[e] = ok ? p.crstd[(long long)b * p.Tsrc + s] : 0.f;
            }
            cm4[i] = make_float4(m_[0], m_[1], m_[2], m_[3]);
            cr4[i] = make_float4(r_[0], r_[1], r_[2], r_[3]);
        }
    }

    float4 xr[XCH], xc[XCH], wr[WCH];

    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            xc[i] = make_float4(0.f, 1.f, 0.f, 0.f);
            if (xval[i]) {
                const int ci = kc * BK + xrow[i];
                const float* src = (ci < p.C1) ? (p.x1 + (long long)b * p.xb1 + (long long)ci * p.Tsrc)
                                               : (p.x2 + (long long)b * p.xb2 + (long long)(ci - p.C1) * p.Tsrc);
                const int s = s_al + xcol[i];
                if (vec_ok && s >= 0 && s + 3 < p.Tsrc) {
                    xr[i] = *reinterpret_cast<const float4*>(src + s);
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (s + e >= 0 && s + e < p.Tsrc) ? src[s + e] : 0.f;
                    xr[i] = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (p.norm_mode == NORM_ROWCOEF) xc[i] = p.coef[(long long)b * p.Ci + ci];
                else if (p.norm_mode == NORM_COLSTAT) { xc[i].x = p.gamma[ci]; xc[i].y = p.beta[ci]; }
            }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int q = tid + j * 256;
            if (q < Cfg::WCHUNKS) {
                const int tap = q / (BK * BM / 4);
                const int rem = q - tap * (BK * BM / 4);
                const int k = rem / (BM / 4), m4 = rem - k * (BM / 4);
                wr[j] = *reinterpret_cast<const float4*>(p.w + ((long long)(tap * p.Ci + kc * BK + k)) * p.Mp + m0 + 4 * m4);
            }
        }
    };

    auto xform = [&](float v, float mu, float a, float bb, bool in) -> float {
        if (!in) return 0.f;
        if (p.norm_mode != NORM_NONE) v = (v - mu) * a + bb;
        if (p.act_in == ACT_SILU) v = silu_f(v);
        else if (p.act_in == ACT_LRELU) v = (v >= 0.f) ? v : v * p.slope;
        return v;
    };

    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            if (xval[i]) {
                const int s = s_al + xcol[i];
                float4 v = xr[i];
                if (p.norm_mode == NORM_COLSTAT) {
                    const float g = xc[i].x, be = xc[i].y;
                    v.x = xform(v.x, cm4[i].x, cr4[i].x * g, be, s + 0 >= 0 && s + 0 < p.Tsrc);
                    v.y = xform(v.y, cm4[i].y, cr4[i].y * g, be, s + 1 >= 0 && s + 1 < p.Tsrc);
                    v.z = xform(v.z, cm4[i].z, cr4[i].z * g, be, s + 2 >= 0 && s + 2 < p.Tsrc);
                    v.w = xform(v.w, cm4[i].w, cr4[i].w * g, be, s + 3 >= 0 && s + 3 < p.Tsrc);
                } else {
                    const float mu = xc[i].x, a = xc[i].y, bb = xc[i].z;
                    v.x = xform(v.x, mu, a, bb, s + 0 >= 0 && s + 0 < p.Tsrc);
                    v.y = xform(v.y, mu, a, bb, s + 1 >= 0 && s + 1 < p.Tsrc);
                    v.z = xform(v.z, mu, a, bb, s + 2 >= 0 && s + 2 < p.Tsrc);
                    v.w = xform(v.w, mu, a, bb, s + 3 >= 0 && s + 3 < p.Tsrc);
                }
                *reinterpret_cast<float4*>(Xs + xrow[i] * xwp + xcol[i]) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int q = tid + j * 256;
            if (q < Cfg::WCHUNKS) *reinterpret_cast<float4*>(Ws + 4 * q) = wr[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B-operand column of this lane for tile j at tap 0
    int bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * TN * 32 + j * 32 + c;
        bcol[j] = UPS ? nl : (off + nl * STRIDE);
    }
    const int arow = wm * TM * 32 + c;

    const int nk = p.Ci / BK;
    fetch(0);
    commit();
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) fetch(kc + 1);
#pragma unroll 1
        for (int tap = 0; tap < KT; ++tap) {
            const float* wt = Ws + tap * BK * BM + arow;
            int col[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) col[j] = UPS ? (((t0 + bcol[j] + tap - 1) >> 1) - s_al) : (bcol[j] + tap * p.dil);
#pragma unroll
            for (int k2 = 0; k2 < BK / 2; ++k2) {
                const int k = 2 * k2 + h;
                float a[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[k * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = Xs[k * xwp + col[j]];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kc + 1 < nk) {
            commit();
            __syncthreads();
        }
    }

    // ---- epilogue ----
    const bool geglu = (p.epi == EPI_GEGLU) && (TM == 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (geglu && i == 1) break;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = t0 + wn * TN * 32 + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rloc = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + wm * TM * 32 + i * 32 + rloc;      // packed row
                float v = acc[i][j][r];
                int orow;
                if (geglu) {
                    float g = acc[TM - 1][j][r];
                    if (p.bias) { v += p.bias[m]; g += p.bias[m + 32]; }
                    v = v * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
                    orow = (m0 + wm * 64) / 2 + rloc;
                } else {
                    if (p.bias) v += p.bias[m];
                    orow = m;
                }
                int co = orow, to = n;
                if (p.phases > 1) { co = orow / p.phases; to = n * p.phases + (orow - co * p.phases) - p.tpad; }
                if (co < p.Cout && n < p.To && to >= 0 && to < p.Tout) {
                    if (p.bias_bc) v += p.bias_bc[(long long)b * p.Cout + co];
                    const long long oi = ((long long)b * p.Cout + co) * p.Tout + to;
                    if (p.res) v += p.res[oi];
                    if (p.epi == EPI_TANH) v = tanhf(v);
                    if (p.accum) v += p.out[oi];
                    if (p.out_div != 1.0f) v = v / p.out_div;
                    p.out[oi] = v;
                }
            }
        }
    }
}

static thread_local char g_cfg[96] = "";
const char* conv_gemm_last_config() { return g_cfg; }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<BM, BN, KT, STRIDE, UPS, DILMAX>;
    int width;
    if (UPS) width = BN / 2 + 2 + 3;
    else width = (BN - 1) * STRIDE + (KT - 1) * a.dil + 1 + 3;
    const int xw4 = (width + 3) / 4;
    if (xw4 > Cfg::XW4MAX) return hipErrorInvalidValue;
    const size_t lds = Cfg::lds_bytes(xw4);
    const int nN = (a.To + BN - 1) / BN;
    dim3 grid((a.Mp / BM) * nN, a.B);
    auto kern = conv_gemm_kernel<BM, BN, KT, STRIDE, UPS, DILMAX>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    snprintf(g_cfg, sizeof(g_cfg), "BM%d BN%d KT%d S%d U%d grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, grid.x, grid.y, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

static int auto_tile(const ConvArgs& a) {
    // Prefer the 128x128 tile (2x2 MFMA tiles per wave, highest operand reuse) when it still yields
    // >= 2 workgroups per CU; otherwise fall back to smaller tiles to fill 256 CUs.
    auto blocks = [&](int bm, int bn) -> long long {
        if (a.Mp % bm) return -1;
        return (long long)(a.Mp / bm) * ((a.To + bn - 1) / bn) * a.B;
    };
    if (a.epi == EPI_GEGLU) return (blocks(128, 128) >= 384) ? 128128 : 128064;
    if (a.Mp % 64 != 0) return 32128;
    if (a.stride == 2 || a.ups) return 64064;
    if ((a.KT == 1 || a.KT == 3) && a.dil == 1) {
        if (blocks(128, 128) >= 512) return 128128;
        if (a.Mp % 128 == 0 && blocks(128, 64) >= 512) return 128064;
        return 64064;
    }
    return (blocks(64, 128) >= 512) ? 64128 : 64064;   // vocoder taps 2/3(dilated)/7/11
}

#define LDS_CASE(BM, BN, KT, ST, UP, DM) return launch_cfg<BM, BN, KT, ST, UP, DM>(a, s)

hipError_t launch_conv_gemm(const ConvArgs& a, int tile, hipStream_t s) {
    if (a.Ci % 16 != 0 || a.C1 % 16 != 0 || a.Mp % 32 != 0 || a.B <= 0 || a.To <= 0) return hipErrorInvalidValue;
    if (tile == 0) tile = auto_tile(a);
    const int bm = tile / 1000;
    if (a.Mp % bm != 0) return hipErrorInvalidValue;
    if (a.epi == EPI_GEGLU && bm != 128) return hipErrorInvalidValue;
    const int key = a.KT * 100 + a.stride * 10 + (a.ups ? 1 : 0);
    const bool wide = a.dil > 1;
    if (a.dil > 5) return hipErrorInvalidValue;
    switch (tile) {
        case 128128:
            if (key == 110) LDS_CASE(128, 128, 1, 1, false, 1);
            if (key == 310 && !wide) LDS_CASE(128, 128, 3, 1, false, 1);
            break;
        case 128064:
            if (key == 110) LDS_CASE(128, 64, 1, 1, false, 1);
            if (key == 310 && !wide) LDS_CASE(128, 64, 3, 1, false, 1);
            break;
        case 64064:
            if (key == 110) LDS_CASE(64, 64, 1, 1, false, 1);
            if (key == 210) LDS_CASE(64, 64, 2, 1, false, 1);
            if (key == 310 && !wide) LDS_CASE(64, 64, 3, 1, false, 1);
            if (key == 310 && wide) LDS_CASE(64, 64, 3, 1, false, 5);
            if (key == 320) LDS_CASE(64, 64, 3, 2, false, 1);
            if (key == 311) LDS_CASE(64, 64, 3, 1, true, 1);
            if (key == 710) LDS_CASE(64, 64, 7, 1, false, 5);
            if (key == 1110) LDS_CASE(64, 64, 11, 1, false, 5);
            break;
        case 64128:
            if (key == 110) LDS_CASE(64, 128, 1, 1, false, 1);
            if (key == 210) LDS_CASE(64, 128, 2, 1, false, 1);
            if (key == 310) LDS_CASE(64, 128, 3, 1, false, 5);
            if (key == 710) LDS_CASE(64, 128, 7, 1, false, 5);
            if (key == 1110) LDS_CASE(64, 128, 11, 1, false, 5);
            break;
        case 32128:
            if (key == 110) LDS_CASE(32, 128, 1, 1, false, 1);
            if (key == 210) LDS_CASE(32, 128, 2, 1, false, 1);
            if (key == 310) LDS_CASE(32, 128, 3, 1, false, 5);
            if (key == 710) LDS_CASE(32, 128, 7, 1, false, 5);
            if (key == 1110) LDS_CASE(32, 128, 11, 1, false, 5);
            break;
        default: break;
    }
    return hipErrorInvalidValue;
}

size_t packed_conv_elems(int Co, int Ci, int K, int* Mp_out) {
    int Mp = (Co + 63) / 64 * 64;
    if (Co <= 32) Mp = 32;
    if (Mp_out) *Mp_out = Mp;
    return (size_t)K * Ci * Mp;
}

}  // namespace lds
