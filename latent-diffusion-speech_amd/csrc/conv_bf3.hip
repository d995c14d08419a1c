// fp32-EQUIVALENT implicit-GEMM convolution for the UNet on the bf16 matrix pipe (gfx950, v_mfma_f32_32x32x16_bf16).
//
// Why this kernel exists.  The exact-fp32 MFMA (conv_dma.hip) runs at the vector rate, 64 FLOP/clk/SIMD; the bf16 MFMA at 1024.
// Every fp32 operand is held as three bf16 terms (k8b3.h: lossless, 6 bytes per element) and a product is the six bf16 products
// >= 2^-24 |ab| accumulated in fp32: 6/16 of the fp32-MFMA time for the same reduction, with FEWER roundings per output (one per
// 16-deep MFMA and product instead of one per k).  Built like conv_dma: no per-element VALU work in the main loop, both operand
// tiles by buffer-addressed LDS-DMA in an NST-stage ring, counted vmcnt, one s_barrier per K-step, operand registers read one
// MFMA group ahead through the K-step boundary.  Differences:
//   * a 16-byte LDS entry = 8 consecutive channels of one plane at one frame = one lane's A / B fragment of a 16-deep MFMA
//     (lanes 0-31: channels 0-7, lanes 32-63: channels 8-15 of the group), three planes per 8-channel block;
//   * an MFMA group = 16 channels of one tap: 3 + 3 fragments per 32 x 32 tile pair, 6 MFMAs per output tile;
//   * DMA chunks are dealt round-robin over the four waves and every wave issues the same number of instructions (lanes past the
//     end of a tile read outside the buffer and store zeros into the stage's padding), so the counted waits are wave-uniform;
//   * the epilogue splits the fp32 results into the three planes (K8B3 outputs) or stores fp32 (q / k for the attention kernel,
//     the value rows in its VT layout, eps in the caller's frame-major layout).
// Reference operator set: LoRACompatibleConv / Linear of the UNet (reference diffusion/unet1d/resnet.py:591-641, attention.py:130-301).
#include "k4p.h"
#include "gn_chan.h"
#include "k8b3.h"
#include "kernels.h"

#include <hip/hip_ext.h>

#include <stdio.h>
#include <stdlib.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// erf for the GEGLU epilogue: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute (same routine as conv_dma.hip)
static __device__ __forceinline__ float erf_fast_b(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return copysignf(fmaf(-pl * t, e, 1.0f), x);
}

template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ float dpp_add_b(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
static __device__ __forceinline__ float wave_sum_to_lane63_b(float v) {      // fixed order; the total is valid in lane 63
    v = dpp_add_b<0x111, 0xf, true>(v);
    v = dpp_add_b<0x112, 0xf, true>(v);
    v = dpp_add_b<0x114, 0xf, true>(v);
    v = dpp_add_b<0x118, 0xf, true>(v);
    v = dpp_add_b<0x142, 0xa, false>(v);
    v = dpp_add_b<0x143, 0xc, false>(v);
    return v;
}

// NPROD: bf16 products per fp32 product: 6 (the shipped form), 9 (all), 3 (a1b1 + a1b2 + a2b1: ~2^-16, probe only)
template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int FMT = FMT_BF16X3>
struct Bf3Cfg {
    static constexpr int NPL = fmt_planes(FMT);                        // planes per 8-channel block: 3 (bf16) or 2 (fp16)
    static constexpr bool SPLIT = BM == 32;                             // 32 x 64 tile, the K-step's k-groups split over the two wave pairs
    static constexpr int TM = SPLIT ? 1 : BM / 64, TN = BN / 64;
    static constexpr int KB = BK / 8;                                   // 8-channel blocks per K-step
    static constexpr int KG = BK / 16;                                  // 16-channel MFMA groups per K-step and tap
    static constexpr int XW = UPS ? (BN / 2 + 2) : ((BN - 1) * STRIDE + KT);      // window entries (frames) per row
    static constexpr int WCH = KT * KB * NPL * BM / 64;                   // 1 KB weight chunks per stage
    static constexpr int NWI = (WCH + 3) / 4;                           // ... per wave (round-robin; chunks past WCH store zeros)
    static constexpr int AE = KB * NPL * XW;                              // activation entries per stage
    static constexpr int ACH = (AE + 63) / 64;
    static constexpr int NXI = (ACH + 3) / 4;
    static constexpr int PER_TILE = NWI + NXI;                          // VMEM ops per wave per tile (the same for every wave)
    static constexpr int W_FLOATS = NWI * 4 * 256, X_FLOATS = NXI * 4 * 256;
    static constexpr int STAGE = W_FLOATS + X_FLOATS;                   // floats
    static constexpr size_t LDS_BYTES = (size_t)NST * STAGE * sizeof(float);
    static constexpr int KGW = SPLIT ? KG / 2 : KG;                     // k-groups per tap handled by one wave
    static constexpr int G = KT * KGW;                                  // MFMA groups per K-step and wave
    static constexpr int OCC_LDS = (int)((160 * 1024) / LDS_BYTES);
    static constexpr int OCC = (TM * TN >= 4) ? (OCC_LDS < 2 ? (OCC_LDS < 1 ? 1 : OCC_LDS) : 2) : (OCC_LDS < 4 ? (OCC_LDS < 1 ? 1 : OCC_LDS) : 4);
    static_assert(BK % 16 == 0, "a K-step is a whole number of 16-channel MFMA groups");
    static_assert(!SPLIT || (BN == 64 && BK % 32 == 0), "split-K tile is 32 x 64 with an even number of k-groups");
    static_assert(LDS_BYTES <= 160 * 1024, "stages exceed the LDS");
};

// GNF: the GroupNorm fold of DmaConvArgs::gnf_part (separate instantiations).  The split planes cannot be scaled on their way to the
// MFMA (a 16-bit term times an fp32 rstd is no 16-bit term), so here the fp32 accumulators change their unit at group boundaries:
// sum_g rstd_g S_g = ((S_0 r_0/r_1 + S_1) r_1/r_2 + ...) r_last, one multiplication of the accumulators per boundary of a wave's K range.
template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int NPROD, int FMT = FMT_BF16X3, bool GNF = false>
struct Bf3Kernel {
    using Cfg = Bf3Cfg<BM, BN, KT, STRIDE, UPS, BK, NST, FMT>;
    static constexpr int NPL = Cfg::NPL;
    static constexpr int TM = Cfg::TM, TN = Cfg::TN, KB = Cfg::KB, XW = Cfg::XW, NWI = Cfg::NWI, NXI = Cfg::NXI;
    static constexpr int G = Cfg::G, KGW = Cfg::KGW, STAGE = Cfg::STAGE, PER_TILE = Cfg::PER_TILE, W_FLOATS = Cfg::W_FLOATS;
    static constexpr bool SPLIT = Cfg::SPLIT;
    static constexpr int NB = 2;                                        // operand ring depth; odd G alternates the slot phase per K-step

    const DmaConvArgs& p;
    float* smem;
    int lane, wave, c, h, wm, wn, ks, b, m0, t0;
    int ksp, kc0, ctile;      // cluster split-K (latency mode; conv_dma.hip cluster_join): share index, first K-step, tile index
    int Lout;                 // valid output frames of this batch element (ragged batches, k4p.h ragged_len; = To otherwise)
    // GroupNorm fold: (threads < BM) the terms of their row's constant, this wave's two groups' partial statistics (both in flight across
    // the first DMAs), the group of this wave's share of the current K-step and the share's channel offset inside it
    float gcg[GNF ? 8 : 1], gkc;
    GnPart gp0, gp1;
    int ggi, gpos;
    int woff[NWI];            // per-lane byte offsets of this wave's weight chunks (loop invariant)
    int xoff[NXI];            // per-lane byte offsets of this wave's activation chunks inside a source slab
    __amdgpu_buffer_rsrc_t rw, rx1, rx2;
    int arow;
    int bcol[TN];
    f32x16 acc[TM][TN];
    u32x4 aop[NB][TM][NPL], bop[NB][TN][NPL];
    float lmu[TN], lrs[TN];   // folded input-LayerNorm statistics of this lane's output columns
    static constexpr bool EARLY = TM * TN == 1;      // single-tile waves fetch the epilogue's residual at kernel start
    u32x2 rsv[EARLY ? 4 * NPL : 1];

    __device__ __forceinline__ Bf3Kernel(const DmaConvArgs& p_, float* s_) : p(p_), smem(s_) {}

    __device__ __forceinline__ void setup_keep_acc() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        c = lane & 31; h = lane >> 5;
        wm = SPLIT ? 0 : wave >> 1; ks = SPLIT ? wave >> 1 : 0; wn = wave & 1;
        // XCD-aware tile order (conv_dma.hip): the M-blocks of one (batch, frame-block) take consecutive slots of ONE XCD
        const int nMb = p.Mp / BM;
        const int gx = gridDim.x, total = gx * gridDim.y;
        const int id = blockIdx.y * gx + blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, per = total >> 3, rem = total & 7;
        const int L = xcd * per + (xcd < rem ? xcd : rem) + slot;
        const int mb = L % nMb;
        const int tb = L / nMb;
        const int nN = gx / nMb;
        const int nb = tb % nN;
        const int by = tb / nN, S = p.ksplit > 1 ? p.ksplit : 1;      // grid.y = B * S
        b = by / S; ksp = by - b * S;
        kc0 = ksp * (p.Ci / BK / S);
        ctile = (b * nN + nb) * nMb + mb;
        m0 = mb * BM; t0 = nb * BN;
        Lout = ragged_len(p.lens, b, p.lvl_out, p.To);
        constexpr int kOob = 0x40000000;      // beyond every buffer: the range check returns zeros
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int e = (wave + 4 * i) * 64 + lane;           // stage entry = (((tap * KB + kb) * 3 + plane) * BM + m)
            const int m = e % BM, r = e / BM;
            const int pl = r % NPL, kbt = r / NPL;
            const int kb = kbt % KB, tap = kbt / KB;
            woff[i] = (tap < KT) ? ((((tap * (p.Ci >> 3) + kb) * NPL + pl) * p.Mp + m0 + m) * 16) : kOob;
        }
        const int Tp = p.Tsrc + 2;
        const int e0 = UPS ? (((t0 - 1) >> 1) + 1) : (t0 * STRIDE - p.pad + 1);   // first window entry (entry 1 = frame 0)
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = (wave + 4 * i) * 64 + lane;           // stage entry = (kb * 3 + plane) * XW + col
            const int r = e / XW, col = e - r * XW;
            xoff[i] = (e < Cfg::AE) ? ((r * Tp + e0 + col) * 16) : kOob;
        }
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, KT * p.Ci * p.Mp * 2 * NPL, 0x00020000);
        const char* x1b = reinterpret_cast<const char*>(p.x1) + (long long)b * p.C1 * Tp * 2 * NPL;
        const char* x2b = p.x2 ? reinterpret_cast<const char*>(p.x2) + (long long)b * p.C2 * Tp * 2 * NPL : x1b;
        rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x1b), 0, p.C1 * Tp * 2 * NPL, 0x00020000);
        rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x2b), 0, (p.x2 ? p.C2 : p.C1) * Tp * 2 * NPL, 0x00020000);
        arow = wm * TM * 32 + c;
#pragma unroll
        for (int j = 0; j < TN; ++j) bcol[j] = wn * TN * 32 + j * 32 + c;
    }
    __device__ __forceinline__ bool dead_tile() const { return p.lens && t0 >= Lout; }
    __device__ __forceinline__ void setup() {
        setup_keep_acc();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // tile kc -> LDS stage `st`: this wave's share of the weight chunks and of the activation window
    __device__ __forceinline__ void issue_tile(int kc, float* st) {
        const int ws = kc * (KB * NPL * 16) * p.Mp;              // bytes
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + (wave + 4 * i) * 256), 16, woff[i], ws, 0, 0);
        const int k0 = kc * BK;
        const bool s2 = k0 >= p.C1;                              // C1 % BK == 0: a K-step reads one source only
        const int xsoff = ((s2 ? k0 - p.C1 : k0) >> 3) * NPL * (p.Tsrc + 2) * 16;
        float* xs = st + W_FLOATS;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 256), 16, xoff[i], xsoff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 256), 16, xoff[i], xsoff, 0, 0);
        }
    }

    // operands of MFMA group (tap, kg): lane half h takes 8-channel block 2 * kg + h of the K-step
    template <int SLOT>
    __device__ __forceinline__ void load_ops(const float* st, int tap, int kgl) {
        const int kb = 2 * (ks * KGW + kgl) + h;
        const float* wt = st + ((tap * KB + kb) * NPL * BM + arow) * 4;
        const float* xs = st + W_FLOATS + (kb * NPL * XW) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) aop[SLOT][i][pl] = *reinterpret_cast<const u32x4*>(wt + (pl * BM + i * 32) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = UPS ? (((t0 + bcol[j] + tap - 1) >> 1) - ((t0 - 1) >> 1)) : (bcol[j] * STRIDE + tap);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bop[SLOT][j][pl] = *reinterpret_cast<const u32x4*>(xs + (pl * XW + col) * 4);
        }
    }
    template <int SLOT, int PA, int PB>
    __device__ __forceinline__ void mfma_pair() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if constexpr (FMT == FMT_F16X2)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aop[SLOT][i][PA]), __builtin_bit_cast(f16x8, bop[SLOT][j][PB]),
                                                                       acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aop[SLOT][i][PA]), __builtin_bit_cast(bf16x8, bop[SLOT][j][PB]),
                                                                        acc[i][j], 0, 0, 0);
    }
    template <int SLOT>
    __device__ __forceinline__ void mfma_ops() {
        // smallest terms first: what they add is below the running sum's rounding either way, the order only fixes the result
        if constexpr (FMT == FMT_F16X2) {      // a1b1 + a1b2 + a2b1 (+ a2b2 with NPROD 4): the dropped term is below 2^-22 |ab|
            if constexpr (NPROD >= 4) mfma_pair<SLOT, 1, 1>();
        } else {
            if constexpr (NPROD >= 9) { mfma_pair<SLOT, 2, 2>(); mfma_pair<SLOT, 1, 2>(); mfma_pair<SLOT, 2, 1>(); }
            if constexpr (NPROD >= 6) { mfma_pair<SLOT, 0, 2>(); mfma_pair<SLOT, 2, 0>(); mfma_pair<SLOT, 1, 1>(); }
        }
        mfma_pair<SLOT, 0, 1>();
        mfma_pair<SLOT, 1, 0>();
        mfma_pair<SLOT, 0, 0>();
    }

    // ---- GroupNorm fold (DmaConvArgs::gnf_part; conv_dma.hip has the fp32 twin) ----
    // LDS behind the ring: [0, BM) the per-row constants, then 8 x rstd_g, 8 x 1 / rstd_g, 8 x rstd_g * mean_g of this batch element
    __device__ __forceinline__ float* gnf_tail() const { return smem + NST * STAGE + BM; }
    __device__ __forceinline__ void gnf_scale(float f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
    }
    __device__ __forceinline__ void gnf_request() {      // kernel start, before the first DMA
        const int G_ = p.gnf_groups, gsz = p.Ci / G_;
        gp0 = gn_part_load(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, wave, lane, 0, ragged_len(p.lens, b, p.lvl_in, p.Tsrc));
        gp1 = gn_part_load(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, (wave + 4 < G_) ? wave + 4 : wave, lane, 0, ragged_len(p.lens, b, p.lvl_in, p.Tsrc));
        if ((int)threadIdx.x < BM) {
            const int m = m0 + (int)threadIdx.x;
            gkc = p.gnf_c2[m];
#pragma unroll
            for (int g = 0; g < 8; ++g) gcg[g] = (g < G_) ? p.gnf_cg[g * p.Mp + m] : 0.f;
        }
    }
    __device__ __forceinline__ void gnf_prepare() {      // before the first barrier: wave w publishes groups w and w + 4
        const int G_ = p.gnf_groups, gsz = p.Ci / G_;
        float* tail = gnf_tail();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int g = wave + 4 * e;
            float rs = 1.f, sd = 1.f, rm = 0.f;
            if (g < G_) {
                float mu, var;
                gnf_group_stats(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, g, lane, e ? gp1 : gp0, mu, var, ragged_len(p.lens, b, p.lvl_in, p.Tsrc));
                sd = sqrtf(var + p.gnf_eps); rs = 1.0f / sd; rm = rs * mu;
            }
            if (lane == 0) { tail[g] = rs; tail[8 + g] = sd; tail[16 + g] = rm; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int span = SPLIT ? BK / 2 : BK;
        const int start = kc0 * BK + ks * span;      // first channel of this wave's share of its first K-step
        ggi = start / gsz; gpos = start - ggi * gsz;
    }
    __device__ __forceinline__ void gnf_rows() {         // after the first barrier: kc[m] = gnf_c2[m] - sum_g rstd_g mean_g gnf_cg[g][m]
        if ((int)threadIdx.x < BM) {
            const float* tail = gnf_tail();
            float kc = gkc;
#pragma unroll
            for (int g = 0; g < 8; ++g) kc = fmaf(-tail[16 + g], gcg[g], kc);
            smem[NST * STAGE + threadIdx.x] = kc;
        }
    }
    __device__ __forceinline__ void gnf_advance() {      // after a K-step that is not the last
        const int gsz = p.Ci / p.gnf_groups;
        gpos += BK;
        int g1 = ggi;
        while (gpos >= gsz) { gpos -= gsz; ++g1; }
        if (g1 != ggi) {
            const float* tail = gnf_tail();
            gnf_scale(tail[ggi] * tail[8 + g1]);
            ggi = g1;
        }
    }

    template <int Y>
    __device__ __forceinline__ void wait_younger(int y) {      // s_waitcnt vmcnt(y * PER_TILE) for a wave-uniform y in [0, Y]
        if constexpr (Y == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            constexpr int N = (Y * PER_TILE > 63) ? 63 : Y * PER_TILE;
            if (y >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
            else wait_younger<Y - 1>(y);
        }
    }
    __device__ __forceinline__ void tile_sync(int kc, int nk, float* cur) {
        const int younger = (nk - 2 - kc < NST - 2) ? (nk - 2 - kc) : (NST - 2);     // tiles after kc+1 still in flight
        wait_younger<NST - 2>(younger);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kc + NST < nk) issue_tile(kc0 + kc + NST, cur);
    }
    // PH: slot phase of this K-step (odd G: the ring's slots swap roles every K-step)
    template <int g, int PH>
    __device__ __forceinline__ void kstep(float* cur, const float* nxt, int kc, int nk) {
        if constexpr (g < G) {
            constexpr int slot = (g + PH * G) % NB, gp = g + 1;
            if constexpr (gp < G) {
                load_ops<(slot + 1) % NB>(cur, gp / KGW, gp % KGW);
            } else {
                if (kc + 1 < nk) {
                    tile_sync(kc, nk, cur);
                    load_ops<(slot + 1) % NB>(nxt, 0, 0);
                }
            }
            mfma_ops<slot>();
            __builtin_amdgcn_sched_barrier(0);
            kstep<g + 1, PH>(cur, nxt, kc, nk);
        }
    }

    __device__ __forceinline__ int k8_ck() const { return (p.plain_from < p.Cout) ? p.plain_from : p.Cout; }
    __device__ __forceinline__ bool res_tile(int tile0, int n) const { return p.res && tile0 < p.plain_from && tile0 < p.Cout && n < p.To; }
    // byte offset of (8-channel block of tile0 + g, plane, frame n), this lane half's 8 bytes
    __device__ __forceinline__ long long k8_off(int Ck, int tile0, int g, int pl, int n) const {
        return ((((long long)b * (Ck >> 3) + (tile0 >> 3) + g) * NPL + pl) * (p.To + 2) + n + 1) * 16 + h * 8;
    }

    __device__ __forceinline__ void early_loads() {
        const int tile0 = m0 + wm * 32, n = t0 + wn * 32 + c;
#pragma unroll
        for (int e = 0; e < 4 * NPL; ++e) rsv[e] = u32x2{0u, 0u};
        if (res_tile(tile0, n)) {
            const char* rb = reinterpret_cast<const char*>(p.res);
            const int Ck = k8_ck();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) rsv[g * NPL + pl] = *reinterpret_cast<const u32x2*>(rb + k8_off(Ck, tile0, g, pl, n));
        }
    }

    // LayerNorm over the input channels folded into the epilogue (conv_dma.hip ln_columns): per output column combine the
    // producer's per-32-channel (mean, M2) partials in a fixed order (gn_chan.h ln_column_stats)
    __device__ __forceinline__ void ln_columns() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = t0 + wn * TN * 32 + j * 32 + c;
            const bool ok = n < p.To;
            const float2* src = p.ln_part + (long long)b * p.ln_np * p.Tsrc + (ok ? n : 0);
            ln_column_stats(src, p.Tsrc, p.ln_np, p.ln_eps, ok, lmu[j], lrs[j]);      // (gn_chan.h: two plain sums in a fixed order)
        }
    }

    __device__ __forceinline__ void mainloop() {
        const int nk = p.Ci / BK / (p.ksplit > 1 ? p.ksplit : 1);      // this workgroup's K-steps: kc0 .. kc0 + nk - 1
        if constexpr (GNF) gnf_request();
        if constexpr (EARLY) early_loads();
        for (int t = 0; t < NST && t < nk; ++t) issue_tile(kc0 + t, smem + t * STAGE);
        if (p.ln_part) ln_columns();
        if constexpr (GNF) gnf_prepare();
        wait_younger<NST - 1>((nk - 1 < NST - 1) ? nk - 1 : NST - 1);      // tile 0 landed (this wave's share)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (GNF) gnf_rows();
        load_ops<0>(smem, 0, 0);
        int sc = 0;
        if constexpr (G % 2 == 0) {
            for (int kc = 0; kc < nk; ++kc) {
                const int sn = (sc + 1 == NST) ? 0 : sc + 1;
                kstep<0, 0>(smem + sc * STAGE, smem + sn * STAGE, kc, nk);
                sc = sn;
                if constexpr (GNF) { if (kc + 1 < nk) gnf_advance(); }
            }
        } else {
            for (int kc = 0; kc < nk; kc += 2) {
                int sn = (sc + 1 == NST) ? 0 : sc + 1;
                kstep<0, 0>(smem + sc * STAGE, smem + sn * STAGE, kc, nk);
                sc = sn;
                if constexpr (GNF) { if (kc + 1 < nk) gnf_advance(); }
                if (kc + 1 < nk) {
                    sn = (sc + 1 == NST) ? 0 : sc + 1;
                    kstep<0, 1>(smem + sc * STAGE, smem + sn * STAGE, kc + 1, nk);
                    sc = sn;
                    if constexpr (GNF) { if (kc + 2 < nk) gnf_advance(); }
                }
            }
        }
        if constexpr (GNF) {
            gnf_scale(gnf_tail()[ggi]);      // the last group's rstd, before partial sums of different waves / workgroups meet
            __syncthreads();                 // (the row constants are read in finalize: a one-K-step launch has met no barrier since)
        }
    }

    // ---- epilogue (phases as in conv_dma.hip: row constants -> residuals of all tiles -> stores) ----
    __device__ __forceinline__ void finalize(bool geglu) {
        const bool ln = p.ln_part != nullptr;
        if (p.acc_scale != 0.f && p.acc_scale != 1.0f) {      // weights stored times a power of two (fp16 planes): undone exactly
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.acc_scale;
        }
        if (ln || p.bias) {
            float k1[TM][16], k2[TM][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // packed weight row
                    k1[i][r] = ln ? p.ln_c1[m] : 0.f;
                    k2[i][r] = ln ? p.ln_c2[m] : p.bias[m];
                }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        acc[i][j][r] = ln ? lrs[j] * (v - lmu[j] * k1[i][r]) + k2[i][r] : v + k2[i][r];
                    }
        }
        if constexpr (GNF) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float kc = smem[NST * STAGE + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] += kc;
                }
        }
        if (geglu) {      // rows of tile 0 are the values, rows of tile 1 the gates (pack_geglu)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = acc[TM - 1][j][r];
                    acc[0][j][r] *= 0.5f * g * (1.0f + erf_fast_b(g * 0.70710678118654752440f));
                }
        }
    }

    __device__ __forceinline__ void add_residual(int tile0, int i, int j, int n) {
        if (n >= p.To) return;
        u32x2 rv[4 * NPL];
        if constexpr (EARLY) {
#pragma unroll
            for (int e = 0; e < 4 * NPL; ++e) rv[e] = rsv[e];
        } else {
            const char* rb = reinterpret_cast<const char*>(p.res);
            const int Ck = k8_ck();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) rv[g * NPL + pl] = *reinterpret_cast<const u32x2*>(rb + k8_off(Ck, tile0, g, pl, n));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned lo[NPL], hi[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { lo[pl] = rv[g * NPL + pl][0]; hi[pl] = rv[g * NPL + pl][1]; }
            float a0, a1, a2, a3;
            sp_join_pair<FMT>(lo, a0, a1);
            sp_join_pair<FMT>(hi, a2, a3);
            acc[i][j][4 * g] += a0; acc[i][j][4 * g + 1] += a1; acc[i][j][4 * g + 2] += a2; acc[i][j][4 * g + 3] += a3;
        }
    }

    // frame-major store of one 32x32 tile: out[b][co][n], co = c0 + local row
    __device__ __forceinline__ void store_plain(float* base, int Cn, int c0, int i, int j, int n) {
        if (n >= p.To) return;
        float* ob = base + ((long long)b * Cn + c0) * p.To + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (c0 + rl < Cn) ob[rl * p.To] = acc[i][j][r];
        }
    }
    // attention's VT layout for the value channels (attention_k4p.hip): [B][head][ceil(To/4)][D][4]
    __device__ __forceinline__ void store_vt(int c0, int i, int j, int n) {
        const int T4 = (p.To + 3) & ~3, D = p.vt_D, Cv = p.Cout - p.plain_from;
        if (n >= T4) return;
        const bool real = n < p.To;
        const int head0 = c0 / D, rem0 = c0 - head0 * D;
        float* ob = p.out2 + (long long)b * Cv * T4 + (long long)(n >> 2) * D * 4 + (n & 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
            int d = rem0 + rl, head = head0;
            if (d >= D) { d -= D; ++head; }
            if (c0 + rl < Cv) ob[(long long)head * D * T4 + d * 4] = real ? acc[i][j][r] : 0.f;
        }
    }
    // fp32 K4P store (q / k for the attention kernel): channel 8q + 2jj + hh at element jj of row (q, hh).  Row 8g + 4h + e of the
    // tile is channel position 4h + e of block g: positions (4h, 4h+2) are elements (2h, 2h+1) of row hh = 0, (4h+1, 4h+3) of hh = 1.
    __device__ __forceinline__ void store_k4p_f32(int tile0, int i, int j, int n) {
        if (n >= p.To) return;
        const int Tpo = p.To + 2, Ck = k8_ck();
        float* ob = p.out + (long long)b * Ck * Tpo + (((tile0 >> 3) * 2) * Tpo + n + 1) * 4 + 2 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float* o = ob + (2 * g + hh) * Tpo * 4;
                k4p_store_wt(o, k4p_f32x2{acc[i][j][4 * g + hh], acc[i][j][4 * g + 2 + hh]});
                if (n == 0) *reinterpret_cast<f32x2*>(o - 4) = f32x2{0.f, 0.f};
                if (n == p.To - 1) *reinterpret_cast<f32x2*>(o + 4) = f32x2{0.f, 0.f};
            }
    }
    // K8B3 store of one 32x32 tile (+ pad frames, + GroupNorm / LayerNorm partials of the fp32 values)
    __device__ __forceinline__ void store_k8(int tile0, int i, int j, int n) {
        const int Ck = k8_ck();
        const bool ok = n < p.To;
        if (ok) {
            char* ob = reinterpret_cast<char*>(p.out);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned pa[NPL], pb[NPL];
                sp_split_pair<FMT>(acc[i][j][4 * g], acc[i][j][4 * g + 1], pa);
                sp_split_pair<FMT>(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], pb);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) k8_store_wt(ob + k8_off(Ck, tile0, g, pl, n), k8_u32x2{pa[pl], pb[pl]});
            }
            if (n == 0 || n == p.To - 1) {            // pad frames (entry 0 / entry To + 1 of every row)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        if (n == 0) *reinterpret_cast<u32x2*>(ob + k8_off(Ck, tile0, g, pl, n) - 16) = u32x2{0u, 0u};
                        if (n == p.To - 1) *reinterpret_cast<u32x2*>(ob + k8_off(Ck, tile0, g, pl, n) + 16) = u32x2{0u, 0u};
                    }
            }
        }
        if (p.gnpart_out) {      // (16 channels x 32 frames) (mean, M2) blocks, as conv_dma.hip store_k4p
            const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[i][j][0])));
            const float k1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[i][j][8])));
            float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float d0 = acc[i][j][r] - k0, d1 = acc[i][j][r + 8] - k1;
                a1 += d0; a2 = fmaf(d0, d0, a2);
                b1 += d1; b2 = fmaf(d1, d1, b2);
            }
            if (n >= Lout) { a1 = 0.f; a2 = 0.f; b1 = 0.f; b2 = 0.f; }      // (Lout <= To: no statistics beyond the utterance's length)
            a1 = wave_sum_to_lane63_b(a1); a2 = wave_sum_to_lane63_b(a2);
            b1 = wave_sum_to_lane63_b(b1); b2 = wave_sum_to_lane63_b(b2);
            const int n0 = n - c;
            const int nv = (Lout - n0 < 32) ? Lout - n0 : 32;
            if (lane == 63 && nv > 0) {
                const float cnt = 16.0f * (float)nv, rc = 1.0f / cnt;
                float2* gp = p.gnpart_out + ((long long)b * (Ck >> 4) + (tile0 >> 4)) * ((p.To + 31) >> 5) + (n0 >> 5);
                gp[0] = make_float2(k0 + a1 * rc, fmaxf(a2 - a1 * a1 * rc, 0.f));
                if (tile0 + 16 < Ck) gp[(p.To + 31) >> 5] = make_float2(k1 + b1 * rc, fmaxf(b2 - b1 * b1 * rc, 0.f));
            }
        }
        if (p.lnpart_out) {      // per-frame (mean, M2) over this tile's 32 channels -> LayerNorm partials
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[i][j][r];
            const float m16 = s1 * (1.0f / 16.0f);
            float qv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - m16; qv += d * d; }
            const float mo = __shfl_xor(m16, 32, 64), qo = __shfl_xor(qv, 32, 64);
            const float d = mo - m16;
            if (h == 0 && ok)
                p.lnpart_out[((long long)b * (Ck >> 5) + (tile0 >> 5)) * p.To + n] = make_float2(0.5f * (m16 + mo), (qv + qo) + d * d * 8.0f);
        }
    }

    __device__ __forceinline__ int tile_ch(int i, bool geglu) const { return geglu ? (m0 + wm * 64) / 2 : (m0 + wm * TM * 32 + i * 32); }

    // split-K: the ks = 1 waves hand their partial tile to the ks = 0 wave of the same column block through LDS
    __device__ __forceinline__ bool join_halves() {
        __syncthreads();
        float* red = smem + wn * 16 * 64 + lane;
        if (ks == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64] = acc[0][0][r];
        }
        __syncthreads();
        if (ks == 1) return false;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += red[r * 64];
        return true;
    }

    // cluster split-K of the latency mode: the hand-off of conv_dma.hip cluster_join, cell for cell (sc1 stores of every joining wave, their
    // acknowledgement, a workgroup barrier, ONE lane's add on ONE counter per tile, the add that returns S - 1 came last, an LDS word behind a
    // second barrier for the other waves, sc1 loads of the S partials in the fixed order s = 0 .. S-1)
    __device__ __forceinline__ bool cluster_join() {
        const int S = p.ksplit;
        const int slot = ctile * 4 + wave;
        float* mine = p.kpart + ((long long)slot * S + ksp) * (TM * TN * 1024) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are acknowledged ...
        __syncthreads();                                      // ... and so are those of every other (live) wave of the workgroup; the stages are free
        int* last = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {                               // (wave 0 is a joining wave of every tile shape: split tiles keep waves 0 and 1)
            const unsigned old = __hip_atomic_fetch_add(p.kcount + ctile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last = (old == (unsigned)(S - 1)) ? 1 : 0;
        }
        __syncthreads();
        if (*last == 0) return false;
        const float* all = p.kpart + (long long)slot * S * (TM * TN * 1024) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = __hip_atomic_load(all + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 1; q < S; ++q) {
            const float* pq = all + (long long)q * (TM * TN * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += __hip_atomic_load(pq + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) __hip_atomic_store(p.kcount + ctile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }

    __device__ __forceinline__ void epilogue() {
        const bool geglu = (p.epi == EPI_GEGLU) && (TM == 2);
        if constexpr (SPLIT) {
            if (!join_halves()) return;
        }
        if constexpr (TM * TN <= 2) {
            if (p.ksplit > 1) {
                if (!cluster_join()) return;
            }
        }
        finalize(geglu);
        const int ni = geglu ? 1 : TM;
        if (p.res) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i >= ni) break;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int tile0 = tile_ch(i, geglu);
                    if (tile0 < p.plain_from && tile0 < p.Cout) add_residual(tile0, i, j, t0 + wn * TN * 32 + j * 32 + c);
                }
            }
        }
        if (p.lens) {      // ragged batch: frames at and beyond this utterance's length are written as zeros (conv_dma.hip epilogue)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (t0 + wn * TN * 32 + j * 32 + c >= Lout) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i >= ni) break;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = t0 + wn * TN * 32 + j * 32 + c;
                const int tile0 = tile_ch(i, geglu);
                if (p.out_plain) store_plain(p.out, p.Cout, tile0, i, j, n);
                else if (tile0 >= p.plain_from) {
                    if (p.vt_D) store_vt(tile0 - p.plain_from, i, j, n);
                    else store_plain(p.out2, p.Cout - p.plain_from, tile0 - p.plain_from, i, j, n);
                }
                else if (tile0 < p.Cout) {
                    if (p.out_f32) store_k4p_f32(tile0, i, j, n);
                    else store_k8(tile0, i, j, n);
                }
            }
        }
    }
};

template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int NPROD, int FMT, bool GNF = false>
__global__ void __launch_bounds__(256, (Bf3Cfg<BM, BN, KT, STRIDE, UPS, BK, NST, FMT>::OCC)) conv_bf3_kernel(const DmaConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Bf3Kernel<BM, BN, KT, STRIDE, UPS, BK, NST, NPROD, FMT, GNF> k(p, smem);
    k.setup();
    if (!k.dead_tile()) k.mainloop();      // (ragged batch: a tile wholly beyond its utterance's length only writes its zeros)
    k.epilogue();
}

// resnet tail: conv2 (k 3 over h) and the 1x1 shortcut (over the block input) into one set of accumulators, one epilogue
struct Bf3PairArgs { DmaConvArgs a3, a1; };
template <int BM, int BN, int BK3, int BK1, int NST, int FMT>
struct Bf3PairCfg {
    using C3 = Bf3Cfg<BM, BN, 3, 1, false, BK3, NST, FMT>;
    using C1 = Bf3Cfg<BM, BN, 1, 1, false, BK1, NST, FMT>;
    static constexpr size_t LDS_BYTES = C3::LDS_BYTES > C1::LDS_BYTES ? C3::LDS_BYTES : C1::LDS_BYTES;
    static constexpr int OCC = C3::OCC < C1::OCC ? C3::OCC : C1::OCC;
};
template <int BM, int BN, int BK3, int BK1, int NST, int FMT>
__global__ void __launch_bounds__(256, (Bf3PairCfg<BM, BN, BK3, BK1, NST, FMT>::OCC)) conv_bf3_pair_kernel(const Bf3PairArgs pp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = FMT == FMT_F16X2 ? 3 : 6;
    using K3 = Bf3Kernel<BM, BN, 3, 1, false, BK3, NST, NP, FMT>;
    using K1 = Bf3Kernel<BM, BN, 1, 1, false, BK1, NST, NP, FMT>;
    K1 k1(pp.a1, smem);
    {
        K3 k3(pp.a3, smem);
        k3.setup();
        if (!k3.dead_tile()) k3.mainloop();
#pragma unroll
        for (int i = 0; i < K3::TM; ++i)
#pragma unroll
            for (int j = 0; j < K3::TN; ++j) k1.acc[i][j] = k3.acc[i][j];
    }
    __syncthreads();                // every wave is done reading the first phase's stages
    k1.setup_keep_acc();
    if (!k1.dead_tile()) k1.mainloop();
    k1.epilogue();
}

static thread_local char g_bcfg[112] = "";
const char* conv_bf3_last_config() { return g_bcfg; }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int NPROD, int FMT, bool GNF = false>
static hipError_t launch_bf3_cfg(const DmaConvArgs& a, hipStream_t s) {
    using Cfg = Bf3Cfg<BM, BN, KT, STRIDE, UPS, BK, NST, FMT>;
    if (GNF != (a.gnf_part != nullptr)) return hipErrorInvalidValue;
    const size_t lds_bytes = Cfg::LDS_BYTES + (GNF ? (BM + 32) * sizeof(float) : 0);      // the fold's row constants and group statistics sit behind the ring
    const int nN = (a.To + BN - 1) / BN;
    const int S = a.ksplit > 1 ? a.ksplit : 1;
    if (S > 1 && (Cfg::TM * Cfg::TN > 2 || (a.Ci / BK) % S)) return hipErrorInvalidValue;
    dim3 grid((a.Mp / BM) * nN, a.B * S);
    auto kern = conv_bf3_kernel<BM, BN, KT, STRIDE, UPS, BK, NST, NPROD, FMT, GNF>;
    if (lds_bytes > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    const char* gtag = GNF ? " GNF" : "";
    if (S > 1) snprintf(g_bcfg, sizeof(g_bcfg), "BM%d BN%d KT%d S%d U%d BK%d NST%d %s%d KS%d%s grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, NST,
                        FMT == FMT_F16X2 ? "H" : "P", NPROD, S, gtag, grid.x, grid.y, lds_bytes);
    else snprintf(g_bcfg, sizeof(g_bcfg), "BM%d BN%d KT%d S%d U%d BK%d NST%d %s%d%s grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, NST, FMT == FMT_F16X2 ? "H" : "P",
             NPROD, gtag, grid.x, grid.y, lds_bytes);
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

// tile shape / K-step / ring depth of one launch.  As in conv_dma the choice is made from per-utterance sizes at the nominal
// per-GPU batch (16), never from the actual batch: an utterance's result is bit-identical for any batch split.
// tuning knob (tools/tune_split_rules.py through lds_debug_set_split_rule): bit mask of alternative tile rules, 0 = the shipped rules
static std::atomic<int> g_split_rule{0};
void conv_bf3_set_debug_rule(int r) { g_split_rule.store(r); }

static void bf3_pick(const DmaConvArgs& a, int cfg, int& bm, int& bn, int& bk, int& nst, int fmt = FMT_BF16X3) {
    const int rule = g_split_rule.load(std::memory_order_relaxed);
    const long long kNominalBatch = a.tile_batch > 0 ? a.tile_batch : 16;      // (latency mode: the actual batch, kernels.h)
    auto blocks = [&](int bm_, int bn_) -> long long { return (a.Mp % bm_) ? -1 : (long long)(a.Mp / bm_) * ((a.To + bn_ - 1) / bn_) * kNominalBatch; };
    const bool k32 = (a.Ci % 32 == 0) && (a.C1 % 32 == 0), k64 = (a.Ci % 64 == 0) && (a.C1 % 64 == 0);
    if (cfg) {
        bm = cfg / 1000000; bn = (cfg / 1000) % 1000; bk = (cfg / 10) % 100; nst = cfg % 10;
        return;
    }
    // Rules from tools/split_bf16_probe.py (profiles/r03_split_bf16_probe.json).  The bf16 pipe eats operands 2.7x faster than the fp32
    // one while a staged element is 6 bytes instead of 4, so what decides is bytes in flight per CU: stages of <= 80 KB (two
    // co-resident workgroups) beat deeper K-steps everywhere, wide tiles (fewer bytes per MFMA) win as soon as they still fill the chip.
    if (a.stride == 2 || a.ups) { bm = 64; bn = 64; bk = 16; nst = 3; return; }
    if (a.epi == EPI_GEGLU) {
        bm = 128;
        // (fp16 planes: BK 32 x 2 stages wins back to back on hot operands, 33.9 vs 36.4 us at 256 -> 2048 @ 512, and loses inside the model,
        //  53 vs 40 us at 384 -> 3072 @ 256: three co-resident workgroups of 16-deep stages hide the cold first tiles better)
        // (latency mode, a grid that leaves most of the chip idle: the half-width tile, whose two-block waves may share K in a cluster)
        if (a.tile_batch > 0 && (long long)(a.Mp / 128) * ((a.To + 127) / 128) * a.tile_batch < 192) { bn = 64; bk = k32 ? 32 : 16; nst = k32 ? 2 : 3; }
        else if (a.To > 64 && fmt == FMT_F16X2 && (rule & 16)) { bn = 128; bk = 16; nst = 4; }
        else if (a.To > 64) { bn = 128; bk = 16; nst = 3; }
        else { bn = 64; bk = k32 ? 32 : 16; nst = k32 ? 2 : 3; }
        return;
    }
    const long long b64 = blocks(64, 64);
    // fewer 64 x 64 tiles than CUs, or 1.5 per CU (conv_dma.hip dma_pick): 32 x 64 tiles with the K-step's groups split over the wave pairs
    const bool small = k32 && (b64 <= 256 || (b64 < 512 && b64 % 256));
    if (a.KT == 3) {
        if (small) { bm = 32; bn = 64; bk = 32; nst = 2; }
        else if ((rule & 4) && k32) { bm = 64; bn = 64; bk = 32; nst = 2; }
        else if (blocks(64, 128) >= 256 && a.To >= 128) { bm = 64; bn = 128; bk = 16; nst = 2; }
        else { bm = 64; bn = 64; bk = 16; nst = 3; }
        return;
    }
    // (in the model, where operands arrive cold, the split tiles are bound by the number of sequential DMA round trips: the deepest K-step)
    if (small) { bm = 32; bn = 64; bk = (k64 && !(rule & 8)) ? 64 : 32; nst = 2; }
    else if (k32 && blocks(128, 64) >= 512 && !(rule & 1)) { bm = 128; bn = 64; bk = 32; nst = 2; }
    else if (k64 && (rule & 2)) { bm = 64; bn = 64; bk = 64; nst = 2; }
    else if (k32) { bm = 64; bn = 64; bk = 32; nst = 3; }
    else { bm = 64; bn = 64; bk = 16; nst = 3; }
}

template <int BM, int BN, int BK3, int BK1, int NST, int FMT>
static hipError_t launch_bf3_pair_cfg(const DmaConvArgs& a3, const DmaConvArgs& a1, hipStream_t s) {
    using Cfg = Bf3PairCfg<BM, BN, BK3, BK1, NST, FMT>;
    const int nN = (a1.To + BN - 1) / BN;
    const int S = a1.ksplit > 1 ? a1.ksplit : 1;
    if (S > 1 && (BM * BN > 128 * 64 || a3.ksplit != a1.ksplit || (a3.Ci / BK3) % S || (a1.Ci / BK1) % S)) return hipErrorInvalidValue;
    dim3 grid((a1.Mp / BM) * nN, a1.B * S);
    auto kern = conv_bf3_pair_kernel<BM, BN, BK3, BK1, NST, FMT>;
    if (Cfg::LDS_BYTES > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    snprintf(g_bcfg, sizeof(g_bcfg), "BM%d BN%d KT3+1 S1 U0 BK%d+%d NST%d %s grid %ux%u lds %zu", BM, BN, BK3, BK1, NST, FMT == FMT_F16X2 ? "H3" : "P6", grid.x, grid.y,
             Cfg::LDS_BYTES);
    Bf3PairArgs pp{a3, a1};
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, s, e0, e1, 0, pp);
    else hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, s, pp);
    return hipGetLastError();
}

// fused variant of the pair, 0 = none
static int bf3_pair_variant(const DmaConvArgs& a3, const DmaConvArgs& a1) {
    auto plain = [](const DmaConvArgs& a) {
        return !a.voc && a.stride == 1 && !a.ups && a.epi == EPI_NONE && !a.ln_part && !a.out_f32 && a.Ci % 32 == 0 && a.C1 % 32 == 0 && a.Mp % 64 == 0 && a.B > 0 && a.To > 0;
    };
    if (!plain(a3) || !plain(a1) || a3.KT != 3 || a3.pad != 1 || a1.KT != 1 || a1.pad != 0 || a3.Mp != a1.Mp || a3.To != a1.To || a3.B != a1.B ||
        a3.Tsrc != a1.Tsrc || a1.res || a1.out_plain || a1.plain_from < a1.Cout || a1.lnpart_out)
        return 0;
    int bm, bn, bk, nst;
    bf3_pick(a3, 0, bm, bn, bk, nst);      // the k 3 half carries most of the work: its tile
    const bool k1_64 = (a1.Ci % 64 == 0) && (a1.C1 % 64 == 0);
    (void)k1_64;
    if (bm == 32 && bn == 64 && bk == 32 && nst == 2) return 1;
    if (bm == 64 && bn == 128 && bk == 16 && nst == 2) return 2;
    if (bm == 64 && bn == 64 && bk == 16 && nst == 3) return 3;
    return 0;
}
bool conv_bf3_pair_applies(const DmaConvArgs& a3, const DmaConvArgs& a1) { return bf3_pair_variant(a3, a1) != 0; }

// latency mode: workgroups per output tile (conv_dma.hip cluster_split)
static int bf3_cluster_split(const DmaConvArgs& a, int bm, int bn, int nk, int nk2, int min_steps) {
    if (a.tile_batch <= 0 || !a.kpart || !a.kcount || bm * bn > 128 * 64) return 1;
    const int blk = bm * bn > 64 * 64 ? 2 : 1;
    const long long tiles = (long long)(a.Mp / bm) * ((a.To + bn - 1) / bn) * a.B;
    if (tiles * 4 > a.kcount_cap) return 1;
    int S = 1;
    while (S < 16 && tiles * S * 2 <= 256 && nk % (S * 2) == 0 && nk / (S * 2) >= min_steps && (nk2 == 0 || (nk2 % (S * 2) == 0)) &&
           tiles * 4 * (S * 2) * blk * 1024 <= a.kpart_cap)
        S *= 2;
    return S;
}

template <int FMT>
static hipError_t pair_dispatch(const DmaConvArgs& a3_, const DmaConvArgs& a1_, hipStream_t s) {
    DmaConvArgs a3 = a3_, a1 = a1_;
    {
        const int v = bf3_pair_variant(a3, a1);
        const int bm = v == 1 ? 32 : 64, bn = v == 2 ? 128 : 64, bk3 = v == 1 ? 32 : 16;
        a3.ksplit = a1.ksplit = v ? bf3_cluster_split(a1, bm, bn, a3.Ci / bk3, a1.Ci / 32, 1) : 1;
    }
    switch (bf3_pair_variant(a3, a1)) {      // the 1x1 half runs BK 32 on the k 3 half's tile
        case 1: return launch_bf3_pair_cfg<32, 64, 32, 32, 2, FMT>(a3, a1, s);
        case 2: return launch_bf3_pair_cfg<64, 128, 16, 32, 2, FMT>(a3, a1, s);
        case 3: return launch_bf3_pair_cfg<64, 64, 16, 32, 3, FMT>(a3, a1, s);
        default: return hipErrorNotSupported;
    }
}
hipError_t launch_conv_bf3_pair(const DmaConvArgs& a3, const DmaConvArgs& a1, int fmt, hipStream_t s) {
    return fmt == FMT_F16X2 ? pair_dispatch<FMT_F16X2>(a3, a1, s) : pair_dispatch<FMT_BF16X3>(a3, a1, s);
}

#define BCASE(BM, BN, KT, ST, UP, BK, NS) return launch_bf3_cfg<BM, BN, KT, ST, UP, BK, NS, (FMT == FMT_F16X2 ? 3 : 6), FMT>(a, s)

// cfg = BM*1000000 + BN*1000 + BK*10 + NST (0 = auto); nprod = the format's default (bf16x3: 6, fp16x2: 3) or, on the probe's 128 x 128 x BK32
// tile only, bf16x3: 3 / 9, fp16x2: 4
template <int FMT>
static hipError_t split_dispatch(const DmaConvArgs& a_, int cfg, int nprod, hipStream_t s) {
    DmaConvArgs a = a_;
    a.ksplit = 1;
    if (a.Ci % 16 || a.C1 % 16 || a.Mp % 32 || a.B <= 0 || a.To <= 0 || a.pad < 0 || a.pad > 1 || a.voc) return hipErrorInvalidValue;
    if (a.KT != 1 && a.KT != 3) return hipErrorInvalidValue;
    int bm, bn, bk, nst;
    bf3_pick(a, cfg, bm, bn, bk, nst, FMT);
    const bool k32 = (a.Ci % 32 == 0) && (a.C1 % 32 == 0), k64 = (a.Ci % 64 == 0) && (a.C1 % 64 == 0);
    if ((bk == 64 && !k64) || (bk == 32 && !k32)) return hipErrorInvalidValue;
    if (a.epi == EPI_GEGLU && bm != 128) return hipErrorInvalidValue;
    if (a.Mp % bm) return hipErrorInvalidValue;
    if (a.gnf_part) {
        // GroupNorm fold: a 1x1 convolution over one source; every wave's share of a K-step inside one group (conv_dma.hip launch_conv_dma)
        if (a.KT != 1 || a.stride != 1 || a.ups || a.C2 != 0 || a.x2 || a.ln_part || a.epi != EPI_NONE || a.bias || !a.gnf_cg || !a.gnf_c2 || a.gnf_groups < 1 ||
            a.gnf_groups > 8 || a.Ci % a.gnf_groups || (a.Ci / a.gnf_groups) % 16 || nprod != (FMT == FMT_F16X2 ? 3 : 6))
            return hipErrorInvalidValue;
        const int gsz = a.Ci / a.gnf_groups;
        if (bm * bn > 128 * 64) bn = 64;
        while (gsz % (bm == 32 ? bk / 2 : bk)) bk /= 2;
        if (bm == 32) nst = 2;
        else if (bm == 64) nst = 3;
        else nst = (bk == 32) ? 2 : 3;
        if (cfg == 0) a.ksplit = bf3_cluster_split(a, bm, bn, a.Ci / bk, 0, 2);
#define GBCASE(BM, BN, BK, NS) if (bm == BM && bn == BN && bk == BK && nst == NS) return launch_bf3_cfg<BM, BN, 1, 1, false, BK, NS, (FMT == FMT_F16X2 ? 3 : 6), FMT, true>(a, s)
        GBCASE(32, 64, 32, 2); GBCASE(32, 64, 64, 2); GBCASE(64, 64, 32, 3); GBCASE(64, 64, 16, 3); GBCASE(128, 64, 32, 2); GBCASE(128, 64, 16, 3);
#undef GBCASE
        return hipErrorInvalidValue;
    }
    if (cfg == 0) a.ksplit = bf3_cluster_split(a, bm, bn, a.Ci / bk, 0, 2);
    const int key = a.KT * 100 + a.stride * 10 + (a.ups ? 1 : 0);
    const int tk = bm * 1000 + bn;
    if (nprod != (FMT == FMT_F16X2 ? 3 : 6)) {
        if (key == 110 && tk == 128128 && bk == 32 && nst == 2) {
            if constexpr (FMT == FMT_F16X2) {
                if (nprod == 4) return launch_bf3_cfg<128, 128, 1, 1, false, 32, 2, 4, FMT>(a, s);
            } else {
                if (nprod == 3) return launch_bf3_cfg<128, 128, 1, 1, false, 32, 2, 3, FMT>(a, s);
                if (nprod == 9) return launch_bf3_cfg<128, 128, 1, 1, false, 32, 2, 9, FMT>(a, s);
            }
        }
        return hipErrorInvalidValue;
    }
    if (key == 110) {
        if (tk == 128128) {
            if (bk == 32 && nst == 2) BCASE(128, 128, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) BCASE(128, 128, 1, 1, false, 32, 3);
            if (bk == 16 && nst == 3) BCASE(128, 128, 1, 1, false, 16, 3);
            if (bk == 16 && nst == 4) BCASE(128, 128, 1, 1, false, 16, 4);
        } else if (tk == 128064) {
            if (bk == 32 && nst == 2) BCASE(128, 64, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) BCASE(128, 64, 1, 1, false, 32, 3);
            if (bk == 64 && nst == 2) BCASE(128, 64, 1, 1, false, 64, 2);
            if (bk == 16 && nst == 3) BCASE(128, 64, 1, 1, false, 16, 3);
        } else if (tk == 64064) {
            if (bk == 64 && nst == 2) BCASE(64, 64, 1, 1, false, 64, 2);
            if (bk == 64 && nst == 3) BCASE(64, 64, 1, 1, false, 64, 3);
            if (bk == 32 && nst == 2) BCASE(64, 64, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) BCASE(64, 64, 1, 1, false, 32, 3);
            if (bk == 32 && nst == 4) BCASE(64, 64, 1, 1, false, 32, 4);
            if (bk == 16 && nst == 3) BCASE(64, 64, 1, 1, false, 16, 3);
            if (bk == 16 && nst == 4) BCASE(64, 64, 1, 1, false, 16, 4);
        } else if (tk == 32064) {
            if (bk == 64 && nst == 2) BCASE(32, 64, 1, 1, false, 64, 2);
            if (bk == 64 && nst == 3) BCASE(32, 64, 1, 1, false, 64, 3);
            if (bk == 32 && nst == 2) BCASE(32, 64, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) BCASE(32, 64, 1, 1, false, 32, 3);
        }
    } else if (key == 310) {
        if (tk == 128128) {
            if (bk == 16 && nst == 2) BCASE(128, 128, 3, 1, false, 16, 2);
        } else if (tk == 128064) {
            if (bk == 16 && nst == 2) BCASE(128, 64, 3, 1, false, 16, 2);
        } else if (tk == 64128) {
            if (bk == 16 && nst == 2) BCASE(64, 128, 3, 1, false, 16, 2);
            if (bk == 16 && nst == 3) BCASE(64, 128, 3, 1, false, 16, 3);
        } else if (tk == 64064) {
            if (bk == 32 && nst == 2) BCASE(64, 64, 3, 1, false, 32, 2);
            if (bk == 16 && nst == 2) BCASE(64, 64, 3, 1, false, 16, 2);
            if (bk == 16 && nst == 3) BCASE(64, 64, 3, 1, false, 16, 3);
            if (bk == 16 && nst == 4) BCASE(64, 64, 3, 1, false, 16, 4);
        } else if (tk == 32064) {
            if (bk == 32 && nst == 2) BCASE(32, 64, 3, 1, false, 32, 2);
            if (bk == 32 && nst == 3) BCASE(32, 64, 3, 1, false, 32, 3);
        }
    } else if (key == 320 && tk == 64064) {
        if (bk == 32 && nst == 2) BCASE(64, 64, 3, 2, false, 32, 2);
        if (bk == 16 && nst == 2) BCASE(64, 64, 3, 2, false, 16, 2);
        if (bk == 16 && nst == 3) BCASE(64, 64, 3, 2, false, 16, 3);
    } else if (key == 311 && tk == 64064) {
        if (bk == 32 && nst == 2) BCASE(64, 64, 3, 1, true, 32, 2);
        if (bk == 16 && nst == 2) BCASE(64, 64, 3, 1, true, 16, 2);
        if (bk == 16 && nst == 3) BCASE(64, 64, 3, 1, true, 16, 3);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_conv_bf3(const DmaConvArgs& a, int cfg, int nprod, int fmt, hipStream_t s) {
    if (fmt == FMT_F16X2) return split_dispatch<FMT_F16X2>(a, cfg, nprod ? nprod : 3, s);
    return split_dispatch<FMT_BF16X3>(a, cfg, nprod ? nprod : 6, s);
}

}  // namespace lds
