// DMA-fed, VALU-free implicit-GEMM convolution for the UNet (gfx950, v_mfma_f32_32x32x2_f32).
//
// Why this kernel exists.  On gfx950 the fp32 MFMA executes on the SIMD's vector ALU: measured with
// tools/ubench/mfma_valu_coexec.hip, an MFMA-only wave and a VALU-only wave on one SIMD take the SUM of
// their times (154.7 TFLOP/s alone; every co-resident FMA / exp / integer instruction adds its full issue
// time).  So in an fp32 GEMM every VALU instruction -- address arithmetic, normalisation, SiLU, bounds
// selects, register->LDS staging -- is paid in matrix throughput, whichever wave issues it.  This kernel
// therefore contains no per-element VALU work at all:
//   * activations live in HBM in the "K4P" layout (k4p.h): [B][C/8][2][T+2][4], channel 8q+2j+h at element j of
//     row (q,h), one zero pad frame on both sides.  A k-interleaved row is byte-for-byte the LDS image the MFMA
//     wants (one ds_read_b128 = the B operands of four consecutive MFMAs), and the pad frames ARE the
//     convolution's zero padding, so a tile is a plain linear copy;
//   * both operand tiles are moved HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//     ds_write, loop-invariant per-lane offsets, one scalar pointer bump per K-step;
//   * normalisation + activation are applied ONCE per tensor by the small memory-bound kernels in k4p_ops.hip
//     (instead of once per output-channel block inside the GEMM);
//   * the k3 taps are shifted reads of the same LDS window; skip-concat = second source pointer selected per
//     K-step; nearest-2x upsample = (frame >> 1) in the operand address; stride 2 = operand stride.
// Pipeline: NST-stage LDS ring; every wave issues 1/4 of each tile's DMA.  The operand registers form a ring that is
// read NB-1 MFMA groups ahead and runs THROUGH the K-step boundary: the per-tile synchronisation (counted vmcnt for
// this wave's share of the next tile, one s_barrier, refill of the stage just finished) sits NB-1 groups before the
// end of a K-step, where the ring starts reading the next tile, so no K-step begins with an exposed LDS latency.
#include "gn_chan.h"
#include "k4p.h"
#include "kernels.h"

#include <hip/hip_ext.h>

#include <stdio.h>
#include <stdlib.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf for the GEGLU epilogue (reference activations.py GEGLU -> F.gelu, exact form).  Abramowitz & Stegun 7.1.26:
// |error| <= 1.5e-7 absolute, i.e. at the level of fp32 rounding of erf itself; one v_rcp_f32, one v_exp_f32 and seven
// multiply-adds instead of the ~40-instruction library routine -- the epilogue's vector work is paid in matrix time.
static __device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return copysignf(fmaf(-pl * t, e, 1.0f), x);
}

// Wave64 sum in a fixed order (DPP row shifts + row broadcasts, no LDS): the total is valid in lane 63.
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
static __device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xf, true>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf, true>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf, true>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf, true>(v);      // row_shr:8   lane 15 of every 16-lane row = the row's sum
    v = dpp_add<0x142, 0xa, false>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc, false>(v);     // row_bcast:31 into rows 2 and 3
    return v;
}

template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int DIL = 1>
struct DmaCfg {
    // BM = 32: split-K inside the workgroup for grids smaller than the chip.  The tile is 32 x 64; waves (wn, ks) take
    // column block wn and half ks of every K-step's k range, and the two partial sums meet through LDS in the epilogue
    // (fixed order: ks = 0 adds the ks = 1 partial), so a 64 x 64 tile's work spreads over two workgroups / CUs.
    static constexpr bool SPLIT = BM == 32;
    static constexpr int TM = SPLIT ? 1 : BM / 64, TN = BN / 64;
    static constexpr int KR = BK / 4;                                   // staged k-rows per K-step
    static constexpr int XW = UPS ? (BN / 2 + 2) : ((BN - 1) * STRIDE + (KT - 1) * DIL + 1);   // window entries (frames) per k-row; taps are DIL entries apart
    static constexpr int WI = KT * KR * BM / 64;                        // weight DMA wave-instructions per tile
    static constexpr int WPW = WI / 4;                                  // ... per wave
    static constexpr int RPW = KR / 4;                                  // activation k-rows staged by each wave
    static constexpr int NXI = (RPW * XW + 63) / 64;                    // activation DMA wave-instructions per wave
    static constexpr int PER_TILE = WPW + NXI;                          // VMEM ops per wave per tile
    static constexpr int KQW = SPLIT ? BK / 16 : BK / 8;                // 8-channel k groups per tap handled by one wave
    static constexpr int G = KT * KQW;                                  // MFMA groups per K-step and wave (4 k-pairs each)
    static constexpr int NB = (G % 3 == 0 && TM * TN < 4) ? 3 : 2;      // operand ring depth (divides G: slots keep their phase across K-steps)
    static constexpr int NACC = (TM * TN >= 2) ? 1 : 2;                 // independent accumulator chains per tile
    static constexpr int STAGE = KT * BK * BM + KR * XW * 4;            // floats
    static constexpr size_t LDS_BYTES = (size_t)NST * STAGE * sizeof(float);
    // workgroups per CU allowed by LDS = waves per SIMD to ask the register allocator for (each workgroup puts one wave on each SIMD)
    static constexpr int OCC_LDS = (int)((160 * 1024) / LDS_BYTES);
    static constexpr int OCC = (TM * TN >= 4) ? (OCC_LDS < 2 ? OCC_LDS : 2) : (OCC_LDS < 4 ? (OCC_LDS < 1 ? 1 : OCC_LDS) : 4);
    static_assert(WI % 4 == 0 && KR % 4 == 0, "tile does not split evenly over 4 waves");
    static_assert(!SPLIT || (BN == 64 && BK % 16 == 0), "split-K tile is 32 x 64");
};

// GNF: the GroupNorm fold of DmaConvArgs::gnf_part (its own instantiations: the other launches carry none of its registers)
template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int DIL = 1, bool VOC = false, bool GNF = false>
struct DmaKernel {
    using Cfg = DmaCfg<BM, BN, KT, STRIDE, UPS, BK, NST, DIL>;
    static constexpr int TM = Cfg::TM, TN = Cfg::TN, KR = Cfg::KR, XW = Cfg::XW, WPW = Cfg::WPW, RPW = Cfg::RPW, NXI = Cfg::NXI;
    static constexpr int G = Cfg::G, NB = Cfg::NB, NACC = Cfg::NACC, STAGE = Cfg::STAGE, PER_TILE = Cfg::PER_TILE, KQW = Cfg::KQW;
    static constexpr bool SPLIT = Cfg::SPLIT;

    const DmaConvArgs& p;
    float* smem;
    int lane, wave, c, h, wm, wn, ks, b, m0, t0;
    int ksp, kc0, ctile;      // cluster split-K (DmaConvArgs::ksplit): this workgroup's share index, its first K-step, its tile's linear index
    int Lout;                 // valid output frames of this batch element (ragged batches, k4p.h ragged_len; = To otherwise)
    float gcg[GNF ? 8 : 1], gkc;      // (threads < BM: the terms of their row's constant, in flight across the first barrier)
    GnPart gp0, gp1;                  // this wave's two groups' partial statistics, in flight across the first DMAs
    float gs, gsn, ginv;              // rstd of the group of this wave's share of the current / the next K-step; 1 / (channels per group)
    int kofs_a, kofs_b;       // split-K: float offsets of this wave's half of the staged k rows (weights / activations)
    int woff[WPW];            // per-lane byte offsets of this wave's weight chunks (loop invariant)
    int xoff[NXI];            // per-lane byte offsets of this wave's activation chunks inside a source slab
    __amdgpu_buffer_rsrc_t rw, rx1, rx2;   // packed weights; this batch element's slab of either source
    bool xact[NXI];
    int arow;
    int bcol[TN];
    f32x16 acc[NACC][TM][TN];
    f32x4 aop[NB][TM], bop[NB][TN];
    float lmu[TN], lrs[TN];   // folded input-LayerNorm statistics of this lane's output columns
    // single-tile waves fetch the epilogue's operands at kernel start (16 + 16 registers): the residual entries and the bias
    // rows have landed long before the main loop ends, so the epilogue begins without a memory round trip
    // (vocoder tiles too: their K loops run 20-40 us, the 8 residual registers per 32 x 32 tile fit beside the accumulators)
    static constexpr bool EARLY = TM * TN == 1 || VOC;
    f32x2 rsv[EARLY ? TM * TN * 8 : 1];
    float kbv[EARLY ? TM * 16 : 1];

    __device__ __forceinline__ DmaKernel(const DmaConvArgs& p_, float* s_) : p(p_), smem(s_) {}

    __device__ __forceinline__ void setup_keep_acc() {      // everything but the accumulators (second phase of conv_dma_pair_kernel)
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        c = lane & 31; h = lane >> 5;
        wm = SPLIT ? 0 : wave >> 1; ks = SPLIT ? wave >> 1 : 0; wn = wave & 1;
        kofs_a = ks * KQW * 2 * BM * 4; kofs_b = ks * KQW * 2 * XW * 4;
        // XCD-aware tile order.  Workgroups are dispatched in linear order, round-robin over the 8 XCDs, each with a private
        // L2.  The workgroups that read the same activation window (all M-blocks of one (batch, frame-block)) are given
        // consecutive slots of ONE XCD, so the window is fetched from the fabric once instead of once per XCD.
        const int nMb = p.Mp / BM;
        const int gx = gridDim.x, total = gx * gridDim.y;
        const int id = blockIdx.y * gx + blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, per = total >> 3, rem = total & 7;
        const int L = xcd * per + (xcd < rem ? xcd : rem) + slot;
        const int mb = L % nMb;                   // M fastest: neighbours share the activation tile
        const int tb = L / nMb;
        const int nN = gx / nMb;
        const int nb = tb % nN;
        const int by = tb / nN, S = p.ksplit > 1 ? p.ksplit : 1;      // grid.y = B * S
        b = by / S; ksp = by - b * S;
        kc0 = ksp * (p.Ci / BK / S);
        ctile = (b * nN + nb) * nMb + mb;
        m0 = mb * BM; t0 = nb * BN;
        Lout = ragged_len(p.lens, b, p.lvl_out, p.To);
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int q = (wave + 4 * i) * 64 + lane;           // chunk = (tap, k-row, m)
            const int tap = q / (KR * BM);
            const int rem = q - tap * (KR * BM);
            const int rr = rem / BM, m = rem - rr * BM;
            woff[i] = ((tap * (p.Ci / 4) + rr) * p.Mp + m0 + m) * 16;
        }
        const int Tp = p.Tsrc + 2 * p.xpad;
        const int e0 = UPS ? (((t0 - 1) >> 1) + p.xpad) : (t0 * STRIDE - p.pad + p.xpad);   // first window entry (entry xpad = frame 0)
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int qq = i * 64 + lane;
            const int rl = qq / XW, col = qq - rl * XW;
            xact[i] = qq < RPW * XW;
            xoff[i] = ((wave * RPW + rl) * Tp + e0 + col) * 16;
        }
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, KT * p.Ci * p.Mp * 4, 0x00020000);
        rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long long)b * p.C1 * Tp), 0, p.C1 * Tp * 4, 0x00020000);
        rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 ? p.x2 + (long long)b * p.C2 * Tp : p.x1), 0, p.C2 * Tp * 4, 0x00020000);
        arow = wm * TM * 32 + c;
#pragma unroll
        for (int j = 0; j < TN; ++j) bcol[j] = wn * TN * 32 + j * 32 + c;
    }
    __device__ __forceinline__ bool dead_tile() const { return !VOC && p.lens && t0 >= Lout; }
    // the vocoder's lengths multiply per stage: given outright (kernels.h DmaConvArgs::vlen), fetched where they are used (the epilogue)
    __device__ __forceinline__ int voc_len() const { return p.vlen[b]; }
    __device__ __forceinline__ void setup() {
        setup_keep_acc();
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    }

    // tile kc -> LDS stage `st`: this wave's share of the weight chunks and of the activation window.  Buffer-addressed
    // LDS-DMA: per-lane byte offsets are loop invariant (VGPR), the K-step advance is a scalar offset, so issuing a tile
    // costs no vector ALU work; reads past a source slab return zeros (hardware range check).
    __device__ __forceinline__ void issue_tile(int kc, float* st) {
        const int ws = kc * (KR * 16) * p.Mp;                    // bytes
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + ((wave + 4 * i) * 64) * 4), 16, woff[i], ws, 0, 0);
        const int k0 = kc * BK;
        const bool s2 = k0 >= p.C1;                              // C1 % BK == 0: a K-step reads one source only
        const int xsoff = (s2 ? k0 - p.C1 : k0) * (p.Tsrc + 2 * p.xpad) * 4;
        float* xs = st + KT * BK * BM + wave * RPW * XW * 4;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            if (xact[i]) {
                if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (__attribute__((address_space(3))) void*)(xs + i * 64 * 4), 16, xoff[i], xsoff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (__attribute__((address_space(3))) void*)(xs + i * 64 * 4), 16, xoff[i], xsoff, 0, 0);
            }
    }

    template <int SLOT>
    __device__ __forceinline__ void load_ops(const float* st, int tap, int kq) {
        const float* wt = st + ((tap * KR + kq * 2 + h) * BM + arow) * 4 + kofs_a;
        const float* xs = st + KT * BK * BM + (kq * 2 + h) * XW * 4 + kofs_b;
#pragma unroll
        for (int i = 0; i < TM; ++i) aop[SLOT][i] = *reinterpret_cast<const f32x4*>(wt + i * 32 * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = UPS ? (((t0 + bcol[j] + tap - 1) >> 1) - ((t0 - 1) >> 1)) : (bcol[j] * STRIDE + tap * DIL);
            bop[SLOT][j] = *reinterpret_cast<const f32x4*>(xs + col * 4);
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void mfma_ops() {
        if constexpr (GNF) {      // GroupNorm fold: the activations of this K-step's group times its rstd (2 packed multiplies per operand entry)
#pragma unroll
            for (int j = 0; j < TN; ++j) bop[SLOT][j] *= gs;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[jj % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[SLOT][i][jj], bop[SLOT][j][jj], acc[jj % NACC][i][j], 0, 0, 0);
    }
    template <int g0, int g1>
    __device__ __forceinline__ void preload(const float* cur) {
        if constexpr (g0 < g1 && g0 < G) {
            load_ops<g0 % NB>(cur, g0 / KQW, g0 % KQW);
            preload<g0 + 1, g1>(cur);
        }
    }
    // Tile boundary, reached when the operand ring is about to read its first group of tile kc+1: this wave has issued
    // (and, after the lgkmcnt wait, completed) all its reads of tile kc.  After the barrier tile kc+1 is visible to
    // everyone and tile kc's stage is free, so it is refilled with tile kc+NST.
    __device__ __forceinline__ void tile_sync(int kc, int nk, float* cur) {
        const int younger = (nk - 2 - kc < NST - 2) ? (nk - 2 - kc) : (NST - 2);     // tiles after kc+1 still in flight
        wait_younger<NST - 2>(younger);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kc + NST < nk) issue_tile(kc0 + kc + NST, cur);
    }
    template <int g>
    __device__ __forceinline__ void kstep(float* cur, const float* nxt, int kc, int nk) {
        if constexpr (g < G) {
            constexpr int gp = g + NB - 1;                     // group whose operands are fetched now
            if constexpr (gp < G) {
                load_ops<gp % NB>(cur, gp / KQW, gp % KQW);
            } else {
                if (kc + 1 < nk) {
                    if constexpr (gp == G) tile_sync(kc, nk, cur);
                    load_ops<gp % NB>(nxt, (gp - G) / KQW, (gp - G) % KQW);
                }
            }
            mfma_ops<g % NB>();
            __builtin_amdgcn_sched_barrier(0);     // keep the operand reads NB-1 groups ahead of their MFMAs
            kstep<g + 1>(cur, nxt, kc, nk);
        }
    }

    // LayerNorm over the input channels (reference attention.py:83,102,118), folded into the epilogue: per output column
    // combine the producer's per-32-channel (mean, M2) partials in a fixed order (gn_chan.h ln_column_stats).  Called right
    // after the first tiles' DMAs are issued; all partials of a column are requested at once (no chain of dependent round trips).
    __device__ __forceinline__ void ln_columns() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = t0 + wn * TN * 32 + j * 32 + c;
            const bool ok = n < p.To;
            const float2* src = p.ln_part + (long long)b * p.ln_np * p.Tsrc + (ok ? n : 0);
            ln_column_stats(src, p.Tsrc, p.ln_np, p.ln_eps, ok, lmu[j], lrs[j]);      // (gn_chan.h: two plain sums in a fixed order)
        }
    }

    // ---- GroupNorm fold (DmaConvArgs::gnf_part) ----
    // LDS behind the ring: [0, BM) the per-row constants, then 8 x rstd_g and 8 x rstd_g * mean_g of this batch element
    __device__ __forceinline__ float* gnf_tail() const { return smem + NST * STAGE + BM; }
    // group of this wave's share of K-step k (of this workgroup's range), clamped to the last one (look-ahead past the end)
    __device__ __forceinline__ int gnf_group(int k) const {
        const int pos = (kc0 + k) * BK + ks * (SPLIT ? BK / 2 : BK);
        const int g = (int)(((float)pos + 0.5f) * ginv);
        return g < p.gnf_groups ? g : p.gnf_groups - 1;
    }
    // kernel start, before the first DMA (so that waiting for them does not wait for the tiles): the requests for this wave's two groups'
    // partials (wave w takes groups w and w + 4) and for the terms of the row constants
    __device__ __forceinline__ void gnf_request() {
        const int G_ = p.gnf_groups, gsz = p.Ci / G_;
        const int gTv = ragged_len(p.lens, b, p.lvl_in, p.Tsrc);      // the statistics stop at the utterance's own length
        ginv = 1.0f / (float)gsz;
        gp0 = gn_part_load(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, wave, lane, 0, gTv);                              // (8 groups at most: wave < G_ or an unused slot)
        gp1 = gn_part_load(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, (wave + 4 < G_) ? wave + 4 : wave, lane, 0, gTv);
        if ((int)threadIdx.x < BM) {
            const int m = m0 + (int)threadIdx.x;
            gkc = p.gnf_c2[m];
#pragma unroll
            for (int g = 0; g < 8; ++g) gcg[g] = (g < G_) ? p.gnf_cg[g * p.Mp + m] : 0.f;
        }
    }
    // after the first tiles' DMAs are issued, before the first barrier: the statistics into LDS
    __device__ __forceinline__ void gnf_prepare() {
        const int G_ = p.gnf_groups, gsz = p.Ci / G_;
        float* tail = gnf_tail();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int g = wave + 4 * e;
            float rs = 1.f, rm = 0.f;      // (slots of absent groups: finite, they multiply zeros)
            if (g < G_) {
                float mu, var;
                gnf_group_stats(p.gnf_part, b, p.Ci, p.Tsrc, gsz >> 4, g, lane, e ? gp1 : gp0, mu, var, ragged_len(p.lens, b, p.lvl_in, p.Tsrc));
                rs = __builtin_amdgcn_rsqf(var + p.gnf_eps); rm = rs * mu;
            }
            if (lane == 0) { tail[g] = rs; tail[8 + g] = rm; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the stores are in LDS before this wave passes the barrier (s_barrier alone does not wait)
    }
    // after the first barrier: kc[m] = gnf_c2[m] - sum_g rstd_g mean_g gnf_cg[g][m]  (read in finalize, behind further barriers), and the
    // scale of this wave's first two K-steps
    __device__ __forceinline__ void gnf_rows() {
        const float* tail = gnf_tail();
        if ((int)threadIdx.x < BM) {
            float kc = gkc;
#pragma unroll
            for (int g = 0; g < 8; ++g) kc = fmaf(-tail[8 + g], gcg[g], kc);      // (groups beyond gnf_groups: gcg = 0, tail finite)
            smem[NST * STAGE + threadIdx.x] = kc;
        }
        gs = tail[gnf_group(0)];
        gsn = tail[gnf_group(1)];
    }
    // after K-step k: the next step's scale becomes current, the one after it is requested (its LDS latency hides under the next step)
    __device__ __forceinline__ void gnf_advance(int k) {
        gs = gsn;
        gsn = gnf_tail()[gnf_group(k + 2)];
    }

    // s_waitcnt vmcnt(y * PER_TILE) for a wave-uniform y in [0, Y]: the y younger tiles' DMAs may stay in flight
    template <int Y>
    __device__ __forceinline__ void wait_younger(int y) {
        if constexpr (Y == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            constexpr int N = (Y * PER_TILE > 63) ? 63 : Y * PER_TILE;      // vmcnt is a 6-bit field; a smaller count only waits longer
            if (y >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
            else wait_younger<Y - 1>(y);
        }
    }

    __device__ __forceinline__ bool res_tile(int tile0, int n) const {
        return p.res && tile0 < p.plain_from && tile0 < p.Cout && n < p.To;      // the residual is always a K4P tensor
    }
    __device__ __forceinline__ void early_loads() {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int tile0 = m0 + wm * TM * 32 + i * 32;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = t0 + wn * TN * 32 + j * 32 + c;
                f32x2* rs = rsv + (i * TN + j) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = f32x2{0.f, 0.f};
                if (res_tile(tile0, n)) {
                    const int Tpo = p.To + 2 * p.opad;
                    const float* rb = p.res + (long long)b * k4p_ck() * Tpo + k4p_off(tile0, n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) rs[e] = *reinterpret_cast<const f32x2*>(rb + e * Tpo * 4);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) kbv[i * 16 + r] = (p.bias && !p.ln_part) ? p.bias[tile0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
        }
    }

    __device__ __forceinline__ void mainloop() {
        static_assert(G % NB == 0 && NB - 1 <= G, "ring slots must keep their phase across K-steps");
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
        preload<0, NB - 1>(smem);
        int sc = 0;
        for (int kc = 0; kc < nk; ++kc) {
            const int sn = (sc + 1 == NST) ? 0 : sc + 1;
            kstep<0>(smem + sc * STAGE, smem + sn * STAGE, kc, nk);
            sc = sn;
            if constexpr (GNF) gnf_advance(kc);
        }
        if constexpr (GNF) __syncthreads();      // (the row constants are read in finalize: a one-K-step launch has met no barrier since)
    }

    // Epilogue, phase 1: accumulators -> output values in place (acc[0]).  All loads of a phase are issued before their
    // first use and before any store: a load placed after a store cannot be moved above it (possible aliasing), and a
    // load -> wait -> store chain costs one memory round trip per element.
    __device__ __forceinline__ void finalize(bool geglu) {
        if constexpr (NACC == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][i][j] += acc[1][i][j];
        }
        const bool ln = p.ln_part != nullptr;
        if (EARLY && !ln) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] += kbv[i * 16 + r];
        } else if (ln || p.bias) {
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
                        const float v = acc[0][i][j][r];
                        acc[0][i][j][r] = ln ? lrs[j] * (v - lmu[j] * k1[i][r]) + k2[i][r] : v + k2[i][r];
                    }
        }
        if constexpr (GNF) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float kc = smem[NST * STAGE + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[0][i][j][r] += kc;
                }
        }
        if (geglu) {      // rows of tile 0 are the values, rows of tile 1 the gates (pack_geglu)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = acc[0][TM - 1][j][r];
                    acc[0][0][j][r] *= 0.5f * g * (1.0f + erf_fast(g * 0.70710678118654752440f));
                }
        }
    }

    // frame-major store of one 32x32 tile: out[b][co][n], co = c0 + local row
    __device__ __forceinline__ void store_plain(float* base, int Cn, int c0, int i, int j, int n) {
        if (n >= p.To) return;
        float* ob = base + ((long long)b * Cn + c0) * p.To + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[0][i][j][r];
            if constexpr (VOC) {
                if (p.out_div != 1.0f) v = v / p.out_div;
                if (p.act_slope != 0.f) v = (v >= 0.f) ? v : v * p.act_slope;
                if constexpr (BN < 256) {      // (ragged batch: zeros beyond the utterance's length; the launcher keeps ragged calls off the 256-wide tile, which has no register to spare)
                    if (p.vlen && n >= voc_len()) v = 0.f;
                }
            }
            if (c0 + rl < Cn) ob[rl * p.To] = v;
        }
    }

    // attention's VT layout for the value channels (attention_k4p.hip): [B][head][ceil(To/4)][D][4], element = V[d] at 4
    // consecutive frames; frames To .. ceil4(To)-1 are written as zeros (they are multiplied by zero probabilities)
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
            if (c0 + rl < Cv) ob[(long long)head * D * T4 + d * 4] = real ? acc[0][i][j][r] : 0.f;
        }
    }

    // K4P addressing of one 32x32 tile.  Registers (4g+hh, 4g+2+hh) of this lane are elements (2h, 2h+1) of row
    // (q = tile0/8 + g, hh): two 8-byte accesses per 8-channel block; the two lane halves together fill the 16-byte entry,
    // and consecutive lanes are consecutive frames.
    __device__ __forceinline__ int k4p_off(int tile0, int n) const { return (((tile0 >> 3) * 2) * (p.To + 2 * p.opad) + n + p.opad) * 4 + 2 * h; }
    __device__ __forceinline__ int k4p_ck() const { return (p.plain_from < p.Cout) ? p.plain_from : p.Cout; }

    // phase 2: residual add (all loads in flight together, before any store)
    __device__ __forceinline__ void add_residual(int tile0, int i, int j, int n) {
        if (n >= p.To) return;
        f32x2 rv[8];
        if constexpr (EARLY) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = rsv[(i * TN + j) * 8 + e];
        } else {
            const int Tpo = p.To + 2 * p.opad;
            const float* rb = p.res + (long long)b * k4p_ck() * Tpo + k4p_off(tile0, n);
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = *reinterpret_cast<const f32x2*>(rb + e * Tpo * 4);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                acc[0][i][j][4 * g + hh] += rv[2 * g + hh][0];
                acc[0][i][j][4 * g + 2 + hh] += rv[2 * g + hh][1];
            }
    }

    // phase 2b (vocoder): the MRF running sum, a K4P tensor with the output's channels / frames / padding whatever the output layout is;
    // its loads for all tiles of the wave are in flight together, before any store
    __device__ __forceinline__ void add_running_sum(int tile0, int i, int j, int n) {
        if (n >= p.To) return;
        const int Tpo = p.To + 2 * p.opad;
        const float* ab = p.acc_in + (long long)b * (p.out_plain ? p.Cout : k4p_ck()) * Tpo + k4p_off(tile0, n);
        f32x2 av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = *reinterpret_cast<const f32x2*>(ab + e * Tpo * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                acc[0][i][j][4 * g + hh] += av[2 * g + hh][0];
                acc[0][i][j][4 * g + 2 + hh] += av[2 * g + hh][1];
            }
    }

    // polyphase ConvTranspose store (vocoder upsamplers): row m = co * phases + phase, column n -> frame n * phases + phase - tpad of
    // channel co; the raw value goes to `out` and its LeakyReLU to `out_act` (the next resblocks' residual and first input).  A
    // workgroup's 64 rows x 128 columns cover whole 64-byte lines between them (8 channels x 4 frames per line), written back from L2.
    __device__ __forceinline__ void store_phases(int tile0, int i, int j, int n) {
        const int lg = p.ph_log2, Tpo = p.ph_Tout + 2 * p.opad;
        const long long ob = (long long)b * p.ph_Cout * Tpo;
        const int Lv = p.vlen ? voc_len() : 0x7fffffff;      // ragged batch: zeros beyond the utterance's output length
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = tile0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = m >> lg, f = (n << lg) + (m & ((1 << lg) - 1)) - p.ph_tpad;
            const bool ok = n < p.To && f >= 0 && f < p.ph_Tout && co < p.ph_Cout;
            const long long off = ob + (long long)(((co >> 3) * 2 + (co & 1)) * Tpo + f + p.opad) * 4 + ((co & 7) >> 1);
            const float v = (f < Lv) ? acc[0][i][j][r] : 0.f;
            if (ok) {
                p.out[off] = v;
                if (p.out_act) p.out_act[off] = (v >= 0.f) ? v : v * p.act_slope;
            }
        }
    }

    // phase 3: K4P store of one 32x32 tile (+ pad frames, + GroupNorm / LayerNorm partials).  Vocoder epilogues: the running
    // sum of the MRF (acc_in, out_div), LeakyReLU of the value for the next convolution (act_slope; out_act = second tensor
    // when the raw value is needed too, as a residual).
    __device__ __forceinline__ void store_k4p(int tile0, int i, int j, int n) {
        const int Tpo = p.To + 2 * p.opad;
        const int Ck = k4p_ck();
        const bool ok = n < p.To;
        const long long o0 = (long long)b * Ck * Tpo + k4p_off(tile0, n);
        float* ob = p.out + o0;
        if (ok) {
            if constexpr (VOC) {
            if constexpr (BN < 256) {
                if (p.vlen && n >= voc_len()) {      // ragged batch: this stage's frames beyond the utterance's length are written as zeros
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] = 0.f;
                }
            }
            if (p.out_div != 1.0f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][j][r] = acc[0][i][j][r] / p.out_div;
            }
            if (p.act_slope != 0.f && p.out_act) {      // raw value to `out`, activated value to `out_act`
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) k4p_store_wt(ob + (2 * g + hh) * Tpo * 4, f32x2{acc[0][i][j][4 * g + hh], acc[0][i][j][4 * g + 2 + hh]});
                ob = p.out_act + o0;
            }
            if (p.act_slope != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[0][i][j][r]; acc[0][i][j][r] = (v >= 0.f) ? v : v * p.act_slope; }
            }
            }      // VOC
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    k4p_store_wt(ob + (2 * g + hh) * Tpo * 4, f32x2{acc[0][i][j][4 * g + hh], acc[0][i][j][4 * g + 2 + hh]});
            if (p.opad == 1) {                    // one pad frame per side is written with the tensor; wider pads are zeroed by k4p_zero_pads
                if (n == 0) {                     // left pad frame
#pragma unroll
                    for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x2*>(ob + e * Tpo * 4 - 4) = f32x2{0.f, 0.f};
                }
                if (n == p.To - 1) {              // right pad frame
#pragma unroll
                    for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x2*>(ob + e * Tpo * 4 + 4) = f32x2{0.f, 0.f};
                }
            }
        }
        if (p.gnpart_out) {
            // GroupNorm statistics of the tensor being written, for the consumer's streaming normalisation (k4p_ops.hip
            // gn_stream): per (16 channels x 32 frames) block one (mean, M2) over its valid frames.  Registers 0..7 of a
            // lane are channels tile0 + 0..15, registers 8..15 channels tile0 + 16..31; sums of (x - k) and (x - k)^2 with
            // k = the block's first element (a shift near the mean keeps the sum-of-squares form free of cancellation),
            // reduced over the wave in a fixed order.
            const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0][i][j][0])));
            const float k1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0][i][j][8])));
            float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float d0 = acc[0][i][j][r] - k0, d1 = acc[0][i][j][r + 8] - k1;
                a1 += d0; a2 = fmaf(d0, d0, a2);
                b1 += d1; b2 = fmaf(d1, d1, b2);
            }
            if (n >= Lout) { a1 = 0.f; a2 = 0.f; b1 = 0.f; b2 = 0.f; }      // (Lout <= To: frames beyond the utterance's length carry no statistics)
            a1 = wave_sum_to_lane63(a1); a2 = wave_sum_to_lane63(a2);
            b1 = wave_sum_to_lane63(b1); b2 = wave_sum_to_lane63(b2);
            const int n0 = n - c;                                   // first frame of this 32-frame block
            const int nv = (Lout - n0 < 32) ? Lout - n0 : 32;       // valid frames
            if (lane == 63 && nv > 0) {
                const float cnt = 16.0f * (float)nv, rc = 1.0f / cnt;
                float2* gp = p.gnpart_out + ((long long)b * (Ck >> 4) + (tile0 >> 4)) * ((p.To + 31) >> 5) + (n0 >> 5);
                gp[0] = make_float2(k0 + a1 * rc, fmaxf(a2 - a1 * a1 * rc, 0.f));
                if (tile0 + 16 < Ck) gp[(p.To + 31) >> 5] = make_float2(k1 + b1 * rc, fmaxf(b2 - b1 * b1 * rc, 0.f));
            }
        }
        if (p.lnpart_out) {
            // per-frame (mean, M2) over this tile's 32 channels -> LayerNorm partials (combined by the consumer, ln_columns)
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[0][i][j][r];
            const float m16 = s1 * (1.0f / 16.0f);
            float qv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[0][i][j][r] - m16; qv += d * d; }
            const float mo = __shfl_xor(m16, 32, 64), qo = __shfl_xor(qv, 32, 64);
            const float d = mo - m16;
            if (h == 0 && ok)
                p.lnpart_out[((long long)b * (Ck >> 5) + (tile0 >> 5)) * p.To + n] = make_float2(0.5f * (m16 + mo), (qv + qo) + d * d * 8.0f);
        }
    }

    __device__ __forceinline__ int tile_ch(int i, bool geglu) const {      // first output channel of tile row i
        return geglu ? (m0 + wm * 64) / 2 : (m0 + wm * TM * 32 + i * 32);
    }

    // split-K: the ks = 1 waves hand their partial tile to the ks = 0 wave of the same column block through LDS
    __device__ __forceinline__ bool join_halves() {
        if constexpr (NACC == 2) acc[0][0][0] += acc[1][0][0];
        __syncthreads();                                   // every wave is done reading operand tiles: the stages are free
        float* red = smem + wn * 16 * 64 + lane;
        if (ks == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64] = acc[0][0][0][r];
        }
        __syncthreads();
        if (ks == 1) return false;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][0][r] += red[r * 64];
        if constexpr (NACC == 2) acc[1][0][0] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        return true;
    }

    // Cluster split-K (latency mode, DmaConvArgs::ksplit = S > 1): S workgroups share one output tile, each reducing 1/S of the K range.
    // The hand-off follows ONE row of the hand-off table in MI355X_MICROARCH.md (inter-workgroup visibility, "ONE lane of each storing
    // workgroup, for ALL that workgroup's stores ... all those lanes add to ONE unsharded counter, the workgroup whose add came last, told
    // by the value its add returned"), in every cell:
    //   * every wave stores its partial accumulators with sc1 (write-through) stores, 4 bytes each, and waits for their acknowledgement
    //     (s_waitcnt vmcnt(0)); a workgroup barrier then makes that true of ALL the workgroup's stores;
    //   * one lane of the workgroup adds 1 to the tile's counter (agent-scope atomic, one counter per tile); the add that returns S - 1 came last;
    //   * that lane tells the workgroup's other waves through an LDS word behind a second barrier -- nobody loads a partial before it;
    //   * the last workgroup's waves read all S partials back with sc1 loads (its own included) in the fixed order s = 0 .. S-1, so the sum does
    //     not depend on the order of arrival, and run the epilogue; the others leave.  Nobody ever waits for another workgroup.
    // One workgroup per CU (the launcher keeps a split grid within the chip), memory from hipMalloc.  The counter is put back to zero by the
    // last workgroup after its loads are issued: the next reader of the counter is the next launch.
    __device__ __forceinline__ bool cluster_join() {
        const int S = p.ksplit;
        if constexpr (NACC == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[0][i][j] += acc[1][i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[1][i][j][r] = 0.f;
                }
        }
        const int slot = ctile * 4 + wave;
        float* mine = p.kpart + ((long long)slot * S + ksp) * (TM * TN * 1024) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * 64, acc[0][i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                for (int r = 0; r < 16; ++r) acc[0][i][j][r] = __hip_atomic_load(all + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 1; q < S; ++q) {
            const float* pq = all + (long long)q * (TM * TN * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] += __hip_atomic_load(pq + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) __hip_atomic_store(p.kcount + ctile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }

    __device__ __forceinline__ void epilogue() {
        const bool geglu = (p.epi == EPI_GEGLU) && (TM == 2);
        if constexpr (SPLIT) {
            if (!join_halves()) return;
        }
        if constexpr (TM * TN <= 2 && !VOC) {      // (the launcher asks for a cluster split on waves of one or two blocks only)
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
        if constexpr (VOC) {
            if (p.acc_in) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int tile0 = tile_ch(i, geglu);
                        if (tile0 < p.Cout) add_running_sum(tile0, i, j, t0 + wn * TN * 32 + j * 32 + c);
                    }
            }
        } else {
            if (p.lens) {      // ragged batch: frames at and beyond this utterance's length are written as zeros, in whatever layout
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (t0 + wn * TN * 32 + j * 32 + c >= Lout) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[0][i][j][r] = 0.f;
                    }
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
                if constexpr (VOC && KT == 2) {      // (the polyphase upsamplers are the 2-tap instantiations)
                    if (p.ph_Tout) { store_phases(tile0, i, j, n); continue; }
                }
                if (p.out_plain) store_plain(p.out, p.Cout, tile0, i, j, n);
                else if (tile0 >= p.plain_from) {
                    if (p.vt_D) store_vt(tile0 - p.plain_from, i, j, n);
                    else store_plain(p.out2, p.Cout - p.plain_from, tile0 - p.plain_from, i, j, n);
                }
                else if (tile0 < p.Cout) store_k4p(tile0, i, j, n);
            }
        }
    }
};

template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int DIL = 1, bool VOC = false, bool GNF = false>
// (the polyphase upsampler's scatter epilogue needs more than the 128 registers of 4 workgroups per CU: 2 per CU, no spills)
__global__ void __launch_bounds__(256, ((VOC && KT == 2) ? 2 : DmaCfg<BM, BN, KT, STRIDE, UPS, BK, NST, DIL>::OCC)) conv_dma_kernel(const DmaConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    DmaKernel<BM, BN, KT, STRIDE, UPS, BK, NST, DIL, VOC, GNF> k(p, smem);
    k.setup();
    // ragged batch: a tile that lies wholly beyond its utterance's length has nothing to reduce -- the epilogue writes its zeros (what it
    // computes from the untouched accumulators never reaches memory: masked columns are stored as literal zeros)
    if (!k.dead_tile()) k.mainloop();
    k.epilogue();
}

// Two reductions into one set of accumulators: the k 3 convolution of a resnet's second half, then its 1x1 shortcut over the block
// input (two K4P sources), then ONE epilogue (launch_conv_dma_pair).  The second phase starts its own DMA ring after the first has
// drained (one exposed first-tile latency, ~2 us, against a launch, ~8 us, and the shortcut tensor's round trip through memory).
struct DmaPairArgs { DmaConvArgs a3, a1; };

template <int BM, int BN, int BK3, int BK1, int NST>
struct PairCfg {
    using C3 = DmaCfg<BM, BN, 3, 1, false, BK3, NST, 1>;
    using C1 = DmaCfg<BM, BN, 1, 1, false, BK1, NST, 1>;
    static constexpr size_t LDS_BYTES = C3::LDS_BYTES > C1::LDS_BYTES ? C3::LDS_BYTES : C1::LDS_BYTES;
    static constexpr int OCC = C3::OCC < C1::OCC ? C3::OCC : C1::OCC;
    static_assert(C3::NACC == C1::NACC && C3::TM == C1::TM && C3::TN == C1::TN && C3::SPLIT == C1::SPLIT, "the phases share the accumulators");
};

template <int BM, int BN, int BK3, int BK1, int NST>
__global__ void __launch_bounds__(256, (PairCfg<BM, BN, BK3, BK1, NST>::OCC)) conv_dma_pair_kernel(const DmaPairArgs pp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using K3 = DmaKernel<BM, BN, 3, 1, false, BK3, NST, 1, false>;
    using K1 = DmaKernel<BM, BN, 1, 1, false, BK1, NST, 1, false>;
    K1 k1(pp.a1, smem);
    {
        K3 k3(pp.a3, smem);
        k3.setup();
        if (!k3.dead_tile()) k3.mainloop();
#pragma unroll
        for (int a = 0; a < K3::NACC; ++a)
#pragma unroll
            for (int i = 0; i < K3::TM; ++i)
#pragma unroll
                for (int j = 0; j < K3::TN; ++j) k1.acc[a][i][j] = k3.acc[a][i][j];
    }
    __syncthreads();                // every wave is done reading the first phase's stages
    k1.setup_keep_acc();
    if (!k1.dead_tile()) k1.mainloop();
    k1.epilogue();
}

static thread_local char g_dcfg[96] = "";
const char* conv_dma_last_config() { return g_dcfg; }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int BK, int NST, int DIL = 1, bool VOC = false, bool GNF = false>
static hipError_t launch_dma_cfg(const DmaConvArgs& a, hipStream_t s) {
    using Cfg = DmaCfg<BM, BN, KT, STRIDE, UPS, BK, NST, DIL>;
    if (GNF != (a.gnf_part != nullptr)) return hipErrorInvalidValue;
    const size_t lds_bytes = Cfg::LDS_BYTES + (GNF ? (BM + 32) * sizeof(float) : 0);      // the fold's row constants and group statistics sit behind the ring
    const int nN = (a.To + BN - 1) / BN;
    const int S = a.ksplit > 1 ? a.ksplit : 1;
    if (S > 1 && (Cfg::TM * Cfg::TN > 2 || VOC || (a.Ci / BK) % S)) return hipErrorInvalidValue;
    dim3 grid((a.Mp / BM) * nN, a.B * S);
    auto kern = conv_dma_kernel<BM, BN, KT, STRIDE, UPS, BK, NST, DIL, VOC, GNF>;
    if (lds_bytes > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    const char* gtag = GNF ? " GNF" : "";
    if (!VOC && S > 1) snprintf(g_dcfg, sizeof(g_dcfg), "BM%d BN%d KT%d S%d U%d BK%d NST%d KS%d%s grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, NST, S, gtag, grid.x, grid.y, lds_bytes);
    else if (!VOC) snprintf(g_dcfg, sizeof(g_dcfg), "BM%d BN%d KT%d S%d U%d BK%d NST%d%s grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, NST, gtag, grid.x, grid.y, lds_bytes);
    else snprintf(g_dcfg, sizeof(g_dcfg), "BM%d BN%d KT%d S%d U%d BK%d NST%d D%d grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, NST, DIL, grid.x, grid.y, Cfg::LDS_BYTES);
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, e0, e1, 0, a);      // bench.py's roofline leg
    else hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

#define DCASE(BM, BN, KT, ST, UP, BK, NS) return launch_dma_cfg<BM, BN, KT, ST, UP, BK, NS>(a, s)

// tile shape / K-step / ring depth of one launch (cfg = 0: the measured rules below)
static void dma_pick(const DmaConvArgs& a, int cfg, int& bm, int& bn, int& bk, int& nst) {
    const bool k32 = (a.Ci % 32 == 0) && (a.C1 % 32 == 0), k64 = (a.Ci % 64 == 0) && (a.C1 % 64 == 0);
    // The tile shape fixes the order of the K reduction, so it must not depend on the batch size: an utterance's result is then
    // bit-identical for any batch split.  Grid sizes are therefore judged at the nominal per-GPU batch of
    // BASELINE.json (16 utterances), whatever a.B is.
    // (a.tile_batch > 0: the opt-in latency mode judges them at the actual batch instead, kernels.h)
    const long long kNominalBatch = a.tile_batch > 0 ? a.tile_batch : 16;
    auto blocks = [&](int bm, int bn) -> long long { return (a.Mp % bm) ? -1 : (long long)(a.Mp / bm) * ((a.To + bn - 1) / bn) * kNominalBatch; };
    if (cfg) {
        bm = cfg / 1000000; bn = (cfg / 1000) % 1000; bk = (cfg / 10) % 100; nst = cfg % 10;
    } else if (a.stride == 1 && !a.ups && a.epi != EPI_GEGLU && k32 && (blocks(64, 64) <= 256 || (blocks(64, 64) < 512 && blocks(64, 64) % 256))) {
        // fewer 64 x 64 tiles than CUs (the T/8 and T/4 levels at small batch): 32 x 64 tiles with the K range split over the
        // workgroup's two wave pairs put twice as many CUs to work (tools/bench_dconv.py: 1.6x at 128 tiles, ~1.1x at 256);
        // 1.5 tiles per CU (384: the T/2 level) leave half the chip idle in the second round, 3 half tiles per CU do not (1.2x)
        bm = 32; bn = 64; bk = (a.KT == 1 && k64) ? 64 : 32; nst = 2;
    } else {
        // Pick the tile by a small occupancy model fitted to tools/bench_dconv.py sweeps: a CU runs its n workgroups r at a
        // time (r = LDS residency, <= 4); r >= 3 co-resident workgroups keep the matrix pipe full, 2 reach ~0.9, a lone one
        // ~0.62 (nothing hides its barriers / LDS latency).  Short reductions (Ci <= 512) prefer many small stages (BK 16, 3-deep
        // ring), long ones few big stages (BK 64, 2-deep).
        struct Cand { int bm, bn, bk, nst; double eff; };
        Cand cands[3];
        int nc = 0;
        if (a.stride == 2 || a.ups) {
            cands[nc++] = {64, 64, 32, 2, 0.93};
        } else if (a.KT == 3) {
            cands[nc++] = {64, 64, 32, 2, 0.93};
            if (a.Mp % 128 == 0) cands[nc++] = {128, 128, 16, 2, 1.0};
        } else if (a.epi == EPI_GEGLU) {
            // measured (tools/bench_dconv.py ff1_*): the 128 x 128 tile wins whenever a frame block is not half empty
            if (a.To > 64) cands[nc++] = {128, 128, a.Ci >= 384 ? 16 : 32, 2, 1.0};
            else cands[nc++] = {128, 64, 32, 2, 0.97};
        } else {
            // short reductions: one resident wave of workgroups (<= 768) prefers BK 32 x 2 stages, a grid several waves deep the
            // lighter BK 16 x 3 (more workgroups per CU to overlap prologues / epilogues); long ones few big stages
            if (a.Ci > 512) cands[nc++] = {64, 64, 64, 2, 0.93};
            else if (blocks(64, 64) <= 768) cands[nc++] = {64, 64, 32, 2, 0.93};
            else cands[nc++] = {64, 64, 16, 3, 0.93};
            if (a.Mp % 128 == 0) cands[nc++] = {128, 64, 32, 2, 0.97};
        }
        double best = 1e300;
        bm = cands[0].bm; bn = cands[0].bn; bk = cands[0].bk; nst = cands[0].nst;
        for (int i = 0; i < nc; ++i) {
            const Cand& c = cands[i];
            const long long nblk = blocks(c.bm, c.bn);
            if (nblk < 0) continue;
            const int xw = a.ups ? c.bn / 2 + 2 : (c.bn - 1) * a.stride + a.KT;
            const double lds = 4.0 * c.nst * (a.KT * c.bk * c.bm + (c.bk / 4) * xw * 4);
            int r = (int)(160.0 * 1024 / lds);
            r = r < 1 ? 1 : (r > 4 ? 4 : r);
            if (c.bm * c.bn >= 128 * 128 && r > 2) r = 2;
            const long long n = (nblk + 255) / 256;
            static const double e[5] = {1.0, 0.62, 0.9, 1.0, 1.0};
            const double cost = ((double)(n / r) * r / e[r] + (double)(n % r) / e[n % r]) * c.bm * c.bn / c.eff;
            if (cost < best) { best = cost; bm = c.bm; bn = c.bn; bk = c.bk; nst = c.nst; }
        }
        // Small ACTUAL batches (the one-utterance caller): a 128 x 64 tile sums k in exactly the order of the 128 x 128 tile (one
        // accumulator chain per output, same BK), so splitting the tile when the real grid leaves most of the chip idle changes the
        // timing and nothing else -- the bit-identity between B = 1 and batched results holds.
        if (bm == 128 && bn == 128 && (long long)(a.Mp / 128) * ((a.To + 127) / 128) * a.B < 192) bn = 64;
    }
}

// Latency mode: how many workgroups share one output tile's K range (1 = no cluster split).  Doubled while the grid stays within
// one workgroup per CU (a CU with two takes twice as long), the K-steps divide evenly and every share keeps >= min_steps of them.
static int cluster_split(const DmaConvArgs& a, int bm, int bn, int nk, int nk2, int min_steps) {
    if (a.tile_batch <= 0 || !a.kpart || !a.kcount || a.voc || bm * bn > 128 * 64) return 1;
    const int blk = bm * bn > 64 * 64 ? 2 : 1;      // 32 x 32 blocks per wave
    const long long tiles = (long long)(a.Mp / bm) * ((a.To + bn - 1) / bn) * a.B;
    if (tiles * 4 > a.kcount_cap) return 1;
    int S = 1;
    while (S < 16 && tiles * S * 2 <= 256 && nk % (S * 2) == 0 && nk / (S * 2) >= min_steps && (nk2 == 0 || (nk2 % (S * 2) == 0)) &&
           tiles * 4 * (S * 2) * blk * 1024 <= a.kpart_cap)
        S *= 2;
    return S;
}

template <int BM, int BN, int BK3, int BK1, int NST>
static hipError_t launch_pair_cfg(const DmaConvArgs& a3, const DmaConvArgs& a1, hipStream_t s) {
    using Cfg = PairCfg<BM, BN, BK3, BK1, NST>;
    const int nN = (a1.To + BN - 1) / BN;
    const int S = a1.ksplit > 1 ? a1.ksplit : 1;
    if (S > 1 && (Cfg::C1::TM * Cfg::C1::TN != 1 || a3.ksplit != a1.ksplit || (a3.Ci / BK3) % S || (a1.Ci / BK1) % S)) return hipErrorInvalidValue;
    dim3 grid((a1.Mp / BM) * nN, a1.B * S);
    auto kern = conv_dma_pair_kernel<BM, BN, BK3, BK1, NST>;
    if (Cfg::LDS_BYTES > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    if (S > 1) snprintf(g_dcfg, sizeof(g_dcfg), "BM%d BN%d KT3+1 S1 U0 BK%d+%d NST%d KS%d grid %ux%u lds %zu", BM, BN, BK3, BK1, NST, S, grid.x, grid.y, Cfg::LDS_BYTES);
    else snprintf(g_dcfg, sizeof(g_dcfg), "BM%d BN%d KT3+1 S1 U0 BK%d+%d NST%d grid %ux%u lds %zu", BM, BN, BK3, BK1, NST, grid.x, grid.y, Cfg::LDS_BYTES);
    DmaPairArgs pp{a3, a1};
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, s, e0, e1, 0, pp);
    else hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, s, pp);
    return hipGetLastError();
}

// fused variant code for the pair, 0 = none: (BM << 16) | (BN << 8) | BK1
static int pair_variant(const DmaConvArgs& a3, const DmaConvArgs& a1) {
    auto plain = [](const DmaConvArgs& a) {
        return !a.voc && a.stride == 1 && !a.ups && a.dil == 1 && a.epi == EPI_NONE && a.xpad == 1 && a.opad == 1 && !a.ln_part && !a.ph_Tout && a.act_slope == 0.f &&
               !a.acc_in && a.out_div == 1.0f && a.Ci % 16 == 0 && a.C1 % 16 == 0 && a.Mp % 64 == 0 && a.B > 0 && a.To > 0;
    };
    if (!plain(a3) || !plain(a1) || a3.KT != 3 || a3.pad != 1 || a1.KT != 1 || a1.pad != 0 || a3.Mp != a1.Mp || a3.To != a1.To || a3.B != a1.B ||
        a3.Tsrc != a1.Tsrc || a1.res || a1.out_plain || a1.plain_from < a1.Cout || a1.lnpart_out)
        return 0;
    int bm, bn, bk, nst;
    dma_pick(a3, 0, bm, bn, bk, nst);      // the k 3 half carries most of the work: its tile
    const bool k3_32 = (a3.Ci % 32 == 0) && (a3.C1 % 32 == 0);
    const bool k1_64 = (a1.Ci % 64 == 0) && (a1.C1 % 64 == 0), k1_32 = (a1.Ci % 32 == 0) && (a1.C1 % 32 == 0);
    if (!k3_32 || bk != 32 || nst != 2 || !k1_32 || bn != 64 || (bm != 32 && bm != 64)) return 0;
    return (bm << 16) | (bn << 8) | (k1_64 ? 64 : 32);
}
bool conv_dma_pair_applies(const DmaConvArgs& a3, const DmaConvArgs& a1) { return pair_variant(a3, a1) != 0; }

hipError_t launch_conv_dma_pair(const DmaConvArgs& a3_, const DmaConvArgs& a1_, hipStream_t s) {
    DmaConvArgs a3 = a3_, a1 = a1_;
    const int pv = pair_variant(a3, a1);
    if (pv) {
        const int bm = pv >> 16, bn = (pv >> 8) & 255, bk1 = pv & 255;
        a3.ksplit = a1.ksplit = cluster_split(a1, bm, bn, a3.Ci / 32, a1.Ci / bk1, 1);
    }
    switch (pv) {
        case (32 << 16) | (64 << 8) | 64: return launch_pair_cfg<32, 64, 32, 64, 2>(a3, a1, s);
        case (32 << 16) | (64 << 8) | 32: return launch_pair_cfg<32, 64, 32, 32, 2>(a3, a1, s);
        case (64 << 16) | (64 << 8) | 64: return launch_pair_cfg<64, 64, 32, 64, 2>(a3, a1, s);
        case (64 << 16) | (64 << 8) | 32: return launch_pair_cfg<64, 64, 32, 32, 2>(a3, a1, s);
        default: return hipErrorNotSupported;
    }
}

// cfg = BM*1000000 + BN*1000 + BK*10 + NST (0 = auto)
hipError_t launch_conv_dma(const DmaConvArgs& a_, int cfg, hipStream_t s) {
    DmaConvArgs a = a_;
    a.ksplit = 1;
    if (a.Ci % 16 || a.C1 % 16 || a.Mp % 64 || a.B <= 0 || a.To <= 0 || a.xpad < 1 || a.opad < 1 || a.pad < 0 || a.pad > a.xpad) return hipErrorInvalidValue;
    if (a.voc) {
        // vocoder resblock convolutions (k 3 / 7 / 11, dilation 1 / 3 / 5; LeakyReLU / running-sum epilogues): one 64 x 128 tile shape,
        // BK 16 (a K-step = 16 channels x all taps: 6 / 14 / 22 MFMA groups of 8 per wave), 2 stages (the weight tile of k 11 is 45 KB per stage)
        if (a.stride != 1 || a.ups || a.epi != EPI_NONE || cfg != 0) return hipErrorInvalidValue;
        if (a.ph_Tout) {      // upsampler: 2 taps per phase
            if (a.KT != 2 || a.dil != 1 || a.pad != 1 || a.res || a.acc_in || a.out_plain || a.out_div != 1.0f || a.ph_log2 < 0 || a.ph_log2 > 4 ||
                a.ph_Cout << a.ph_log2 != a.Co || a.ph_Cout % 8)
                return hipErrorInvalidValue;
            return launch_dma_cfg<64, 128, 2, 1, false, 16, 2, 1, true>(a, s);
        }
#define VCASE(KT_, D_) if (a.KT == KT_ && a.dil == D_) return launch_dma_cfg<64, 128, KT_, 1, false, 16, 2, D_, true>(a, s)
        // one M-block (64 output channels) and 11 taps: a 256-frame tile amortises the 45 KB weight tile over twice the columns
        // (measured 104 -> 114 TFLOP/s with the residual epilogue; k 3 / k 7 lose 7 % on the wide tile)
        if (a.Mp == 64 && a.KT == 11 && a.To >= 1024 && !a.vlen) {
#define WCASE(D_) if (a.dil == D_) return launch_dma_cfg<64, 256, 11, 1, false, 16, 2, D_, true>(a, s)
            WCASE(1); WCASE(3); WCASE(5);
#undef WCASE
        }
        VCASE(3, 1); VCASE(3, 3); VCASE(3, 5); VCASE(7, 1); VCASE(7, 3); VCASE(7, 5); VCASE(11, 1); VCASE(11, 3); VCASE(11, 5);
#undef VCASE
        return hipErrorInvalidValue;
    }
    if (a.dil != 1 || a.act_slope != 0.f || a.acc_in || a.out_div != 1.0f || a.ph_Tout) return hipErrorInvalidValue;      // vocoder-only features
    if (a.KT != 1 && a.KT != 3) return hipErrorInvalidValue;
    const bool k32 = (a.Ci % 32 == 0) && (a.C1 % 32 == 0), k64 = (a.Ci % 64 == 0) && (a.C1 % 64 == 0);
    int bm, bn, bk, nst;
    dma_pick(a, cfg, bm, bn, bk, nst);
    if (a.epi == EPI_GEGLU && bm != 128) return hipErrorInvalidValue;
    if (a.Mp % bm) return hipErrorInvalidValue;
    if ((bk == 64 && !k64) || (bk == 32 && !k32)) bk = 16;
    if (a.gnf_part) {
        // GroupNorm fold: a 1x1 convolution over one source whose every wave's share of a K-step lies inside one group
        if (a.KT != 1 || a.stride != 1 || a.ups || a.C2 != 0 || a.ln_part || a.epi != EPI_NONE || a.bias || !a.gnf_cg || !a.gnf_c2 || a.gnf_groups < 1 ||
            a.gnf_groups > 8 || a.Ci % a.gnf_groups || (a.Ci / a.gnf_groups) % 16)
            return hipErrorInvalidValue;
        const int gsz = a.Ci / a.gnf_groups;
        if (bm * bn > 128 * 64) bn = 64;
        while (gsz % (bm == 32 ? bk / 2 : bk)) bk /= 2;      // (a split tile's half K-step of 16 channels divides every group size)
        if (bm != 32 && bk == 16) nst = 3;
        if (cfg == 0) a.ksplit = cluster_split(a, bm, bn, a.Ci / bk, 0, 2);
#define GCASE(BM, BN, BK, NS) if (bm == BM && bn == BN && bk == BK && nst == NS) return launch_dma_cfg<BM, BN, 1, 1, false, BK, NS, 1, false, true>(a, s)
        GCASE(32, 64, 32, 2); GCASE(32, 64, 64, 2); GCASE(64, 64, 32, 2); GCASE(64, 64, 16, 3); GCASE(128, 64, 32, 2); GCASE(128, 64, 16, 3);
#undef GCASE
        return hipErrorInvalidValue;
    }
    if (cfg == 0) a.ksplit = cluster_split(a, bm, bn, a.Ci / bk, 0, 2);
    const int key = a.KT * 100 + a.stride * 10 + (a.ups ? 1 : 0);
    const int tk = bm * 1000 + bn;
    if (key == 110) {
        if (tk == 64064) {
            if (bk == 64 && nst == 2) DCASE(64, 64, 1, 1, false, 64, 2);
            if (bk == 64 && nst == 3) DCASE(64, 64, 1, 1, false, 64, 3);
            if (bk == 32 && nst == 2) DCASE(64, 64, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) DCASE(64, 64, 1, 1, false, 32, 3);
            if (bk == 16 && nst == 2) DCASE(64, 64, 1, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(64, 64, 1, 1, false, 16, 3);
            if (bk == 64 && nst == 4) DCASE(64, 64, 1, 1, false, 64, 4);
            if (bk == 32 && nst == 4) DCASE(64, 64, 1, 1, false, 32, 4);
            if (bk == 32 && nst == 5) DCASE(64, 64, 1, 1, false, 32, 5);
            if (bk == 16 && nst == 4) DCASE(64, 64, 1, 1, false, 16, 4);
            if (bk == 16 && nst == 6) DCASE(64, 64, 1, 1, false, 16, 6);
        } else if (tk == 32064) {
            if (bk == 64) DCASE(32, 64, 1, 1, false, 64, 2);
            if (bk == 32) DCASE(32, 64, 1, 1, false, 32, 2);
        } else if (tk == 128064) {
            if (bk == 32 && nst == 2) DCASE(128, 64, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) DCASE(128, 64, 1, 1, false, 32, 3);
            if (bk == 16 && nst == 2) DCASE(128, 64, 1, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(128, 64, 1, 1, false, 16, 3);
            if (bk == 32 && nst == 4) DCASE(128, 64, 1, 1, false, 32, 4);
            if (bk == 16 && nst == 4) DCASE(128, 64, 1, 1, false, 16, 4);
        } else if (tk == 128128) {
            if (bk == 32 && nst == 2) DCASE(128, 128, 1, 1, false, 32, 2);
            if (bk == 32 && nst == 3) DCASE(128, 128, 1, 1, false, 32, 3);
            if (bk == 16 && nst == 2) DCASE(128, 128, 1, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(128, 128, 1, 1, false, 16, 3);
            if (bk == 16 && nst == 4) DCASE(128, 128, 1, 1, false, 16, 4);
        }
    } else if (key == 310) {
        if (tk == 64064) {
            if (bk == 32 && nst == 2) DCASE(64, 64, 3, 1, false, 32, 2);
            if (bk == 32 && nst == 3) DCASE(64, 64, 3, 1, false, 32, 3);
            if (bk == 16 && nst == 2) DCASE(64, 64, 3, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(64, 64, 3, 1, false, 16, 3);
            if (bk == 16 && nst == 4) DCASE(64, 64, 3, 1, false, 16, 4);
            if (bk == 32 && nst == 4) DCASE(64, 64, 3, 1, false, 32, 4);
        } else if (tk == 32064) {
            if (bk == 32) DCASE(32, 64, 3, 1, false, 32, 2);      // BK 64 and a 3-deep ring measured 0-25 % slower here (tools/bench_dconv.py)
        } else if (tk == 128064) {
            if (bk == 16 && nst == 2) DCASE(128, 64, 3, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(128, 64, 3, 1, false, 16, 3);
        } else if (tk == 128128) {
            if (bk == 16 && nst == 2) DCASE(128, 128, 3, 1, false, 16, 2);
            if (bk == 16 && nst == 3) DCASE(128, 128, 3, 1, false, 16, 3);
        }
    } else if (key == 320 && tk == 64064) {
        if (bk == 32 && nst == 2) DCASE(64, 64, 3, 2, false, 32, 2);
        if (bk >= 16 && nst == 3) DCASE(64, 64, 3, 2, false, 16, 3);
        if (bk == 16 && nst == 2) DCASE(64, 64, 3, 2, false, 16, 2);
    } else if (key == 311 && tk == 64064) {
        if (bk == 32 && nst == 2) DCASE(64, 64, 3, 1, true, 32, 2);
        if (bk >= 16 && nst == 3) DCASE(64, 64, 3, 1, true, 16, 3);
        if (bk == 16 && nst == 2) DCASE(64, 64, 3, 1, true, 16, 2);
    }
    return hipErrorInvalidValue;
}

}  // namespace lds
