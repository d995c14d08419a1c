// Self-attention for the K4P UNet path (reference Attention + AttnProcessor2_0, attention_processor.py:980-1052).
// Flash-style, fp32 MFMA (v_mfma_f32_32x32x2_f32), swapped QK^T so each lane owns one query column and the
// probabilities go from the accumulator registers straight into the P.V MFMAs.  Built like conv_dma: key/value tiles
// travel L2 -> LDS by buffer-addressed LDS-DMA (no staging registers, no vector address math), in an NST-stage ring
// with counted vmcnt waits and one s_barrier per 64-key tile; every operand is one ds_read_b128 per four MFMAs.
//   q, k arrive in K4P (k4p.h): the 4 floats at one (row, frame) are the head-dim values 8q+2j+h, j=0..3, i.e. the
//   A (keys) / B (queries) operands of four consecutive MFMAs; a 64-key tile of one row is 1 KB contiguous.
//   v arrives in the "VT" layout the QKV convolution writes for it: [B][heads][ceil(T/4)][D][4], the 4 floats being
//   V[d] at 4 consecutive keys = the A operands of the four P.V MFMAs that consume accumulator registers 4g..4g+3; a
//   64-key tile of one head is D*256 contiguous bytes and a wave's read (d = lane) is conflict-free.
// Keys >= T: their scores are masked to -inf in the last tile; their K entries are pad zeros / neighbouring rows
// (finite) and their V entries are zeros (VT tail written by the producer, or the buffer range check), so 0 * V = 0.
// Output is written in K4P (two 8-byte stores per 8-channel block, pad frames included).
#include "k4p.h"
#include "k8b3.h"
#include "kernels.h"

#include <hip/hip_ext.h>

#include <math.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int att_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));

// F16 variant (the split-fp16 GEMM mode, k8b3.h FMT_F16X2): QK^T and P.V on v_mfma_f32_32x32x16_f16 with every fp32 operand split in
// registers into two fp16 terms (a = a1 + a2, 22 significand bits) and three products per tile (a2 b1 + a1 b2 + a1 b1).  The eight k
// values of a lane's fragment are two 16-byte entries as they already lie in LDS / registers: (K4P blocks 2t, 2t+1 of row h) for the
// head-dim reduction, (key quads 4s+h, 4s+2+h) for the key reduction -- which is exactly the order in which the probabilities sit in
// the score accumulators (registers 8s .. 8s+7), so P feeds the second product from registers as before.
static __device__ __forceinline__ void att_split8(const f32x4 e0, const f32x4 e1, att_u32x4& p1, att_u32x4& p2) {
    unsigned a[4], b[4];
    k8h_split_pair(e0[0], e0[1], a[0], b[0]);
    k8h_split_pair(e0[2], e0[3], a[1], b[1]);
    k8h_split_pair(e1[0], e1[1], a[2], b[2]);
    k8h_split_pair(e1[2], e1[3], a[3], b[3]);
    p1 = att_u32x4{a[0], a[1], a[2], a[3]};
    p2 = att_u32x4{b[0], b[1], b[2], b[3]};
}
static __device__ __forceinline__ f32x16 att_mfma3(const att_u32x4 a1, const att_u32x4 a2, const att_u32x4 b1, const att_u32x4 b2, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, a2), __builtin_bit_cast(att_f16x8, b1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, a1), __builtin_bit_cast(att_f16x8, b2), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, a1), __builtin_bit_cast(att_f16x8, b1), c, 0, 0, 0);
    return c;
}

template <int D, int NW, int NST, int KS = 1>
struct AttCfg {
    static constexpr int KB = 64;                       // keys per tile
    static constexpr int DQ = D / 8, DT = (D + 31) / 32;
    static constexpr int KROWS = DQ * 2;                // K4P rows of one head = 1 KB DMA instructions per K tile
    static constexpr int VCH = D / 4;                   // 1 KB DMA instructions per V tile (16 key-quads x D x 16 B)
    static constexpr int KPW = KROWS / NW, VPW = VCH / NW;
    static constexpr int PER_TILE = KPW + VPW;          // VMEM ops per wave per tile
    static constexpr int STAGE = 2 * KB * D;            // floats: K tile + V tile
    // + slack: with D = 48 the second 32-row operand tile reads up to 15 entries past the last V tile (rows never stored)
    static constexpr size_t LDS_BYTES = (size_t)NST * STAGE * sizeof(float) + 256;
    static_assert(KROWS % NW == 0 && VCH % NW == 0, "tile must split evenly over the waves");
    static_assert(KS == 1 || (size_t)(KS - 1) * (NW / KS) * (2 + 16 * DT) * 64 * sizeof(float) <= LDS_BYTES, "the join's partials reuse the K/V stages");
};

static __device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int Y, int PER_TILE>
static __device__ __forceinline__ void att_wait_younger(int y) {
    if constexpr (Y == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        constexpr int N = (Y * PER_TILE > 63) ? 63 : Y * PER_TILE;      // vmcnt is a 6-bit field; a smaller count only waits longer
        if (y >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        else att_wait_younger<Y - 1, PER_TILE>(y);
    }
}

// tile kt -> LDS stage `st`: this wave's share of the K rows and of the V chunks (1 KB per instruction)
template <int D, int NW, int KPW, int VPW>
static __device__ __forceinline__ void att_issue_tile(const __amdgpu_buffer_rsrc_t rk, const __amdgpu_buffer_rsrc_t rv, const int* koff,
                                                      const int* voff, int wave, int kt, float* st) {
#pragma unroll
    for (int i = 0; i < KPW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(st + (wave + NW * i) * 256), 16, koff[i], kt * (64 * 16), 0, 0);
#pragma unroll
    for (int i = 0; i < VPW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(st + 64 * D + (wave + NW * i) * 256), 16, voff[i],
                                                 kt * (64 * D * 4), 0, 0);
}

// KS = 2 (short sequences): the workgroup covers (NW/2)*32 queries and its two wave groups each take one 32-key half of every
// staged tile, so twice as many waves share the work; the partial (max, sum, output) triples meet through LDS at the end.
template <int D, int NW, int NST, int KS, bool F16>
__global__ void __launch_bounds__(NW * 64) attention_k4p_kernel(const float* __restrict__ qk, const float* __restrict__ vt, float* __restrict__ out,
                                                                int C, int Tbuf, float scale2, int out_bf3, const int* __restrict__ lens, int lvl) {
    using Cfg = AttCfg<D, NW, NST, KS>;
    constexpr int NWQ = NW / KS;
    constexpr int KB = Cfg::KB, DQ = Cfg::DQ, DT = Cfg::DT, KPW = Cfg::KPW, VPW = Cfg::VPW, STAGE = Cfg::STAGE, PER_TILE = Cfg::PER_TILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // NST x { K [row][key][4] ; V [keyquad][d][4] }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    // XCD-aware order: the query blocks of one (batch, head) share K and V, so they take consecutive slots of one XCD
    // (workgroups are dispatched round-robin over the 8 XCDs, each with its own L2)
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int id = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const int per = total >> 3, rem = total & 7, xcd = id & 7;
    const int L = xcd * per + (xcd < rem ? xcd : rem) + (id >> 3);
    const int qblk = L % gx, hd = (L / gx) % gy, b = L / (gx * gy);
    const int qw = wave % NWQ, ks = wave / NWQ;       // query tile of this wave; which 32-key half of each tile it takes
    const int tq = qblk * (NWQ * 32) + qw * 32 + c;
    // T = this utterance's own length (ragged batches: keys stop there, queries beyond it are written as zeros); the strides are the buffer's
    const int T = ragged_len(lens, b, lvl, Tbuf);
    const int Tp = Tbuf + 2, T4 = (Tbuf + 3) & ~3;
    const float* qb = qk + ((long long)b * 2 * C + (long long)hd * D) * Tp;            // q rows of this head

    // queries: B operands, pre-multiplied by log2(e)/sqrt(d) so the scores come out in log2 units
    f32x4 qv[DQ];
#pragma unroll
    for (int kq = 0; kq < DQ; ++kq) {
        qv[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tq < T) qv[kq] = *reinterpret_cast<const f32x4*>(qb + ((long long)(kq * 2 + h) * Tp + tq + 1) * 4);
    }

    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qk + ((long long)b * 2 * C + C + (long long)hd * D) * Tp), 0, D * Tp * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vt + ((long long)b * C + (long long)hd * D) * T4), 0, D * T4 * 4, 0x00020000);
    // loop-invariant per-lane byte offsets of this wave's DMA shares
    int koff[KPW], voff[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) koff[i] = (((wave + NW * i) * Tp) + lane + 1) * 16;
#pragma unroll
    for (int i = 0; i < VPW; ++i) voff[i] = ((wave + NW * i) * 64 + lane) * 16;
    // (ragged batch: a query block wholly beyond the utterance's length attends to nothing; its zeros are written below)
    const int nt = (qblk * (NWQ * 32) >= T) ? 0 : (T + KB - 1) / KB;
    for (int t = 0; t < NST - 1 && t < nt; ++t) att_issue_tile<D, NW, KPW, VPW>(rk, rv, koff, voff, wave, t, smem + t * STAGE);
#pragma unroll
    for (int kq = 0; kq < DQ; ++kq) qv[kq] *= scale2;
    att_u32x4 qh1[F16 ? DQ / 2 : 1], qh2[F16 ? DQ / 2 : 1];      // F16: the query fragments, split once
    if constexpr (F16) {
#pragma unroll
        for (int t = 0; t < DQ / 2; ++t) att_split8(qv[2 * t], qv[2 * t + 1], qh1[t], qh2[t]);
    }

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // KS = 4 (latency mode) with the whole sequence resident in the ring: one wait for everything and one barrier, then every wave
    // runs through its own chunks without meeting the others (a per-tile barrier would make the four waves take turns: chunk j of
    // tile kt belongs to one wave only)
    const bool resident = (KS == 4) && nt <= NST - 1;
    if (resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    int sc = 0, sn = NST - 1;
    for (int kt = 0; kt < nt; ++kt) {
        if (!resident) {
            // this wave's share of tile kt has landed when at most the younger tiles' DMAs are outstanding
            const int younger = (nt - 1 - kt < NST - 2) ? (nt - 1 - kt) : (NST - 2);
            att_wait_younger<NST - 2, PER_TILE>(younger);
            __builtin_amdgcn_s_barrier();              // every wave's share landed; every wave is done with the stage refilled below
            asm volatile("" ::: "memory");
            if (kt + NST - 1 < nt) att_issue_tile<D, NW, KPW, VPW>(rk, rv, koff, voff, wave, kt + NST - 1, smem + sn * STAGE);
        }
        const float* Kc = smem + sc * STAGE;
        const float* Vc = Kc + KB * D;
        // KS = 1: both 32-key halves of the tile; KS = 2: half ks; KS = 4: half (ks & 1) of every other tile (the 32-key chunks of the
        // sequence go round-robin over the four waves)
        const int h0 = (KS == 1) ? 0 : (KS == 2) ? ks : (ks & 1);
        const int h1 = (KS == 1) ? KB / 32 : (KS == 2 || (kt & 1) == (ks >> 1)) ? h0 + 1 : h0;
#pragma unroll 1
        for (int half = h0; half < h1; ++half) {
            const int kbase = kt * KB + half * 32;
            if (kbase >= T) break;
            // operands are fetched ahead of their use and pinned there: all K operands before the first QK MFMA, all V operands
            // behind the QK MFMAs so their LDS latency hides under those and the softmax
            f32x4 ka[DQ], va[4][DT];
#pragma unroll
            for (int kq = 0; kq < DQ; ++kq) ka[kq] = *reinterpret_cast<const f32x4*>(Kc + ((kq * 2 + h) * KB + half * 32 + c) * 4);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            if constexpr (F16) {
#pragma unroll
                for (int t = 0; t < DQ / 2; ++t) {
                    att_u32x4 k1, k2;
                    att_split8(ka[2 * t], ka[2 * t + 1], k1, k2);
                    s = att_mfma3(k1, k2, qh1[t], qh2[t], s);
                }
            } else {
#pragma unroll
                for (int kq = 0; kq < DQ; ++kq)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kq][jj], qv[kq][jj], s, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < DT; ++i)
                    // V[d = i*32 + c][keys 8g+4h .. +3]; rows d >= D (second tile of D = 48) read a neighbour's entries and
                    // only feed output rows that are never stored
                    va[g][i] = *reinterpret_cast<const f32x4*>(Vc + ((half * 8 + 2 * g + h) * D + i * 32 + c) * 4);
            __builtin_amdgcn_sched_barrier(0);
            // all of this VALU work is paid in matrix time on gfx950 (the fp32 MFMA shares the vector ALU): one v_exp_f32
            // per score, packed subtract / add / multiply, the key mask only in the ragged last tile
            if (kbase + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 3) + 8 * (r >> 2) + 4 * h >= T) s[r] = -INFINITY;
            }
            // tile maximum per query: seven v_max3 (no NaN canonicalisation), the other 16 keys sit in lane ^ 32
            float mt = max3(max3(max3(s[0], s[1], s[2]), max3(s[3], s[4], s[5]), max3(s[6], s[7], s[8])),
                            max3(max3(s[9], s[10], s[11]), max3(s[12], s[13], s[14]), s[15]), m_run);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = max3(mt, __uint_as_float(sw[0]), __uint_as_float(sw[1]));      // one of the two is this lane's own value
            }
            // branch-free online softmax: alpha = 1 when the maximum did not move (a wave-uniform skip of the rescale costs
            // more in register copies at the join than the 8 packed multiplies it saves)
            const float alpha = __builtin_amdgcn_exp2f(m_run - mt);
            m_run = mt;
            // F16: the probabilities are formed times 2^12 (the exponent rides in the exp2 argument: no extra instruction).  Their fp16 split then
            // keeps 22 bits down to p = 2^-26 instead of 2^-14 -- over 2050 keys a typical p is 5e-4, whose second fp16 term would be subnormal --
            // and the factor cancels exactly in o / l (both sums carry it; partial sums of different waves / tiles carry the same one).
            const float msub = F16 ? mt - 12.0f : mt;
            const f32x2 mm = {msub, msub}, aa = {alpha, alpha};
            f32x2 lsum = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 d = f32x2{s[r], s[r + 1]} - mm;
                s[r] = __builtin_amdgcn_exp2f(d[0]);
                s[r + 1] = __builtin_amdgcn_exp2f(d[1]);
                lsum += f32x2{s[r], s[r + 1]};
            }
            l_run = fmaf(l_run, alpha, lsum[0] + lsum[1]);
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = f32x2{o[i][r], o[i][r + 1]} * aa;
                    o[i][r] = t[0]; o[i][r + 1] = t[1];
                }
            if constexpr (F16) {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    att_u32x4 p1, p2;
                    att_split8(f32x4{s[8 * ss], s[8 * ss + 1], s[8 * ss + 2], s[8 * ss + 3]}, f32x4{s[8 * ss + 4], s[8 * ss + 5], s[8 * ss + 6], s[8 * ss + 7]}, p1, p2);
#pragma unroll
                    for (int i = 0; i < DT; ++i) {
                        att_u32x4 v1, v2;
                        att_split8(va[2 * ss][i], va[2 * ss + 1][i], v1, v2);
                        o[i] = att_mfma3(v1, v2, p1, p2, o[i]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < DT; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[g][i][e], s[4 * g + e], o[i], 0, 0, 0);
            }
        }
        sc = (sc + 1 == NST) ? 0 : sc + 1;
        sn = (sn + 1 == NST) ? 0 : sn + 1;
    }
    if constexpr (KS >= 2) {
        // join the key shares: (m, l, o) of the ks >= 1 waves into their ks = 0 partner, fixed order
        __syncthreads();                                   // all waves are done with the K/V stages
        constexpr int RED = (2 + 16 * DT) * 64;            // floats per partial
        if (ks >= 1) {
            float* red = smem + ((ks - 1) * NWQ + qw) * RED + lane;
            red[0] = m_run; red[64] = l_run;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(2 + i * 16 + r) * 64] = o[i][r];
        }
        __syncthreads();
        if (ks >= 1) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const float* red = smem + ((q - 1) * NWQ + qw) * RED + lane;
            const float m1 = red[0], l1 = red[64];
            const float m = fmaxf(m_run, m1);
            const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);      // a1 = 0 when the other share saw no key
            l_run = l_run * a0 + l1 * a1;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + red[(2 + i * 16 + r) * 64] * a1;
            m_run = m;
        }
    }
    float l;
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // own + partner (each appears once)
    }
    if (tq < Tbuf && out_bf3) {
        const bool live = tq < T;      // (ragged batch: queries beyond the utterance's length are written as zero planes)
        // split-plane output (k8b3.h; the split-GEMM path's to_out projection reads it): rows 8g + 4h + e of a 32-row tile are channel
        // positions 4h + e of 8-channel block g -- this lane half's 8 bytes of each plane's 16-byte entry.  out_bf3: 1 = bf16x3, 2 = fp16x2
        const float rl = 1.0f / l;
        char* ob = reinterpret_cast<char*>(out);
        const int npl = (out_bf3 == 2) ? 2 : 3;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (i * 32 + 8 * g >= D) break;
                const int q = (hd * D + i * 32) / 8 + g;
                unsigned pa[3], pb[3];
                if (out_bf3 == 2) {
                    sp_split_pair<FMT_F16X2>(o[i][4 * g] * rl, o[i][4 * g + 1] * rl, pa);
                    sp_split_pair<FMT_F16X2>(o[i][4 * g + 2] * rl, o[i][4 * g + 3] * rl, pb);
                } else {
                    sp_split_pair<FMT_BF16X3>(o[i][4 * g] * rl, o[i][4 * g + 1] * rl, pa);
                    sp_split_pair<FMT_BF16X3>(o[i][4 * g + 2] * rl, o[i][4 * g + 3] * rl, pb);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (pl >= npl) break;
                    char* e = ob + ((((long long)b * (C >> 3) + q) * npl + pl) * Tp + tq + 1) * 16 + h * 8;
                    k8_store_wt(e, live ? k8_u32x2{pa[pl], pb[pl]} : k8_u32x2{0u, 0u});
                    if (tq == 0) *reinterpret_cast<k8_u32x2*>(e - 16) = k8_u32x2{0u, 0u};
                    if (tq == Tbuf - 1) *reinterpret_cast<k8_u32x2*>(e + 16) = k8_u32x2{0u, 0u};
                }
            }
    } else if (tq < Tbuf) {
        const bool live = tq < T;      // (ragged batch: queries beyond the utterance's length are written as zeros)
        const float rl = 1.0f / l;
        float* ob = out + (long long)b * C * Tp;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (i * 32 + 8 * g >= D) break;
                const int q = (hd * D + i * 32) / 8 + g;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const long long off = ((long long)(q * 2 + hh) * Tp + tq + 1) * 4 + 2 * h;
                    k4p_store_wt(ob + off, live ? f32x2{o[i][4 * g + hh] * rl, o[i][4 * g + 2 + hh] * rl} : f32x2{0.f, 0.f});      // write-through (k4p.h)
                    if (tq == 0) *reinterpret_cast<f32x2*>(ob + off - 4) = f32x2{0.f, 0.f};
                    if (tq == Tbuf - 1) *reinterpret_cast<f32x2*>(ob + off + 4) = f32x2{0.f, 0.f};
                }
            }
    }
}

template <int D, int NW, int NST, int KS, bool F16>
static hipError_t launch_cfg(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, float scale, int out_bf3, const int* lens, int lvl, hipStream_t s) {
    using Cfg = AttCfg<D, NW, NST, KS>;
    auto kern = attention_k4p_kernel<D, NW, NST, KS, F16>;
    if (Cfg::LDS_BYTES > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    constexpr int QPB = NW / KS * 32;       // queries per workgroup
    hipEvent_t e0, e1;
    if (prof_attach_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, dim3((T + QPB - 1) / QPB, heads, B), dim3(NW * 64), Cfg::LDS_BYTES, s, e0, e1, 0, qk, vt, out, C, T, scale, out_bf3, lens, lvl);
    else hipLaunchKernelGGL(kern, dim3((T + QPB - 1) / QPB, heads, B), dim3(NW * 64), Cfg::LDS_BYTES, s, qk, vt, out, C, T, scale, out_bf3, lens, lvl);
    return hipGetLastError();
}

template <int D, bool F16>
static hipError_t launch_dk(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, int out_bf3, int tile_batch, const int* lens, int lvl, hipStream_t s) {
    const float scale = 1.4426950408889634f / sqrtf((float)D);    // log2(e) / sqrt(d)
    constexpr int NST = (D == 64) ? 2 : 3;
    // judged at the nominal per-GPU batch (16): the choice fixes the summation order, which must not depend on the batch size
    // (tile_batch > 0: the latency mode judges at the actual batch, kernels.h)
    const long long hb = (long long)heads * (tile_batch > 0 ? tile_batch : 16);
    // 128 queries per workgroup (four waves share each K/V tile) when that gives every CU two workgroups; for shorter
    // sequences 64 queries with the keys of each tile split over two wave groups; 32-query single-tile case last
    if ((long long)((T + 127) / 128) * hb >= 512) return launch_cfg<D, 4, NST, 1, F16>(qk, vt, out, B, C, T, heads, scale, out_bf3, lens, lvl, s);
    // latency mode, few (batch, head) pairs: 32 queries per workgroup, the keys' 32-key chunks round-robin over its four waves
    // (and a ring deep enough to have every K / V tile of the level's sequence in flight at once: with one workgroup per CU the tiles'
    //  DMA round trips, ~1 us each when taken two at a time, are what a short launch consists of; 144 / 123 / 128 KB of LDS)
    constexpr int NSTL = (D == 32) ? 9 : (D == 48) ? 5 : 4;
    if (tile_batch > 0 && T > 64 && (long long)((T + 63) / 64) * hb < 256) return launch_cfg<D, 4, NSTL, 4, F16>(qk, vt, out, B, C, T, heads, scale, out_bf3, lens, lvl, s);
    if (T > 32) return launch_cfg<D, 4, NST, 2, F16>(qk, vt, out, B, C, T, heads, scale, out_bf3, lens, lvl, s);
    return launch_cfg<D, 1, 2, 1, F16>(qk, vt, out, B, C, T, heads, scale, out_bf3, lens, lvl, s);
}

// math_f16: the two products on the fp16 matrix pipe with operands split in registers (the split-fp16 GEMM mode); else exact fp32
static hipError_t attention_any(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, int out_bf3, bool math_f16, int tile_batch, hipStream_t s,
                                const int* lens = nullptr, int lvl = 0) {
    if (C % heads) return hipErrorInvalidValue;
    ProfScope ps(s, math_f16 ? "attention_f16" : "attention", 4.0 * B * (double)T * T * C, 4.0 * 4.0 * B * C * T, true);
    if (math_f16) {
        switch (C / heads) {
            case 32: return launch_dk<32, true>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
            case 48: return launch_dk<48, true>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
            case 64: return launch_dk<64, true>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (C / heads) {
        case 32: return launch_dk<32, false>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
        case 48: return launch_dk<48, false>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
        case 64: return launch_dk<64, false>(qk, vt, out, B, C, T, heads, out_bf3, tile_batch, lens, lvl, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_attention_k4p(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, hipStream_t s, int tile_batch, const int* lens, int lvl) {
    return attention_any(qk, vt, out, B, C, T, heads, 0, false, tile_batch, s, lens, lvl);
}
hipError_t launch_attention_k4p_out_bf3(const float* qk, const float* vt, void* out, int B, int C, int T, int heads, hipStream_t s, int fmt, int tile_batch, const int* lens, int lvl) {
    return attention_any(qk, vt, (float*)out, B, C, T, heads, fmt == FMT_F16X2 ? 2 : 1, fmt == FMT_F16X2, tile_batch, s, lens, lvl);
}
// test entry: K4P fp32 in and out, the products on the fp16 pipe
hipError_t launch_attention_k4p_f16math(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, hipStream_t s, int tile_batch) {
    return attention_any(qk, vt, out, B, C, T, heads, 0, true, tile_batch, s);
}

}  // namespace lds
