// Generic implicit-GEMM 1-D convolution for gfx950 on the exact-fp32 matrix core (v_mfma_f32_32x32x2_f32), used where the
// UNet's K4P / LDS-DMA kernel (conv_dma.hip) does not apply: the HiFi-VAEGAN generator (reference
// encoder/hifi_vaegan/modules/models.py:161-262: k3/k7/k11 dilated convolutions with LeakyReLU on the input, ConvTranspose1d
// as polyphase filters, residual / running-sum epilogues, tanh) and the unit-embedding Linear (unit2mel.py:79-82).
//
// Data layout: activations plain [B][C][T] (frames contiguous), weights packed once to [tap][Ci/8][2][Mp][4] (k-interleaved,
// output channel contiguous).  The activation window (BN frames + dilated halo of BK channels) is staged in LDS once per
// K-step and re-read at shifted offsets by every tap; LeakyReLU is applied while staging.
//
// Pipeline: producer / consumer wave specialisation.  A workgroup is 8 waves: waves 4-7 (one per SIMD) fetch the next
// activation tile with range-checked buffer loads, apply the activation, transpose in registers and write the
// double-buffered LDS stage, and issue the weight tile's LDS-DMA; waves 0-3 only read LDS operands (register ring, NB-1
// groups ahead) and issue MFMAs.  One LDS-only barrier per K-step; the global prefetch stays in flight across it.  (On
// gfx950 the producers' vector work still competes with the fp32 MFMA for issue slots -- DESIGN.md 3 -- which is why the
// UNet moved to conv_dma; here the long reductions (K = Ci * 7..11) keep the matrix pipe at ~100 TFLOP/s.)
#include "kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// SiLU on the staging path: hardware exp2 / rcp (about 1 ulp each); the extra ~2e-7 relative error is far inside
// the stated 2e-5 UNet tolerance and keeps the transform at ~6 VALU ops per element

static __device__ __forceinline__ int floor4(int s) { return (s >= 0) ? (s & ~3) : -(((-s) + 3) & ~3); }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX, int BK>
struct ConvCfg {
    static constexpr int WAVES_M = (BM >= 64) ? 2 : 1;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int TM = BM / (32 * WAVES_M);
    static constexpr int TN = BN / (32 * WAVES_N);
    static constexpr int KR = BK / 4;       // staged "k-rows": (kq, h) pairs, each holding 4 k values per column
    static constexpr int TPR = 256 / KR;    // threads that stage one k-row of activations
    static constexpr int XWMAX = UPS ? (BN / 2 + 2 + 3) : ((BN - 1) * STRIDE + (KT - 1) * DILMAX + 1 + 3);
    static constexpr int XW4MAX = (XWMAX + 3) / 4;
    static constexpr int XCH = (XW4MAX + TPR - 1) / TPR;   // 4-frame chunks per thread (x 4 channel rows each)
    static constexpr int WCHUNKS = KT * KR * BM;           // float4 chunks of the weight tile
    static constexpr int WCH = (WCHUNKS + 255) / 256;
    static constexpr int G = KT * BK / 8;                  // MFMA groups per K-step: one (tap, kq) = 4 k-pairs
    static constexpr bool INTERLEAVE = (KT <= 3);          // fully unrolled K-step with commit pieces between MFMA groups
    static constexpr int NB = (TM * TN >= 4) ? 2 : 3;      // operand ring depth (groups)
    // independent accumulator chains per output tile: a single dependent chain of v_mfma_f32_32x32x2_f32 issued only
    // every ~106 cycles in the consumer-only ablation (64 is the pipe rate), two interleaved chains hide the
    // accumulator read-after-write gap; the chains are summed once in the epilogue
    static constexpr int NACC = (TM * TN >= 2) ? 1 : 2;
    static constexpr int NST = 2;                          // LDS ring stages (3 measured slower here: one workgroup per CU instead of two)
    static constexpr size_t lds_bytes(int xw4) { return (size_t)(NST * (KT * BK * BM + BK * xw4 * 4) + 2 * xw4 * 4) * sizeof(float); }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// staging transform, resolved once per K-step so the per-element code carries no mode switches
enum { M_PLAIN = 0, M_LRELU = 1 };

// All per-thread state lives in one struct whose methods are force-inlined: every register array is a
// member indexed by compile-time constants (template recursion), so nothing falls back to scratch.
//
// k-interleaved tiles: v_mfma_f32_32x32x2_f32 takes ONE scalar of A and B per lane (lane half h supplies
// k = 2*kp + h), so a naive [k][n] LDS image costs one ds_read_b32 per operand per MFMA and the kernel becomes
// LDS-latency / issue bound.  Both tiles are therefore stored as [kq][h][column][4] with
// k = 8*kq + 2*j + h (j = float4 element): one ds_read_b128 per lane delivers the operands of FOUR
// consecutive MFMAs.  The weights are packed that way once on the host; the activations are transposed for
// free in registers (each staging thread loads the same 4 frames of 4 channel rows and writes 4 float4s).
template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX, int BK>
struct ConvKernel {
    using Cfg = ConvCfg<BM, BN, KT, STRIDE, UPS, DILMAX, BK>;
    static constexpr int TM = Cfg::TM, TN = Cfg::TN, XCH = Cfg::XCH, WCH = Cfg::WCH, TPR = Cfg::TPR, G = Cfg::G, KR = Cfg::KR;
    static constexpr int NB = Cfg::NB, NACC = Cfg::NACC;
    static constexpr int P = XCH * 4;            // activation commit pieces per K-step (one ds_write_b128 each)

    const ConvArgs& p;
    float* smem;
    bool producer;
    int tid, c, h, wm, wn, b, m0, t0;
    int in_valid, out_valid;      // ragged batch: this batch element's valid input / output frames
    int s_al, off, xw4, xwp, stage;
    bool vec_ok;
    int xrr, xc0, arow;
    int bcol[TN];
    f32x4 xr[XCH][4];                  // [chunk][channel row j] = 4 frames
    __amdgpu_buffer_rsrc_t rs1, rs2;   // per-batch slabs of the two sources: out-of-slab reads return 0 (hardware range check)
    __amdgpu_buffer_rsrc_t rw;         // packed weights
    int woffv[WCH];                    // per-lane byte offsets of this thread's weight chunks inside one K-step slab
    f32x16 acc[NACC][TM][TN];
    f32x4 aop[NB][TM], bop[NB][TN];

    __device__ __forceinline__ ConvKernel(const ConvArgs& p_, float* smem_) : p(p_), smem(smem_) {}

    __device__ __forceinline__ void setup() {
        // 8 waves: 0-3 consume (ds_read + MFMA only), 4-7 produce (global loads, normalise/activate, LDS writes).
        // The role is made provably wave-uniform so the branch is scalar and each wave executes exactly one
        // role's barriers.
        producer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
        tid = threadIdx.x & 255;
        const int lane = tid & 63, wave = tid >> 6;
        c = lane & 31; h = lane >> 5;
        wm = wave / Cfg::WAVES_N; wn = wave % Cfg::WAVES_N;
        const int nMb = p.Mp / BM;
        const int mb = blockIdx.x % nMb;          // M fastest: blocks that share an activation tile are neighbours
        const int nb = blockIdx.x / nMb;
        b = blockIdx.y;
        in_valid = p.vlen_in ? (p.vlen_in[b] < p.Tsrc ? p.vlen_in[b] : p.Tsrc) : p.Tsrc;      // ragged batch (kernels.h ConvArgs::vlen_in / vlen)
        out_valid = p.vlen ? p.vlen[b] : 0x7fffffff;
        m0 = mb * BM; t0 = nb * BN;
        int width;
        if (UPS) {
            s_al = floor4((t0 - 1) >> 1);
            off = 0;
            width = ((t0 + BN) >> 1) - s_al + 1;
        } else {
            const int s0 = t0 * STRIDE - p.pad;
            s_al = floor4(s0);
            off = s0 - s_al;
            width = (BN - 1) * STRIDE + (KT - 1) * p.dil + 1 + off;
        }
        xw4 = (width + 3) >> 2;
        xwp = xw4 * 4;
        stage = KT * BK * BM + BK * xwp;   // floats per LDS stage
        vec_ok = ((p.Tsrc & 3) == 0);
        xrr = tid / TPR;                   // staged k-row (kq*2 + h') of this thread, chunks xc0, xc0+TPR, ...
        xc0 = tid - xrr * TPR;
        arow = wm * TM * 32 + c;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * TN * 32 + j * 32 + c;
            bcol[j] = UPS ? nl : (off + nl * STRIDE);
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < XCH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long long)b * p.xb1), 0, p.C1 * p.Tsrc * 4, 0x00020000);
        rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 + (long long)b * p.xb2), 0, p.C2 * p.Tsrc * 4, 0x00020000);
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, KT * p.Ci * p.Mp * 4, 0x00020000);
#pragma unroll
        for (int J = 0; J < WCH; ++J) {
            const int q = tid + J * 256;
            const int tap = q / (KR * BM);
            const int rem = q - tap * (KR * BM);
            const int rr = rem / BM, m = rem - rr * BM;
            woffv[J] = ((tap * (p.Ci / 4) + rr) * p.Mp + m0 + m) * 16;
        }
    }

    // channel of row j (0..3) staged by this thread in K-step kc: k = 8*kq + 2*j + h'
    __device__ __forceinline__ int chan(int kc, int j) const { return kc * BK + 8 * (xrr >> 1) + 2 * j + (xrr & 1); }
    // 4 frames of channel row J for chunk I: one buffer_load_dwordx4 whose range check supplies the zeros outside the
    // tensor; frames outside [0, Tsrc) that land in a neighbouring row are zeroed by the select in commit_piece
    template <int I, int J>
    __device__ __forceinline__ void fetch_x(int kc, bool src2) {
        if constexpr (I < XCH) {
            const int c4 = xc0 + I * TPR;
            const int ci = chan(kc, J) - (src2 ? p.C1 : 0);
            const int voff = (ci * p.Tsrc + s_al + c4 * 4) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c4 < xw4) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 raw = src2 ? __builtin_amdgcn_raw_buffer_load_b128(rs2, voff, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, 0, 0);
                v = __builtin_bit_cast(f32x4, raw);
            }
            xr[I][J] = v;
            if constexpr (J < 3) fetch_x<I, J + 1>(kc, src2);
            else fetch_x<I + 1, 0>(kc, src2);
        }
    }
    // weight tile kc -> LDS stage by LDS-DMA (global_load_lds_dwordx4): a pure linear copy, no registers, no ds_write.
    // Each wave-instruction moves 64 consecutive 16-byte chunks; the LDS base must be wave-uniform.
    template <int J>
    __device__ __forceinline__ void dma_w(int kc, float* st) {
        if constexpr (J < WCH) {
            const int q0 = __builtin_amdgcn_readfirstlane(tid & ~63) + J * 256;     // first chunk of this wave
            if (q0 < Cfg::WCHUNKS)      // buffer-addressed: loop-invariant per-lane offset + scalar K-step offset, no vector address math
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + 4 * q0), 16, woffv[J], kc * (KR * 16) * p.Mp, 0, 0);
            dma_w<J + 1>(kc, st);
        }
    }
    __device__ __forceinline__ void fetch(int kc) {
        fetch_x<0, 0>(kc, kc * BK >= p.C1);      // C1 % BK == 0 (checked at launch): a K-step never straddles the two sources
    }

    // piece PC of the staged tile: PC < 4*XCH -> one frame of an activation chunk (transform of 4 channel rows +
    // one ds_write_b128), else one weight chunk.  Out-of-range frames are selected to exact zero (conv padding
    // applies AFTER normalisation/activation in the reference), never branched around.
    template <int PC, int MODE>
    __device__ __forceinline__ void commit_piece(float* st) {
        if constexpr (PC < XCH * 4) {
            constexpr int I = PC / 4, E = PC % 4;
            const int c4 = xc0 + I * TPR;
            if (c4 < xw4) {
                const int s = s_al + c4 * 4 + E;
                const bool inb = (s >= 0 && s < in_valid);
                f32x4 v;
                if constexpr (MODE == M_PLAIN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = xr[I][j][E];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float x = xr[I][j][E]; v[j] = (x >= 0.f) ? x : x * p.slope; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = inb ? v[j] : 0.f;
                *reinterpret_cast<f32x4*>(st + KT * BK * BM + (xrr * xwp + c4 * 4 + E) * 4) = v;
            }
        }
    }
    template <int LO, int HI, int MODE>
    __device__ __forceinline__ void commit_range(float* st) {
        if constexpr (LO < HI) {
            commit_piece<LO, MODE>(st);
            commit_range<LO + 1, HI, MODE>(st);
        }
    }
    __device__ __forceinline__ void commit_tile(float* st, int mode) {
        if (mode == M_PLAIN) commit_range<0, P, M_PLAIN>(st);
        else commit_range<0, P, M_LRELU>(st);
    }
    __device__ __forceinline__ int staging_mode() const { return p.act_in == ACT_LRELU ? M_LRELU : M_PLAIN; }

    // MFMA operands of group (tap, kq) are read from LDS NB-1 groups ahead of their use into a small register
    // ring, so the LDS latency is covered by the MFMAs in between
    template <int SLOT>
    __device__ __forceinline__ void load_ops(const float* st, int tap, int kq) {
        const float* wt = st + ((tap * KR + kq * 2 + h) * BM + arow) * 4;
        const float* xs = st + KT * BK * BM + (kq * 2 + h) * xwp * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) aop[SLOT][i] = *reinterpret_cast<const f32x4*>(wt + i * 32 * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = UPS ? (((t0 + bcol[j] + tap - 1) >> 1) - s_al) : (bcol[j] + tap * p.dil);
            bop[SLOT][j] = *reinterpret_cast<const f32x4*>(xs + col * 4);
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void mfma_ops() {
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
            load_ops<g0 % NB>(cur, g0 / (BK / 8), g0 % (BK / 8));
            preload<g0 + 1, g1>(cur);
        }
    }

    // consumer: all MFMA groups of one K-step, operands prefetched NB-1 groups ahead
    template <int g>
    __device__ __forceinline__ void kstep(const float* cur) {
        if constexpr (g < G) {
            if constexpr (g == 0) preload<0, NB - 1>(cur);
            if constexpr (g + NB - 1 < G) load_ops<(g + NB - 1) % NB>(cur, (g + NB - 1) / (BK / 8), (g + NB - 1) % (BK / 8));
            mfma_ops<g % NB>();
            // pin the software pipeline: otherwise the machine scheduler sinks every operand read back next to its
            // MFMA (ds_read; s_waitcnt lgkmcnt(0); v_mfma) and the LDS latency is exposed once per group
            __builtin_amdgcn_sched_barrier(0);
            kstep<g + 1>(cur);
        }
    }

    // the BK/8 groups of one (runtime) tap (vocoder taps 7 / 11)
    template <int kq>
    __device__ __forceinline__ void tap_groups(const float* st, int tap) {
        if constexpr (kq < BK / 8) {
            if constexpr (kq == 0) load_ops<0>(st, tap, 0);
            if constexpr (kq + 1 < BK / 8) load_ops<(kq + 1) % NB>(st, tap, kq + 1);
            mfma_ops<kq % NB>();
            __builtin_amdgcn_sched_barrier(0);
            tap_groups<kq + 1>(st, tap);
        }
    }

    // workgroup barrier that only drains LDS traffic: __syncthreads() would also wait for vmcnt(0), i.e. for
    // the producers' global prefetch of the tile after next
    static __device__ __forceinline__ void lds_barrier() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    __device__ __forceinline__ void mainloop() {
        constexpr int NST = Cfg::NST;
        constexpr int AHEAD = NST - 1;                // tiles staged ahead of the one being consumed
        constexpr int INFLIGHT = WCH + XCH * 4;   // VMEM ops of one producer iteration: weight DMA batch + activation loads
        const int nk = p.Ci / BK;
        if (producer) {
            const int mode = staging_mode();
            // prologue: tiles 0..AHEAD-1 staged, tile AHEAD's activations in registers
            fetch(0);
            for (int t = 0; t < AHEAD && t < nk; ++t) {
                commit_tile(smem + t * stage, mode);
                __builtin_amdgcn_sched_barrier(0);
                dma_w<0>(t, smem + t * stage);
                if (t + 1 < nk) fetch(t + 1);
            }
            if (nk > AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XCH * 4) : "memory");   // every DMA landed, newest loads in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            int sn = (AHEAD == NST) ? 0 : AHEAD;            // stage of tile kc+AHEAD
            for (int kc = 0; kc < nk; ++kc) {
                float* nxt = smem + sn * stage;
                if (kc + AHEAD < nk) {
                    // activations of tile kc+AHEAD (loaded one K-step ago): transform + LDS writes.  Done BEFORE issuing the
                    // DMA: hipcc drains vmcnt(0) at the first use of a register load while an LDS-DMA is in flight.
                    commit_tile(nxt, mode);
                    __builtin_amdgcn_sched_barrier(0);
                    dma_w<0>(kc + AHEAD, nxt);               // weights of tile kc+AHEAD
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc + AHEAD + 1 < nk) {
                        fetch(kc + AHEAD + 1);
                        if constexpr (AHEAD >= 2) {
                            // the weight DMA issued one K-step ago must have landed; this K-step's DMA + loads stay in flight
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XCH * 4) : "memory");   // this K-step's DMA landed
                        }
                    } else {
                        if constexpr (AHEAD >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCH) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                lds_barrier();
                sn = (sn + 1 == NST) ? 0 : sn + 1;
            }
        } else {
            lds_barrier();
            int sc = 0;
            for (int kc = 0; kc < nk; ++kc) {
                const float* cur = smem + sc * stage;
                if constexpr (Cfg::INTERLEAVE) {
                    kstep<0>(cur);
                } else {
#pragma unroll 1
                    for (int tap = 0; tap < KT; ++tap) tap_groups<0>(cur, tap);
                }
                lds_barrier();
                sc = (sc + 1 == NST) ? 0 : sc + 1;
            }
        }
    }

    // PH: polyphase (ConvTranspose) output mapping.  A template flag, like every other wave-uniform condition below it is
    // kept OUT of the per-element loops: a branch inside them splits the loop body into basic blocks and the loads of
    // different elements can then no longer be issued together.
    template <bool PH>
    __device__ __forceinline__ bool elem(int i, int r, int n, int& co, long long& oi, bool* live = nullptr) const {
        const int orow = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        co = orow;
        int to = n;
        if constexpr (PH) { co = orow / p.phases; to = n * p.phases + (orow - co * p.phases) - p.tpad; }
        oi = ((long long)b * p.Cout + co) * p.Tout + to;
        if (live) *live = to < out_valid;      // (ragged batch: output frames beyond the utterance's length are stored as zeros)
        return co < p.Cout && n < p.To && to >= 0 && to < p.Tout;
    }

    // acc[0] += src[index] for every valid element of every tile: unconditional loads from a clamped index + select
    template <bool PH, bool PER_CO>
    __device__ __forceinline__ void add_from(const float* src) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = t0 + wn * TN * 32 + j * 32 + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co; long long oi;
                    const bool ok = elem<PH>(i, r, n, co, oi);
                    const long long idx = PER_CO ? (long long)b * p.Cout + co : oi;
                    const float v = src[ok ? idx : 0];
                    acc[0][i][j][r] += ok ? v : 0.f;
                }
            }
    }

    template <bool PH>
    __device__ __forceinline__ void epilogue_tail() {
        if (p.bias_bc) add_from<PH, true>(p.bias_bc);
        if (p.res) add_from<PH, false>(p.res);
        if (p.epi == EPI_TANH) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] = tanhf(acc[0][i][j][r]);
        }
        if (p.accum) add_from<PH, false>(p.out);
        if (p.out_div != 1.0f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] = acc[0][i][j][r] / p.out_div;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = t0 + wn * TN * 32 + j * 32 + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co; long long oi; bool live;
                    if (elem<PH>(i, r, n, co, oi, &live)) p.out[oi] = live ? acc[0][i][j][r] : 0.f;
                }
            }
    }

    // Epilogue in phases: every load of a phase (row bias; then per-element residual / broadcast bias / accumulate operands of
    // ALL tiles) is issued before the first store.  A load placed after a store cannot be moved above it (possible aliasing
    // -- `accum` even reads the output buffer), and a load -> wait -> store chain per element costs one memory round trip.
    __device__ __forceinline__ void epilogue() {
        if constexpr (NACC == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][i][j] += acc[NACC - 1][i][j];
        }
        if (p.bias) {
            float kb[TM][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) kb[i][r] = p.bias[m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] += kb[i][r];
        }
        if (p.phases > 1) epilogue_tail<true>();
        else epilogue_tail<false>();
    }
};

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX, int BK>
__global__ void __launch_bounds__(512) conv_gemm_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ConvKernel<BM, BN, KT, STRIDE, UPS, DILMAX, BK> k(p, smem);
    k.setup();
    k.mainloop();
    if (!k.producer) k.epilogue();
}

static thread_local char g_cfg[96] = "";
const char* conv_gemm_last_config() { return g_cfg; }

template <int BM, int BN, int KT, int STRIDE, bool UPS, int DILMAX, int BK>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<BM, BN, KT, STRIDE, UPS, DILMAX, BK>;
    int width;
    if (UPS) width = BN / 2 + 2 + 3;
    else width = (BN - 1) * STRIDE + (KT - 1) * a.dil + 1 + 3;
    const int xw4 = (width + 3) / 4;
    if (xw4 > Cfg::XW4MAX) return hipErrorInvalidValue;
    const size_t lds = Cfg::lds_bytes(xw4);
    const int nN = (a.To + BN - 1) / BN;
    dim3 grid((a.Mp / BM) * nN, a.B);
    auto kern = conv_gemm_kernel<BM, BN, KT, STRIDE, UPS, DILMAX, BK>;
    if (lds > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
        if (e != hipSuccess) return e;
    }
    snprintf(g_cfg, sizeof(g_cfg), "BM%d BN%d KT%d S%d U%d BK%d grid %ux%u lds %zu", BM, BN, KT, STRIDE, (int)UPS, BK, grid.x, grid.y, lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
    return hipGetLastError();
}

static int auto_tile(const ConvArgs& a) {
    // Prefer the 128x128 tile (2x2 MFMA tiles per wave, highest operand reuse) when it still yields
    // >= 2 workgroups per CU; otherwise fall back to smaller tiles to fill 256 CUs.
    // judged at the nominal per-GPU batch (16 utterances): the tile fixes the reduction order, which must not depend on the
    // batch size (per-utterance results bit-identical for any batch split)
    auto blocks = [&](int bm, int bn) -> long long {
        if (a.Mp % bm) return -1;
        return (long long)(a.Mp / bm) * ((a.To + bn - 1) / bn) * 16;
    };
    if (a.Mp % 64 != 0) return 32128;
    if ((a.KT == 1 || a.KT == 3) && a.dil == 1) {
        if (blocks(128, 128) >= 512) return 128128;
        if (a.Mp % 128 == 0 && blocks(128, 64) >= 512) return 128064;
        return 64064;
    }
    return (blocks(64, 128) >= 512) ? 64128 : 64064;   // vocoder taps 2/3(dilated)/7/11
}

#define LDS_CASE(BM, BN, KT, ST, UP, DM, BK) return launch_cfg<BM, BN, KT, ST, UP, DM, BK>(a, s)

hipError_t launch_conv_gemm(const ConvArgs& a, int tile, hipStream_t s) {
    if (a.Ci % 16 != 0 || a.C1 % 16 != 0 || a.Mp % 32 != 0 || a.B <= 0 || a.To <= 0) return hipErrorInvalidValue;
    if (a.stride != 1 || a.ups || (a.epi != EPI_NONE && a.epi != EPI_TANH)) return hipErrorInvalidValue;      // conv_dma covers those for the UNet
    if (tile == 0) tile = auto_tile(a);
    const int bm = tile / 1000;
    if (a.Mp % bm != 0) return hipErrorInvalidValue;
    const int key = a.KT * 100 + a.stride * 10 + (a.ups ? 1 : 0);
    const bool wide = a.dil > 1;
    if (a.dil > 5) return hipErrorInvalidValue;
    const bool k32 = (a.Ci % 32 == 0) && (a.C1 % 32 == 0);
    const bool k64 = (a.Ci % 64 == 0) && (a.C1 % 64 == 0);
    switch (tile) {
        case 128128:
            if (key == 110 && k32) LDS_CASE(128, 128, 1, 1, false, 1, 32);
            if (key == 110) LDS_CASE(128, 128, 1, 1, false, 1, 16);
            if (key == 310 && !wide) LDS_CASE(128, 128, 3, 1, false, 1, 16);
            break;
        case 128064:
            if (key == 110 && k32) LDS_CASE(128, 64, 1, 1, false, 1, 32);
            if (key == 110) LDS_CASE(128, 64, 1, 1, false, 1, 16);
            if (key == 310 && !wide) LDS_CASE(128, 64, 3, 1, false, 1, 16);
            break;
        case 64064:
            if (key == 110 && k64) LDS_CASE(64, 64, 1, 1, false, 1, 64);
            if (key == 110 && k32) LDS_CASE(64, 64, 1, 1, false, 1, 32);
            if (key == 110) LDS_CASE(64, 64, 1, 1, false, 1, 16);
            if (key == 210) LDS_CASE(64, 64, 2, 1, false, 1, 16);
            if (key == 310 && !wide && k32) LDS_CASE(64, 64, 3, 1, false, 1, 32);
            if (key == 310 && !wide) LDS_CASE(64, 64, 3, 1, false, 1, 16);
            if (key == 310 && wide) LDS_CASE(64, 64, 3, 1, false, 5, 16);
            if (key == 710) LDS_CASE(64, 64, 7, 1, false, 5, 16);
            if (key == 1110) LDS_CASE(64, 64, 11, 1, false, 5, 16);
            break;
        case 64128:
            if (key == 110) LDS_CASE(64, 128, 1, 1, false, 1, 16);
            if (key == 210) LDS_CASE(64, 128, 2, 1, false, 1, 16);
            if (key == 310) LDS_CASE(64, 128, 3, 1, false, 5, 16);
            if (key == 710) LDS_CASE(64, 128, 7, 1, false, 5, 16);
            if (key == 1110) LDS_CASE(64, 128, 11, 1, false, 5, 16);
            break;
        case 32128:
            if (key == 110) LDS_CASE(32, 128, 1, 1, false, 1, 16);
            if (key == 210) LDS_CASE(32, 128, 2, 1, false, 1, 16);
            if (key == 310) LDS_CASE(32, 128, 3, 1, false, 5, 16);
            if (key == 710) LDS_CASE(32, 128, 7, 1, false, 5, 16);
            if (key == 1110) LDS_CASE(32, 128, 11, 1, false, 5, 16);
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
