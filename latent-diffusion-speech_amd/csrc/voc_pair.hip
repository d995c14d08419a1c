// One residual step of a HiFi-GAN ResBlock1 as ONE launch for the vocoder's narrow tail (16 and 32 channels: 262,144 / 131,072 samples
// per utterance):   out = c2(lrelu(c1(lrelu(x)))) + x   [+ running MRF sum, / n_kernels]
// (reference encoder/hifi_vaegan/modules/models.py:161-192, and :250-259 for the sum over the kernel sizes).
//
// The two convolutions of a step used to be two launches of conv_small / conv_gemm: 5 tensor passes over HBM per step and, for the
// short windows, more epilogue instructions than matrix work (fp32 MFMA does not co-execute with VALU: DESIGN.md section 3.1).  Here a
// workgroup owns N output frames of all channels:
//   * the input window (N + 2 (H0 + H1) frames, H0 = dilated half-width of c1, H1 = half-width of c2) is staged once into LDS with
//     LeakyReLU applied while staging;
//   * phase 1 computes c1 on N + 2 H1 frames and keeps lrelu(c1 + b1) in LDS, written as ZERO outside the utterance (that is the zero
//     padding c2 sees in the reference, not c1 of a zero-extended input);
//   * phase 2 computes c2 from that LDS tile and finishes with bias + residual (+ running sum, division) on 16-byte row segments.
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32) with FRAMES as the M rows and output channels as the N columns, so that a lane's four
// accumulators are four consecutive frames of one channel: the LDS tile of phase 1 and every global access of phase 2's epilogue are
// 16 bytes per lane.  A wave keeps the weights of BOTH convolutions for its 16 output channels in registers (2 x KT*C/4) for the
// lifetime of the persistent workgroup (32 channels at k 11: at one workgroup per CU, PairCfg::OCC); one ds_read_b32 per MFMA
// feeds the frame operand (rows padded to 16 mod 32 floats).
// N is chosen so that both phases are exactly 16 / 32 / 64 blocks of 16 frames (8 per wave, 16 for the shape that runs one workgroup per
// CU): every wave runs four (eight) two-chain iterations per phase.
// Per-element summation order: k = (tap, input channel) ascending, one accumulator chain -- independent of batch and tiling, and the
// order conv_small uses.
#include "kernels.h"

#include <stdio.h>

#include <type_traits>
#include <utility>

namespace lds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C, int KT, int DIL>
struct PairCfg {
    static constexpr int NCT = C / 16;                   // output-channel tiles of 16
    static constexpr int NFS = 4 / NCT;                  // waves per channel tile
    static constexpr int OCC = (C == 32 && KT == 11) ? 1 : 2;      // workgroups per CU (see RELOAD below)
    static constexpr int BPW = OCC == 1 ? 16 : 8;        // 16-frame blocks per wave and phase: the lone workgroup of a CU takes a tile twice as long
    static constexpr int IT = BPW / 2;                   // two-chain iterations per wave and phase
    static constexpr int NBLK = BPW * NFS;               // 16-frame blocks per phase
    static constexpr int NH = 16 * NBLK;                 // frames of the intermediate tile
    static constexpr int H1 = (KT - 1) / 2, H0 = H1 * DIL;
    static constexpr int N = (NH - 2 * H1) & ~3;         // output frames per tile (tiles start on 16-byte boundaries)
    static constexpr int WIN = NH + (KT - 1) * DIL;      // input frames the taps of phase 1 touch
    static constexpr int W4 = (WIN + 3 + 3) / 4;         // 16-byte chunks per staged row (up to 3 frames of alignment slack in front)
    static constexpr int WPX = ((4 * W4 - 16 + 31) / 32) * 32 + 16;      // LDS row strides in floats, = 16 (mod 32): the 4 k-rows of one
    static constexpr int WPH = ((NH - 16 + 31) / 32) * 32 + 16;          // ds_read_b32 (lanes 0-15 / 16-31 / ...) fall on disjoint bank groups
    static constexpr int NA = KT * C / 4;                // MFMAs (= weight registers) per 16 x 16 tile and convolution
    static constexpr int NCH = (C * W4 + 255) / 256;     // staged chunks per thread
    // Register budget: at two workgroups per CU (256 registers per wave) both convolutions' weights stay resident up to 2 x 56 (32 channels,
    // k 7).  32 channels at k 11 (2 x 88) runs ONE workgroup per CU with everything resident (350 registers): 850-890 us per launch, against
    // 917-936 us at two per CU with one convolution's weights at a time refetched from L2 behind the last matrix pass of a phase (RELOAD,
    // kept for the scalar-access instantiation), DESIGN.md section 14.8.
    static constexpr bool RELOAD = false;
    static constexpr bool LATE = C == 32 && KT == 7;     // the next tile's window is requested behind the LAST matrix pass (its 40 registers do not fit beside the weights)
    static constexpr bool HOLD = C == 16 || KT == 3 || OCC == 1;     // the window chunks' (row, LDS offset) pairs kept in registers across tiles
    static constexpr bool PIPE = C == 16 || KT == 3 || OCC == 1;     // residual quads requested one iteration ahead into a second register set (where 32 more registers fit)
    static constexpr size_t LDS_BYTES = (size_t)C * (WPX + WPH) * sizeof(float);
    static_assert(C == 16 || C == 32, "tail widths");
    static_assert(N % 4 == 0 && N > 0, "tiles start on 16-byte boundaries");
    static_assert(WPX >= 4 * W4 && WPH >= NH, "row strides");
};

// One two-chain accumulation pass over all taps.  `lds`: the workgroup's LDS; `boff`: byte offset of this lane's frame operand for tap 0,
// channel block 0, chain 0 (row stride ROW floats; chain 1 is 16 frames on; tap t is t * TSTEP frames on).  The per-channel-block bases are
// formed ONCE per pass and hidden from the optimiser, so that every read of the pass is base + immediate (no address arithmetic per tap:
// VALU instructions are paid in matrix time); operands are double-buffered per tap (the next tap's reads are in
// flight behind the current tap's MFMAs), which keeps the operands in flight at two taps' worth.
template <class F, int... I>
__device__ __forceinline__ void pair_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

template <int C4, int KT, int TSTEP, int ROW>
__device__ __forceinline__ void pair_mma_pass(const float* lds, unsigned boff, const float* w, f32x4& d0, f32x4& d1) {
    unsigned bk[C4];
#pragma unroll
    for (int kb = 0; kb < C4; ++kb) {
        bk[kb] = boff + (unsigned)(4 * kb * ROW * 4);
        asm volatile("" : "+v"(bk[kb]));
    }
    auto rd = [&](int kb, int tap, int chain) -> float {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + bk[kb] + (tap * TSTEP + 16 * chain) * 4);
    };
    float a[2][2 * C4];
#pragma unroll
    for (int kb = 0; kb < C4; ++kb) { a[0][kb] = rd(kb, 0, 0); a[0][C4 + kb] = rd(kb, 0, 1); }
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
        if (tap + 1 < KT) {
#pragma unroll
            for (int kb = 0; kb < C4; ++kb) { a[(tap + 1) & 1][kb] = rd(kb, tap + 1, 0); a[(tap + 1) & 1][C4 + kb] = rd(kb, tap + 1, 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < C4; ++kb) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap & 1][kb], w[tap * C4 + kb], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap & 1][C4 + kb], w[tap * C4 + kb], d1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int C, int KT, int DIL, bool VEC>
__global__ void __launch_bounds__(256, (PairCfg<C, KT, DIL>::OCC)) voc_pair_kernel(const VocPairArgs p, int n_tiles) {
    using Cfg = PairCfg<C, KT, DIL>;
    constexpr int N = Cfg::N, NH = Cfg::NH, H1 = Cfg::H1, H0 = Cfg::H0, W4 = Cfg::W4, WPX = Cfg::WPX, WPH = Cfg::WPH, NA = Cfg::NA, NCH = Cfg::NCH;
    constexpr int NCT = Cfg::NCT, C4 = C / 4;
    constexpr bool vec = VEC;      // T % 4 == 0: every 16-byte chunk is wholly inside or wholly outside a row
    constexpr bool RELOAD = Cfg::RELOAD || (!VEC && 2 * NA > 88);
    constexpr bool HOLD = Cfg::HOLD && VEC, PIPE = Cfg::PIPE && VEC;      // (the scalar-access instantiation keeps its registers for addresses)
    constexpr bool EXACT = NCH * 256 == C * W4;      // else the last chunk index of some threads is past the window
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xbuf = smem;                  // [C][WPX]  lrelu(x) window
    // (hbuf = smem + C * WPX: [C][WPH]  lrelu(c1(.) + b1), zero outside the utterance)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int cot = wave % NCT, fsl = wave / NCT;         // this wave's output-channel tile and its eight 16-frame blocks
    const int T = p.T;
    const int nt = (T + N - 1) / N;

    // ---- weights of this wave's 16 output channels: B operand of MFMA (tap, kb) = W[co = 16 cot + (l & 15)][ci = 4 kb + (l >> 4)][tap] ----
    // RELOAD (the scalar-access instantiations of the wide shapes): one convolution's weights at a time, the other's requested from L2 behind
    // the last matrix pass of a phase.  (Keeping three taps of each resident and streaming the other eight behind them -- so that a refill has
    // MFMAs to arrive behind -- was tried: the compiler's register allocation of that form spills ~100 registers.)
    float w1[NA], w2[RELOAD ? 1 : NA];
    auto load_w = [&](const float* wp, float* dst) {
        // element (tap, ci = 4 kb + l4, co) of [tap][ci / 8][ci % 2][Mp = 32][(ci % 8) / 2]: a compile-time part (tap, kb) in the buffer load's scalar
        // offset + ONE per-lane offset (l4, co): no per-load address registers for the loop-invariant-code pass to hoist
        constexpr int Mp = 32;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, KT * C * Mp * 4, 0x00020000);
        const int voff = (((l4 & 1) * Mp + cot * 16 + l15) * 4 + (l4 >> 1)) * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int tap = i / C4, kb = i % C4;
            const int soff = (((tap * (C / 8) + (kb >> 1)) * 2 * Mp) * 4 + (kb & 1) * 2) * 4;
            dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff, soff, 0));
        }
    };
    load_w(p.w1, w1);
    if constexpr (!RELOAD) load_w(p.w2, w2);
    const float bias1 = p.b1 ? p.b1[cot * 16 + l15] : 0.f;
    const float bias2 = p.b2 ? p.b2[cot * 16 + l15] : 0.f;
    const float slope = p.slope;      // in (0, 1): lrelu(v) = max(v, slope v)
    const bool accum = p.accum != 0;

    // Every global access of the steady state is UNCONDITIONAL (addresses clamped into the tensor, values selected afterwards): a load
    // inside a divergent branch makes the compiler wait for everything in flight at the end of the branch.
    // Window chunk i of this thread: row ci = q / W4, chunk c4 = q % W4 of q = tid + 256 i -- the same for every tile, kept packed
    // (ci << 20 | LDS byte offset) where the register budget allows, so that a tile costs one add per chunk instead of a division.
    unsigned pk[HOLD ? NCH : 1];
    auto chunk = [&](int i, int& ci, int& c4, unsigned& lo) {
        if constexpr (HOLD) {
            ci = (int)(pk[i] >> 20);
            lo = pk[i] & 0xfffffu;
            c4 = (int)(lo - (unsigned)ci * (WPX * 4)) >> 4;
        } else {
            const int q = tid + 256 * i;
            ci = q / W4; c4 = q - ci * W4;
            lo = (unsigned)(ci * WPX + 4 * c4) * 4u;
        }
    };
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + 256 * i;
            const int ci = q / W4, c4 = q - ci * W4;
            pk[i] = ((unsigned)ci << 20) | ((unsigned)(ci * WPX + 4 * c4) * 4u);
        }
    }
    f32x4 xr[NCH];
    auto fetch = [&](int tile) {          // window chunks of this thread, all issued together; zeros outside [0, T)
        const int b = tile / nt, t0 = (tile - b * nt) * N;
        const float* xb = p.x + (long long)b * C * T;
        const int s0 = t0 - H1 - H0;
        const int s_al = (s0 >= 0) ? (s0 & ~3) : -(((-s0) + 3) & ~3);
        const bool inner = s_al >= 0 && s_al + 4 * W4 <= T;      // (wave-uniform) the whole window is inside the row
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ci, c4; unsigned lo;
            chunk(i, ci, c4, lo);
            const bool live = EXACT || i < NCH - 1 || tid + 256 * i < C * W4;
            const int s = s_al + 4 * c4;
            if constexpr (vec) {
                if (inner) {
                    xr[i] = *reinterpret_cast<const f32x4*>(xb + (live ? ci * T + s : 0));
                } else {
                    const bool in = live && s >= 0 && s < T;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (in ? ci * T + s : 0));
                    xr[i] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (live) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (s + e >= 0 && s + e < T) ? xb[ci * T + s + e] : 0.f;
                }
                xr[i] = v;
            }
        }
    };
    auto commit = [&]() {                 // LeakyReLU once per element, then into the LDS window
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ci, c4; unsigned lo;
            chunk(i, ci, c4, lo);
            f32x4 v = xr[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
            if (EXACT || i < NCH - 1 || tid + 256 * i < C * W4) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xbuf) + lo) = v;
        }
    };

    // per-lane constants of the two phases
    const unsigned x_boff = (unsigned)((l4 * WPX + l15) * 4);                       // frame operand of phase 1, before the tile's alignment offset
    const unsigned h_boff = (unsigned)((C * WPX + l4 * WPH + l15) * 4);             // ... of phase 2
    const unsigned h_woff = (unsigned)((C * WPX + (cot * 16 + l15) * WPH + 4 * l4) * 4);      // where this lane's quad of phase 1's output goes
    const int rowb = (cot * 16 + l15) * T + 4 * l4;                                 // this lane's quad in the plain tensors, before t0 + block offset

    if ((int)blockIdx.x < n_tiles) { fetch(blockIdx.x); commit(); }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / nt, t0 = (tile - b * nt) * N;
        const int s0 = t0 - H1 - H0;
        const int s_al = (s0 >= 0) ? (s0 & ~3) : -(((-s0) + 3) & ~3);
        const int off = s0 - s_al;                                  // 0..3
        const int Tv = p.vlen ? (p.vlen[b] < T ? p.vlen[b] : T) : T;      // ragged batch: the utterance ends here
        __syncthreads();      // the window is committed; every wave is past the previous tile's phase 2 (its reads of hbuf)

        // ---- phase 1: h[ci][j] = lrelu(b1 + sum_(tap, cj) W1 * lrelu(x)[cj][t0 - H1 + j + (tap - H1) DIL]) for j < NH, zero outside [0, Tv) ----
        {
            const int u0 = t0 - H1;                                 // frame of h column 0
            const bool inner = u0 >= 0 && u0 + NH <= Tv;            // (wave-uniform) no column of the tile is outside the utterance
#pragma unroll 1
            for (int it = 0; it < Cfg::IT; ++it) {
                const int j0 = 16 * (Cfg::BPW * fsl + 2 * it);
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
                pair_mma_pass<C4, KT, DIL, WPX>(smem, x_boff + (unsigned)((off + j0) * 4), w1, d0, d1);
                // D[row = frame 4 (l >> 4) + r][col = channel l & 15]
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 d = half ? d1 : d0;
                    f32x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float y = d[r] + bias1; h[r] = fmaxf(y, y * slope); }
                    if (!inner) {
                        const unsigned u = (unsigned)(u0 + j0 + 16 * half + 4 * l4);      // (negative frames wrap to huge values)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = (u + r < (unsigned)Tv) ? h[r] : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + h_woff + (j0 + 16 * half) * 4) = h;
                }
            }
        }
        if constexpr (RELOAD) load_w(p.w2, w1);      // c2's weights travel behind the barrier
        __syncthreads();      // hbuf is complete; xbuf is free
        // ---- phase 2: out[co][t0 + j] = b2 + sum_(tap, ci) W2 * h[ci][j + tap] + x[co][t0 + j]  (j < N) ----
        {
            const int next = tile + (int)gridDim.x;
            const bool has_next = next < n_tiles;
            if constexpr (!Cfg::LATE) fetch(has_next ? next : tile);      // the next tile's window: requested now, committed behind the last matrix pass of this phase
            const float* xrow = p.x + (long long)b * C * T;      // (wave-uniform bases, 32-bit per-lane offsets)
            float* orow = p.out + (long long)b * C * T;
            const float* arow = accum ? orow : xrow;             // (no running sum: the second load repeats the first, its value is dropped)
            const bool inner = t0 + NH <= Tv;                    // (wave-uniform) every frame of the tile's blocks is inside the utterance (and the row)
            // residual (and running-sum) quads of both chains of iteration `it`: requested one matrix pass before they are consumed
            auto request = [&](int it, f32x4* r_, f32x4* a_) {
                const int j0 = 16 * (Cfg::BPW * fsl + 2 * it);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int jb = j0 + 16 * half;               // (wave-uniform)
                    if constexpr (vec) {                         // (T and N are multiples of 4: a quad is wholly inside or outside the row)
                        int o = rowb + t0 + jb;
                        if (!inner) o = (t0 + jb + 4 * l4 < T) ? o : rowb - 4 * l4;      // (blocks past the row's end: the row's first quad, whose value is not stored)
                        r_[half] = *reinterpret_cast<const f32x4*>(xrow + o);
                        a_[half] = *reinterpret_cast<const f32x4*>(arow + o);
                    } else {
                        r_[half] = f32x4{0.f, 0.f, 0.f, 0.f};
                        a_[half] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int t = t0 + jb + 4 * l4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (jb + 4 * l4 < N && t + r < T) { r_[half][r] = xrow[rowb + t0 + jb + r]; a_[half][r] = arow[rowb + t0 + jb + r]; }
                        }
                    }
                }
            };
            // one iteration: matrix pass, the NEXT iteration's request into the other register set, then this iteration's epilogue
            auto step = [&](int it, auto last_c, const f32x4* rc, const f32x4* ac, f32x4* rn, f32x4* an) {      // (rc / ac: this iteration's quads; rn / an: where the request goes)
                const int j0 = 16 * (Cfg::BPW * fsl + 2 * it);
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!PIPE) request(it, rn, an);      // (one register set: requested before this iteration's own pass)
                pair_mma_pass<C4, KT, 1, WPH>(smem, h_boff + (unsigned)(j0 * 4), RELOAD ? w1 : w2, d0, d1);      // (the last block's taps stay inside the row: j0 + 31 + KT - 1 < WPH)
                if constexpr (PIPE) request(it < Cfg::IT - 1 ? it + 1 : Cfg::IT - 1, rn, an);      // (the last iteration repeats its own request: the count of loads in flight stays fixed)
                if constexpr (!PIPE) { rc = rn; ac = an; }
                if constexpr (decltype(last_c)::value) {                // (the tile's last iteration, known at compile time: nothing here is loop-carried)
                    if constexpr (RELOAD) load_w(p.w1, w1);    // c1's weights for the next tile
                    if constexpr (Cfg::LATE) fetch(has_next ? next : tile);
                    else if (has_next) commit();
                }
                f32x4 y[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 d = half ? d1 : d0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[half][r] = (d[r] + bias2) + rc[half][r];
                    if (accum) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[half][r] += ac[half][r];
                    }
                    if (p.out_div != 1.0f) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[half][r] = y[half][r] / p.out_div;
                    }
                    if (!inner) {
                        const int t = t0 + j0 + 16 * half + 4 * l4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[half][r] = (t + r < Tv) ? y[half][r] : 0.f;      // (ragged batch: zeros beyond the utterance's length)
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int j = j0 + 16 * half + 4 * l4;
                    const int t = t0 + j;
                    if (j < N && t < T) {
                        if constexpr (vec) {
                            *reinterpret_cast<f32x4*>(orow + rowb + t0 + j0 + 16 * half) = y[half];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (t + r < T) orow[rowb + t0 + j0 + 16 * half + r] = y[half][r];
                        }
                    }
                }
            };
            f32x4 rA[2], aA[2], rB[2], aB[2];      // two register sets, alternating: no copy of a load still in flight
            if constexpr (PIPE) {
                request(0, rA, aA);
                // straight-line code (a loop header would merge two register assignments of the sets in flight and make the compiler wait for
                // fresh loads there): iterations 0 .. IT-1, the register sets alternating
                pair_static_for([&](auto ic) {
                    constexpr int it = decltype(ic)::value;
                    if constexpr (it & 1) step(it, std::bool_constant<it == Cfg::IT - 1>{}, rB, aB, rA, aA);
                    else step(it, std::bool_constant<it == Cfg::IT - 1>{}, rA, aA, rB, aB);
                }, std::make_integer_sequence<int, Cfg::IT>{});
            } else {
#pragma unroll 1
                for (int it = 0; it < Cfg::IT - 1; ++it) step(it, std::false_type{}, rA, aA, rA, aA);
                step(Cfg::IT - 1, std::true_type{}, rA, aA, rA, aA);
            }
            if constexpr (Cfg::LATE) {
                if (has_next) commit();            // (behind the last epilogue: the other resident workgroup's matrix work covers the wait)
            }
        }
    }
}

static thread_local char g_pcfg[64] = "";
const char* voc_pair_last_config() { return g_pcfg; }

template <int C, int KT, int DIL>
static hipError_t launch_pair_cfg(const VocPairArgs& a, hipStream_t s) {
    using Cfg = PairCfg<C, KT, DIL>;
    auto kern = (a.T & 3) ? voc_pair_kernel<C, KT, DIL, false> : voc_pair_kernel<C, KT, DIL, true>;
    static_assert(Cfg::LDS_BYTES <= (size_t)(160 / Cfg::OCC) * 1024, "OCC workgroups per CU");
    if (Cfg::LDS_BYTES > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0}, attr_done2{0};
        hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(voc_pair_kernel<C, KT, DIL, true>), attr_done);
        if (e == hipSuccess) e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(voc_pair_kernel<C, KT, DIL, false>), attr_done2);
        if (e != hipSuccess) return e;
    }
    const long long n_tiles = (long long)a.B * ((a.T + Cfg::N - 1) / Cfg::N);
    if (n_tiles > 0x7fffffffll) return hipErrorInvalidValue;
    const int grid = n_tiles < 256 * Cfg::OCC ? (int)n_tiles : 256 * Cfg::OCC;      // persistent: two workgroups per CU stride over the tiles, weights loaded once
    snprintf(g_pcfg, sizeof(g_pcfg), "C%d KT%d D%d N%d grid %d", C, KT, DIL, Cfg::N, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, s, a, (int)n_tiles);
    return hipGetLastError();
}

bool voc_pair_applies(int C, int KT, int dil) {
    return (C == 16 || C == 32) && (KT == 3 || KT == 7 || KT == 11) && (dil == 1 || dil == 3 || dil == 5);
}

hipError_t launch_voc_pair(const VocPairArgs& a, hipStream_t s) {
    if (!voc_pair_applies(a.C, a.KT, a.dil) || a.Mp1 != 32 || a.Mp2 != 32 || a.x == a.out || !a.x || !a.out || !a.w1 || !a.w2 || a.B <= 0 || a.T <= 0) return hipErrorInvalidValue;
#define PCASE(C_, KT_, D_) if (a.C == C_ && a.KT == KT_ && a.dil == D_) return launch_pair_cfg<C_, KT_, D_>(a, s)
#define PROW(C_, KT_) PCASE(C_, KT_, 1); PCASE(C_, KT_, 3); PCASE(C_, KT_, 5)
    PROW(16, 3); PROW(16, 7); PROW(16, 11);
    PROW(32, 3); PROW(32, 7); PROW(32, 11);
#undef PROW
#undef PCASE
    return hipErrorInvalidValue;
}

}  // namespace lds
