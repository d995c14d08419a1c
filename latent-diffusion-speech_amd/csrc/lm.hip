// text2semantic RoFormer on gfx950 (reference text2semantic/roformer/roformer.py:59-255 over HF transformers RoFormerModel /
// RoFormerForCausalLM + GenerationMixin): phone/tone encoder prefill and the key/value-cached autoregressive decode with greedy or
// top-k / top-p sampling.  The model is tiny (hidden 256, 4 + 1 layers, 4.3 M parameters) and the caller decodes one to a few
// utterances, so every step is launch- and latency-bound: the design keeps the whole loop on the stream (no host round trip except
// an EOS poll every few steps), fuses bias / residual / LayerNorm / GELU into the matrix-vector kernels and reads each weight once
// per step with coalesced loads (weights stored transposed [K][M]).  Accumulation is one fp32 fmaf chain over k in ascending order
// per output, independent of the batch size.
#include "../../include/lds.h"
#include "kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace lds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wave64 reductions on the DPP path (row shifts + row broadcasts: VALU speed, no LDS crossbar), result in every lane.  A butterfly of
// __shfl_xor is six dependent ds_bpermute round trips (~0.3 us); the decode step runs ~40 of these reductions back to back.
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, BOUND); }
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
static __device__ __forceinline__ float wave_sum(float v) {      // fixed order; the total forms in lane 63
    v += dpp_f<0x111, 0xf, true>(0.f, v);      // row_shr:1
    v += dpp_f<0x112, 0xf, true>(0.f, v);      // row_shr:2
    v += dpp_f<0x114, 0xf, true>(0.f, v);      // row_shr:4
    v += dpp_f<0x118, 0xf, true>(0.f, v);      // row_shr:8: lane 15 of every 16-lane row = the row's sum
    v += dpp_f<0x142, 0xa, false>(0.f, v);     // row_bcast:15 into rows 1 and 3
    v += dpp_f<0x143, 0xc, false>(0.f, v);     // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
static __device__ __forceinline__ float wave_max(float v) {      // lanes without a source keep their own value (old = v)
    v = fmaxf(v, dpp_f<0x111, 0xf, false>(v, v));
    v = fmaxf(v, dpp_f<0x112, 0xf, false>(v, v));
    v = fmaxf(v, dpp_f<0x114, 0xf, false>(v, v));
    v = fmaxf(v, dpp_f<0x118, 0xf, false>(v, v));
    v = fmaxf(v, dpp_f<0x142, 0xa, false>(v, v));
    v = fmaxf(v, dpp_f<0x143, 0xc, false>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
static __device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0x111, 0xf, false>(v, v));
    v = min(v, dpp_i<0x112, 0xf, false>(v, v));
    v = min(v, dpp_i<0x114, 0xf, false>(v, v));
    v = min(v, dpp_i<0x118, 0xf, false>(v, v));
    v = min(v, dpp_i<0x142, 0xa, false>(v, v));
    v = min(v, dpp_i<0x143, 0xc, false>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
// sum over the NT threads of a workgroup (fixed order), result in every thread; red has NT / 64 <= 16 slots
template <int NT>
static __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
static __device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
static __device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- embeddings: x = LN(word[tok] + type[tt]); encoder: x = LN(x + spk[spk_id] + type[0]) as well (roformer.py:196, the
//      embeddings module is applied a second time to inputs_embeds).  One workgroup per token, H <= 1024. ----
__global__ void __launch_bounds__(256) lm_embed_kernel(const float* __restrict__ word, const float* __restrict__ type, const float* __restrict__ spk,
                                                       const float* __restrict__ g, const float* __restrict__ bta, const int64_t* __restrict__ tok,
                                                       int tok_stride, const int64_t* __restrict__ tt, const int64_t* __restrict__ spk_id, int twice,
                                                       float eps, int H, int vocab, int type_vocab, int spk_rows, float* __restrict__ out) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    long long t = tok[(long long)n * tok_stride];
    long long ty = tt ? tt[n] : 0;
    t = (t < 0) ? 0 : (t >= vocab ? vocab - 1 : t);                 // ids are validated on the host; this only keeps a corrupted id from faulting
    ty = (ty < 0) ? 0 : (ty >= type_vocab ? type_vocab - 1 : ty);
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        v[i] = (c < H) ? word[t * H + c] + type[ty * H + c] : 0.f;
        s += v[i];
    }
    for (int pass = 0; pass < (twice ? 2 : 1); ++pass) {
        const float mean = block_sum256(s, red) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = (tid + 256 * i < H) ? v[i] - mean : 0.f; q += d * d; }
        const float rstd = 1.0f / sqrtf(block_sum256(q, red) / (float)H + eps);
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (c < H) {
                v[i] = (v[i] - mean) * rstd * g[c] + bta[c];
                if (pass == 0 && twice) {
                    long long sp = (spk && spk_id) ? spk_id[n] : 0;
                    sp = (sp < 0) ? 0 : (sp >= spk_rows ? spk_rows - 1 : sp);
                    v[i] += (spk && spk_id ? spk[sp * H + c] : 0.f) + type[c];
                }
            }
            s += (c < H) ? v[i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (tid + 256 * i < H) out[(long long)n * H + tid + 256 * i] = v[i];
}

// ---- Y[n][m] = epi(sum_k LNopt(X)[n][k] * Wt[k][m] + b[m]) for a tile of 8 tokens per workgroup.  A decode step has 1-8 rows, so the
//      kernels are a latency problem, not a FLOP problem: a workgroup is 64 output columns x 16 K-slices (each thread walks K/16 weights
//      with 16 coalesced loads in flight), the partial sums meet through LDS in a fixed order (slice 0 adds slices 1, 2, ...). ----
// rotary embedding + cache append of the self-attention q | k | v projection (EPI 4): table row = [sin(d/2) | cos(d/2)], pairs (2p, 2p+1)
// (modeling_roformer.py:220-245); rotated q goes to Y, rotated k and v to kc / vc [B][heads][cap][d] at slot pos0 + l (row n = b * L + l)
struct LmRope { const float* table; float* kc; float* vc; int pos0, L, heads, cap; };

// ---- Deferred LayerNorm.  A linear whose epilogue normalises complete rows needs one workgroup per row tile, i.e. ONE CU streaming the
//      whole weight matrix (~55 GB/s: 11 us for 256 x 256, four of them per decode step).  Here the producing linear keeps its
//      M / 64 column workgroups (5 us) and stores the PRE-normalisation rows; whoever reads them next normalises on the way in:
//        * a consumer linear stages its 8 input rows in LDS anyway -- it computes their mean / variance there (xg, xb != null);
//        * a consumer that adds them as the residual stages those rows as well (rg, rb != null).
//      The statistics are recomputed by every consuming workgroup (8 x 256 values: two block reductions), in a fixed order.
//      EPI 0: none, 1: GELU, 4: rotary + cache append (LmRope), 5: y + residual. ----
template <int NT>
static __device__ __forceinline__ void lm_rows_ln_inplace(float* buf, int L, const float* __restrict__ g, const float* __restrict__ bta, float eps, float* red8) {
    // wave j (< 8) owns row j: two wave reductions, no workgroup-wide reduction; the (mean, rstd) pairs are published through LDS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave < 8) {
        float sum = 0.f;
        for (int k = lane; k < L; k += 64) sum += buf[8 * k + wave];
        const float mean = wave_sum(sum) / (float)L;
        float q = 0.f;
        for (int k = lane; k < L; k += 64) { const float d = buf[8 * k + wave] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)L + eps);
        if (lane == 0) { red8[2 * wave] = mean; red8[2 * wave + 1] = rstd; }
    }
    __syncthreads();
    float mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = red8[2 * j]; rstd[j] = red8[2 * j + 1]; }
    for (int k = tid; k < L; k += NT) {
        const float gk = g[k], bk = bta[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[8 * k + j] = (buf[8 * k + j] - mean[j]) * rstd[j] * gk + bk;
    }
    __syncthreads();
}

template <int EPI>
__global__ void __launch_bounds__(1024) lm_linear_dln_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ xg, const float* __restrict__ xb,
                                                            const float* __restrict__ Wt, const float* __restrict__ bias, const float* __restrict__ R,
                                                            const float* __restrict__ rg, const float* __restrict__ rb, float eps, float* __restrict__ Y, int ldy,
                                                            int N, int K, int M, const LmRope rope) {
    constexpr int COLS = 64, KS = 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // xs [K][8], part [KS-1][COLS][8], rs [M][8] (residual rows), red8 [16][8]
    float* xs = sm;
    float* part = sm + (size_t)K * 8;
    float* rs = part + (size_t)(KS - 1) * COLS * 8;
    float* red8 = rs + (size_t)((EPI == 5) ? M : 0) * 8;
    const int tid = threadIdx.x, col = tid % COLS, ks = tid / COLS;
    const int m = blockIdx.x * COLS + col, n0 = blockIdx.y * 8;
    for (int i = tid; i < K * 8; i += 1024) {
        const int k = i >> 3, j = i & 7;
        xs[i] = (n0 + j < N) ? X[(long long)(n0 + j) * ldx + k] : 0.f;
    }
    if constexpr (EPI == 5) {
        for (int i = tid; i < M * 8; i += 1024) {
            const int k = i >> 3, j = i & 7;
            rs[i] = (n0 + j < N) ? R[(long long)(n0 + j) * M + k] : 0.f;
        }
    }
    __syncthreads();
    if (xg) lm_rows_ln_inplace<1024>(xs, K, xg, xb, eps, red8);
    if constexpr (EPI == 5) {
        if (rg) lm_rows_ln_inplace<1024>(rs, M, rg, rb, eps, red8);
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kq = K / KS, k0 = ks * kq;                            // K % 256 == 0 (checked by the launcher)
    if (m < M) {
        const float* wp = Wt + (long long)k0 * M + m;
        for (int kk = 0; kk < kq; kk += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = wp[(long long)(kk + u) * M];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xs + 8 * (k0 + kk + u)), b = *reinterpret_cast<const f32x4*>(xs + 8 * (k0 + kk + u) + 4);
                acc[0] = fmaf(w[u], a[0], acc[0]); acc[1] = fmaf(w[u], a[1], acc[1]); acc[2] = fmaf(w[u], a[2], acc[2]); acc[3] = fmaf(w[u], a[3], acc[3]);
                acc[4] = fmaf(w[u], b[0], acc[4]); acc[5] = fmaf(w[u], b[1], acc[5]); acc[6] = fmaf(w[u], b[2], acc[6]); acc[7] = fmaf(w[u], b[3], acc[7]);
            }
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[((ks - 1) * COLS + col) * 8 + j] = acc[j];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll 1
    for (int q0 = 0; q0 < KS - 1; q0 += 5) {      // five slices in flight at a time (register budget of a 1024-thread workgroup)
        float t[5][8];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) t[q][j] = part[((q0 + q) * COLS + col) * 8 + j];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t[q][j];
    }
    const float bm = (m < M && bias) ? bias[m] : 0.f;
    float y[8];
    bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        y[j] = acc[j] + bm;
        ok[j] = m < M && n0 + j < N;
        if constexpr (EPI == 1) y[j] = gelu_erf(y[j]);
        if constexpr (EPI == 5) { if (m < M) y[j] += rs[8 * m + j]; }
    }
    if constexpr (EPI == 4) {
        // slice 0 is exactly wave 0 and M % 64 == 0: the rotation partner of column m sits in the neighbouring lane
        const int H = M / 3, d = H / rope.heads, hp = d >> 1;
        const int sec = m / H, c = m - sec * H, hd = c / d, e = c - hd * d;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = (n0 + j < N) ? n0 + j : n0;
            const int b = n / rope.L, pos = rope.pos0 + (n - b * rope.L);
            const float sn = rope.table[(long long)pos * d + (e >> 1)], cs = rope.table[(long long)pos * d + hp + (e >> 1)];
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[j]), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
            const float rot = (e & 1) ? y[j] * cs + other * sn : y[j] * cs - other * sn;
            const long long co = (((long long)b * rope.heads + hd) * rope.cap + pos) * d + e;
            if (ok[j]) {
                if (sec == 0) Y[(long long)n * ldy + m] = rot;
                else if (sec == 1) rope.kc[co] = rot;
                else rope.vc[co] = y[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ok[j]) Y[(long long)(n0 + j) * ldy + m] = y[j];
    }
}

// LayerNorm of rows [N][H] (the encoder's final states, which leave the library normalised): one workgroup per row, H <= 1024
__global__ void __launch_bounds__(256) lm_ln_rows_kernel(const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ bta, float eps, int H,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    const long long n = blockIdx.x;
    const int tid = threadIdx.x;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; v[i] = (c < H) ? in[n * H + c] : 0.f; s += v[i]; }
    const float mean = block_sum256(s, red) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = (tid + 256 * i < H) ? v[i] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, red) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; if (c < H) out[n * H + c] = (v[i] - mean) * rstd * g[c] + bta[c]; }
}

// cross-attention keys / values of the encoder states into the cache layout: kv [N][2H] (k | v) -> kc / vc [B][heads][cap][d]
__global__ void __launch_bounds__(256) lm_kv_pack_kernel(const float* __restrict__ kv, int B, int L, int H, int heads, float* __restrict__ kc,
                                                         float* __restrict__ vc, int cap) {
    const int d = H / heads;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * L * H) return;
    const int c = (int)(i % H);
    const long long n = i / H;
    const int l = (int)(n % L), b = (int)(n / L), hd = c / d, e = c - hd * d;
    const long long co = (((long long)b * heads + hd) * cap + l) * d + e;
    kc[co] = kv[n * 2 * H + c];
    vc[co] = kv[n * 2 * H + H + c];
}

// ---- attention of one query per workgroup: softmax(q K^T / sqrt(d)) V over Lk cached keys (no mask: the decoder's cache holds exactly
//      the causal context, the encoder is bidirectional).  q [N][ldq] (head slice at hd*d), out [N][H].  The four waves take alternate
//      64-key blocks; inside a wave, phase 1 has one key per lane (scores), phase 2 one output channel per lane with the two half-waves on
//      alternate keys; the waves' (max, sum, partial output) meet through LDS in a fixed order.  d <= 32. ----
__global__ void __launch_bounds__(256) lm_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kc, const float* __restrict__ vc, int cap,
                                                      int Lq, int Lk_all, const int* __restrict__ klen, int H, int heads, float* __restrict__ out) {
    extern __shared__ float sc[];                      // [Lk rounded to 64] scores, then 4 x (max, sum, acc[32])
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x;                       // (b, l, head)
    const int d = H / heads, Lkp = (Lk_all + 63) & ~63;
    const int hd = item % heads;
    const long long n = item / heads;                  // token index b * Lq + l
    const int b = (int)(n / Lq);
    // padding mask (reference roformer.py:209-236: attention_mask on the encoder's keys, encoder_attention_mask on the cross-attention's):
    // right-padded rows, so the mask of batch row b is "keys [0, klen[b])"; masked keys get probability exactly 0 as with HF's -inf bias
    const int kl = klen ? klen[b] : Lk_all;
    const int Lk = (kl < Lk_all) ? (kl < 1 ? 1 : kl) : Lk_all;
    float* mrg = sc + Lkp;
    float qr[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) qr[e] = (e < d) ? q[n * ldq + hd * d + e] : 0.f;
    const float* kb = kc + ((long long)b * heads + hd) * cap * d;
    const float* vb = vc + ((long long)b * heads + hd) * cap * d;
    const float scale = 1.0f / sqrtf((float)d);
    float mx = -INFINITY;
    for (int k0 = wave * 64; k0 < Lk; k0 += 256) {
        const int key = k0 + lane;
        float dot = -INFINITY;
        if (key < Lk) {
            dot = 0.f;
            if (d == 32) {      // the key row as eight 16-byte loads (a lane per key: every load instruction gathers 64 rows)
                const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)key * 32);
                f32x4 kv[8];
#pragma unroll
                for (int e4 = 0; e4 < 8; ++e4) kv[e4] = kr[e4];
#pragma unroll
                for (int e = 0; e < 32; ++e) dot = fmaf(qr[e], kv[e >> 2][e & 3], dot);
            } else {
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    if (e < d) dot = fmaf(qr[e], kb[(long long)key * d + e], dot);
            }
            dot *= scale;
        }
        sc[key] = dot;
        mx = fmaxf(mx, dot);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    if (mx > -INFINITY) {
        for (int k0 = wave * 64; k0 < Lk; k0 += 256) {
            const int key = k0 + lane;
            const float p = (key < Lk) ? expf(sc[key] - mx) : 0.f;
            sc[key] = p;
            sum += p;
        }
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's LDS writes are done before its reads below
    const int e = lane & 31, half = lane >> 5;
    float acc = 0.f;
    if (e < d && mx > -INFINITY)
        for (int k0 = wave * 64; k0 < Lk; k0 += 256) {
            const int kend = (k0 + 64 < Lk) ? k0 + 64 : Lk;
#pragma unroll 8
            for (int key = k0 + half; key < kend; key += 2) acc = fmaf(sc[key], vb[(long long)key * d + e], acc);
        }
    acc += __shfl_xor(acc, 32, 64);
    if (lane == 0) { mrg[wave * 34] = mx; mrg[wave * 34 + 1] = sum; }
    if (half == 0) mrg[wave * 34 + 2 + e] = acc;
    __syncthreads();
    if (wave == 0 && half == 0 && e < d) {
        float m = fmaxf(fmaxf(mrg[0], mrg[34]), fmaxf(mrg[68], mrg[102]));
        float tot = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (mrg[w * 34] > -INFINITY) ? expf(mrg[w * 34] - m) : 0.f;
            tot += mrg[w * 34 + 1] * f;
            o += mrg[w * 34 + 2 + e] * f;
        }
        out[n * H + hd * d + e] = o / tot;
    }
}

// ---- next-token choice per sequence (HF GenerationMixin._sample with RepetitionPenalty -> Temperature -> TopK -> TopP, then
//      softmax + one draw; greedy = argmax).  The draw is the inverse-CDF rule over the vocabulary order with a caller-supplied
//      uniform (torch.multinomial's own stream cannot be reproduced outside torch).  One workgroup per sequence, V <= 256 * 32. ----
constexpr int kMaxTopK = 64;
template <int NI>      // NI * 256 >= V: vocabulary slots per thread
__global__ void __launch_bounds__(256) lm_sample_kernel(const float* __restrict__ logits, int V, int do_sample, int top_k, float top_p, float inv_temp,
                                                        float rep_pen, const float* __restrict__ uniforms, int64_t* __restrict__ tokens, int cap_tokens,
                                                        int step, int* __restrict__ unfinished, int eos, int pad, int* __restrict__ any_unfinished,
                                                        const float* __restrict__ word, const float* __restrict__ type, const float* __restrict__ eg,
                                                        const float* __restrict__ eb, float eps, int H, float* __restrict__ xnext) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    __shared__ int chosen;
    __shared__ float topv[kMaxTopK], pr[kMaxTopK];
    __shared__ int topi[kMaxTopK], ord[kMaxTopK];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = logits + (long long)b * V;
    int64_t* seq = tokens + (long long)b * cap_tokens;
    // thread 0 needs these at the very end: fetched now, their latency hides behind the top-k rounds
    const float u_draw = (tid == 0 && do_sample) ? uniforms[b] : 0.f;
    const int alive = (tid == 0) ? unfinished[b] : 0;
    float v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = tid + 256 * i;
        v[i] = (c < V) ? lg[c] : -INFINITY;
    }
    if (rep_pen != 1.0f) {      // RepetitionPenaltyLogitsProcessor: every distinct token of the sequence so far is penalised once (gather / scatter)
        unsigned done = 0;
        for (int j = 0; j <= step; ++j) {
            const int t = (int)seq[j];
            if ((t & 255) == tid && !((done >> (t >> 8)) & 1u)) {
                done |= 1u << (t >> 8);
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (i == (t >> 8)) v[i] = (v[i] < 0.f) ? v[i] * rep_pen : v[i] / rep_pen;
            }
        }
    }
    if (do_sample && inv_temp != 1.0f) {
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] *= inv_temp;
    }
    // TopKLogitsWarper removes the scores BELOW the k-th largest (modeling: `scores < topk(scores, k)[..., -1]`), so every score tied with the
    // k-th one survives too: the selection runs on past k rounds while the next maximum still equals the k-th value (up to kMaxTopK survivors).
    const int rounds = do_sample ? top_k : 1;
    int n_sel = rounds;
    for (int r = 0; r < (do_sample ? kMaxTopK : 1); ++r) {      // r-th largest remaining value, ties to the lowest index
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + 256 * i;
            if (v[i] > bv || (v[i] == bv && c < bi && v[i] > -INFINITY)) { bv = v[i]; bi = c; }
        }
        {   // wave argmax: the largest value, then the lowest index among the lanes that hold it (NaNs never win)
            const float m = wave_max(bv);
            bi = wave_min_i((bv == m) ? bi : 0x7fffffff);
            bv = m;
        }
        __syncthreads();
        if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
        __syncthreads();
        bv = rv[0]; bi = ri[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
        if (r >= rounds) {      // (bv, topv are the same in every thread: the loop's exit is uniform)
            __syncthreads();
            if (!(bv == topv[rounds - 1]) || !(bv > -INFINITY)) { n_sel = r; break; }
            n_sel = r + 1;
        }
        if (tid == 0) { topv[r] = bv; topi[r] = bi; }
        if ((bi & 255) == tid) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (i == (bi >> 8)) v[i] = -INFINITY;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int next;
        if (!do_sample) {
            next = topi[0];
        } else {
            // softmax over the survivors (descending order in topv), optional nucleus cut, renormalise, draw in vocabulary order
            int kept = n_sel;
            float sum = 0.f;
            for (int j = 0; j < kept; ++j) { pr[j] = expf(topv[j] - topv[0]); sum += pr[j]; }
            if (top_p < 1.0f) {      // TopPLogitsWarper: drop the tail whose ascending cumulative probability stays <= 1 - top_p
                float tail = 0.f;
                int cut = kept;
                for (int j = kept - 1; j >= 1; --j) {
                    tail += pr[j] / sum;
                    if (tail <= 1.0f - top_p) cut = j; else break;
                }
                kept = cut;
                sum = 0.f;
                for (int j = 0; j < kept; ++j) sum += pr[j];
            }
            for (int j = 0; j < kept; ++j) pr[j] = pr[j] / sum;
            // vocabulary order (insertion sort of <= 64 entries), running sum, first index whose sum exceeds u
            for (int j = 0; j < kept; ++j) ord[j] = j;
            for (int j = 1; j < kept; ++j) {
                const int o = ord[j];
                int q = j - 1;
                while (q >= 0 && topi[ord[q]] > topi[o]) { ord[q + 1] = ord[q]; --q; }
                ord[q + 1] = o;
            }
            const float u = u_draw;
            float c = 0.f;
            next = topi[ord[kept - 1]];
            for (int j = 0; j < kept; ++j) {
                c += pr[ord[j]];
                if (c > u) { next = topi[ord[j]]; break; }
            }
        }
        if (next < 0 || next >= V) next = pad;                          // all-NaN logits: nothing compares greater than -inf
        if (!alive) next = pad;
        seq[step + 1] = next;
        chosen = next;
        if (alive && next == eos) unfinished[b] = 0;
        if (alive && next != eos) atomicOr(any_unfinished + step, 1);      // flag array indexed by step, zeroed by the host wrapper
    }
    // the next decode step's input row, LayerNorm(word[next] + type[0]) (what lm_embed_kernel computes), without a launch of its own
    if (xnext) {
        __syncthreads();
        long long t = chosen;
        t = (t < 0) ? 0 : (t >= V ? V - 1 : t);
        float ev[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            ev[i] = (c < H) ? word[t * H + c] + type[c] : 0.f;
            s += ev[i];
        }
        const float mean = block_sum256(s, rv) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dd = (tid + 256 * i < H) ? ev[i] - mean : 0.f; q += dd * dd; }
        const float rstd = 1.0f / sqrtf(block_sum256(q, rv) / (float)H + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (c < H) xnext[(long long)b * H + c] = (ev[i] - mean) * rstd * eg[c] + eb[c];
        }
    }
}

// ---- the same choice with NO top-k filter (HF: top_k = None / 0): RepetitionPenalty -> Temperature -> [TopP] -> softmax -> one draw over the whole
//      vocabulary.  An optional path (the reference's caller passes top_k = 5): the softmax normalisation and the inverse-CDF walk are serial loops of
//      one thread over an LDS copy of the row (~4 k adds each: tens of microseconds), in vocabulary order -- exactly what a sequential float32
//      cumsum computes, so the oracle restates it to the bit.  TopP (modeling: sort ascending, drop while the cumulative probability stays
//      <= 1 - top_p, always keep the largest): the cut is found as the largest probability value t with sum{p_j <= t} <= 1 - top_p by bisection
//      over the ordered bit patterns of the probabilities (31 block reductions); probabilities tied with the cut are dropped together. ----
__global__ void __launch_bounds__(256) lm_sample_full_kernel(const float* __restrict__ logits, int V, float top_p, float inv_temp, float rep_pen,
                                                             const float* __restrict__ uniforms, int64_t* __restrict__ tokens, int cap_tokens, int step,
                                                             int* __restrict__ unfinished, int eos, int pad, int* __restrict__ any_unfinished,
                                                             const float* __restrict__ word, const float* __restrict__ type, const float* __restrict__ eg,
                                                             const float* __restrict__ eb, float eps, int H, float* __restrict__ xnext) {
    extern __shared__ float prob[];      // [V]
    __shared__ float rv[4];
    __shared__ int chosen;
    __shared__ float sh_max, sh_z;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (long long)b * V;
    int64_t* seq = tokens + (long long)b * cap_tokens;
    for (int c = tid; c < V; c += 256) prob[c] = lg[c];
    __syncthreads();
    if (rep_pen != 1.0f && tid == 0) {      // every distinct token of the sequence so far, once (a token met again finds its score already changed: marked by a NaN-free sentinel list)
        for (int j = 0; j <= step; ++j) {
            const int t = (int)seq[j];
            bool seen = false;
            for (int q = 0; q < j; ++q) seen = seen || ((int)seq[q] == t);
            if (!seen && t >= 0 && t < V) prob[t] = (prob[t] < 0.f) ? prob[t] * rep_pen : prob[t] / rep_pen;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int c = tid; c < V; c += 256) {
        if (inv_temp != 1.0f) prob[c] *= inv_temp;
        m = fmaxf(m, prob[c]);
    }
    m = wave_max(m);
    if ((tid & 63) == 63) rv[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) sh_max = fmaxf(fmaxf(rv[0], rv[1]), fmaxf(rv[2], rv[3]));
    __syncthreads();
    for (int c = tid; c < V; c += 256) prob[c] = expf(prob[c] - sh_max);
    __syncthreads();
    if (tid == 0) {
        float z = 0.f;
        for (int c = 0; c < V; ++c) z += prob[c];
        sh_z = z;
    }
    __syncthreads();
    for (int c = tid; c < V; c += 256) prob[c] = prob[c] / sh_z;
    __syncthreads();
    if (top_p < 1.0f) {
        unsigned lo = 0u, hi = __float_as_uint(1.0f / sh_z);      // bit patterns of non-negative floats are ordered like the values; the largest p = 1 / z
        // largest t in [lo, hi) with f(t) = sum{p_j : bits(p_j) <= t} <= 1 - top_p; f(0) = 0 always qualifies (p = +0 entries)
        while (lo + 1 < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            float s = 0.f;
            for (int c = tid; c < V; c += 256) s += (__float_as_uint(prob[c]) <= mid) ? prob[c] : 0.f;
            s = block_sum256(s, rv);
            if (s <= 1.0f - top_p) lo = mid; else hi = mid;
            __syncthreads();
        }
        for (int c = tid; c < V; c += 256)
            if (__float_as_uint(prob[c]) <= lo) prob[c] = 0.f;      // (the largest probability has the pattern hi's upper end: never dropped)
        __syncthreads();
    }
    if (tid == 0) {
        const int alive = unfinished[b];
        float z = 0.f;
        for (int c = 0; c < V; ++c) z += prob[c];
        const float u = uniforms[b];
        float cs = 0.f;
        int next = -1, last = pad;
        for (int c = 0; c < V; ++c) {
            if (prob[c] > 0.f) last = c;
            cs += prob[c] / z;
            if (cs > u) { next = c; break; }
        }
        if (next < 0) next = last;      // (rounding left the running sum at or below u: the last candidate)
        if (!alive) next = pad;
        seq[step + 1] = next;
        chosen = next;
        if (alive && next == eos) unfinished[b] = 0;
        if (alive && next != eos) atomicOr(any_unfinished + step, 1);
    }
    if (xnext) {      // the next decode step's input row, as lm_sample_kernel writes it
        __syncthreads();
        long long t = chosen;
        t = (t < 0) ? 0 : (t >= V ? V - 1 : t);
        float ev[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            ev[i] = (c < H) ? word[t * H + c] + type[c] : 0.f;
            s += ev[i];
        }
        const float mean = block_sum256(s, rv) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dd = (tid + 256 * i < H) ? ev[i] - mean : 0.f; q += dd * dd; }
        const float rstd = 1.0f / sqrtf(block_sum256(q, rv) / (float)H + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (c < H) xnext[(long long)b * H + c] = (ev[i] - mean) * rstd * eg[c] + eb[c];
        }
    }
}

}  // namespace lds

using namespace lds;

// ================================================================================================
// host side
// ================================================================================================
#define lm_fail lds::set_error

#define LM_HIP(expr)                                                                                                    \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) return lm_fail(LDS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct LmLinear { float* wt = nullptr; float* b = nullptr; int K = 0, M = 0; };
struct LmLN { float* g = nullptr; float* b = nullptr; };
struct LmAttn { LmLinear qkv, q, kv, o; LmLN ln; };      // self: qkv fused; cross: q and kv separately
struct LmLayer { LmAttn self, cross; LmLinear ff1, ff2; LmLN ln_ff; bool has_cross = false; };
struct LmStack { float *word = nullptr, *type = nullptr, *table = nullptr; LmLN ln_emb; std::vector<LmLayer> layers; };

struct lds_lm {
    lds_lm_cfg cfg;
    std::vector<void*> bufs;
    LmStack enc, dec;
    float* spk = nullptr;
    LmLinear head_t, head_d;
    LmLN head_ln;
    ~lds_lm() { for (void* p : bufs) (void)hipFree(p); }
    float* up(const std::vector<float>& h) {
        void* d = nullptr;
        if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        bufs.push_back(d);
        return (float*)d;
    }
};

namespace {
struct LmTensors {
    std::map<std::string, std::pair<const float*, int64_t>> m;
    std::string missing;
    const float* get(const std::string& k, int64_t n) {
        auto it = m.find(k);
        if (it == m.end() || it->second.second != n) { if (missing.empty()) missing = k; return nullptr; }
        return it->second.first;
    }
};
// reference layout W [M][K] -> transposed [K][M] (several matrices may be concatenated along M)
bool lm_linear(lds_lm* lm, LmTensors& T, const std::vector<std::string>& prefixes, int K, int Mper, LmLinear& out) {
    const int M = Mper * (int)prefixes.size();
    std::vector<float> wt((size_t)K * M), b(M);
    for (size_t s = 0; s < prefixes.size(); ++s) {
        const float* w = T.get(prefixes[s] + "weight", (int64_t)Mper * K);
        const float* bb = T.get(prefixes[s] + "bias", Mper);
        if (!w || !bb) return false;
        for (int m = 0; m < Mper; ++m) {
            for (int k = 0; k < K; ++k) wt[(size_t)k * M + s * Mper + m] = w[(size_t)m * K + k];
            b[s * Mper + m] = bb[m];
        }
    }
    out.wt = lm->up(wt); out.b = lm->up(b); out.K = K; out.M = M;
    return out.wt && out.b;
}
bool lm_vec(lds_lm* lm, LmTensors& T, const std::string& k, int64_t n, float*& out) {
    const float* p = T.get(k, n);
    if (!p) return false;
    out = lm->up(std::vector<float>(p, p + n));
    return out != nullptr;
}
bool lm_ln(lds_lm* lm, LmTensors& T, const std::string& p, int H, LmLN& out) { return lm_vec(lm, T, p + "weight", H, out.g) && lm_vec(lm, T, p + "bias", H, out.b); }
bool lm_attn(lds_lm* lm, LmTensors& T, const std::string& p, int H, bool cross, LmAttn& a) {
    bool ok = true;
    if (cross) ok = lm_linear(lm, T, {p + "self.query."}, H, H, a.q) && lm_linear(lm, T, {p + "self.key.", p + "self.value."}, H, H, a.kv);
    else ok = lm_linear(lm, T, {p + "self.query.", p + "self.key.", p + "self.value."}, H, H, a.qkv);
    return ok && lm_linear(lm, T, {p + "output.dense."}, H, H, a.o) && lm_ln(lm, T, p + "output.LayerNorm.", H, a.ln);
}
bool lm_stack(lds_lm* lm, LmTensors& T, const std::string& p, int vocab, int type_vocab, int n_layers, bool cross, LmStack& s) {
    const lds_lm_cfg& c = lm->cfg;
    const int H = c.hidden, d = H / c.heads;
    bool ok = lm_vec(lm, T, p + "embeddings.word_embeddings.weight", (int64_t)vocab * H, s.word) &&
              lm_vec(lm, T, p + "embeddings.token_type_embeddings.weight", (int64_t)type_vocab * H, s.type) &&
              lm_ln(lm, T, p + "embeddings.LayerNorm.", H, s.ln_emb) && lm_vec(lm, T, p + "encoder.embed_positions.weight", (int64_t)c.max_pos * d, s.table);
    for (int i = 0; i < n_layers && ok; ++i) {
        LmLayer L;
        const std::string q = p + "encoder.layer." + std::to_string(i) + ".";
        L.has_cross = cross;
        ok = lm_attn(lm, T, q + "attention.", H, false, L.self) && (!cross || lm_attn(lm, T, q + "crossattention.", H, true, L.cross)) &&
             lm_linear(lm, T, {q + "intermediate.dense."}, H, c.inter, L.ff1) && lm_linear(lm, T, {q + "output.dense."}, c.inter, H, L.ff2) &&
             lm_ln(lm, T, q + "output.LayerNorm.", H, L.ln_ff);
        s.layers.push_back(L);
    }
    return ok;
}
}  // namespace

extern "C" int lds_lm_create(const lds_lm_cfg* cfg, int n, const char* const* names, const float* const* ptrs, const int64_t* numel, lds_lm** out) {
    if (!cfg || !names || !ptrs || !numel || !out) return lm_fail(LDS_EINVAL, "null argument");
    if (cfg->hidden != 256 || cfg->heads <= 0 || cfg->hidden % cfg->heads || cfg->hidden / cfg->heads > 32 || (cfg->hidden / cfg->heads) % 2 ||
        cfg->sem_vocab > 256 * 32 || cfg->inter <= 0 || cfg->inter % 256 || cfg->inter > 3840)      // (inter: the staged rows of ff2 must fit in LDS)
        return lm_fail(LDS_EINVAL, "unsupported LM shape (hidden must be 256, head dim even and <= 32, vocabulary <= 8192, intermediate size a multiple of 256 <= 3840)");
    LmTensors T;
    for (int i = 0; i < n; ++i) T.m[names[i]] = {ptrs[i], numel[i]};
    lds_lm* lm = new lds_lm();
    lm->cfg = *cfg;
    const int H = cfg->hidden;
    bool ok = lm_stack(lm, T, "text_encoder.", cfg->text_vocab, cfg->type_vocab, cfg->enc_layers, false, lm->enc) &&
              lm_stack(lm, T, "semantic_decoder.roformer.", cfg->sem_vocab, 1, cfg->dec_layers, true, lm->dec);
    const std::string c = "semantic_decoder.cls.predictions.";
    ok = ok && lm_linear(lm, T, {c + "transform.dense."}, H, H, lm->head_t) && lm_ln(lm, T, c + "transform.LayerNorm.", H, lm->head_ln) &&
         lm_linear(lm, T, {c + "decoder."}, H, cfg->sem_vocab, lm->head_d);
    if (ok && cfg->n_spk_rows > 0) ok = lm_vec(lm, T, "spk_emb.weight", (int64_t)cfg->n_spk_rows * H, lm->spk);
    if (!ok) {
        const std::string miss = T.missing;
        delete lm;
        if (!miss.empty()) return lm_fail(LDS_EMISSING, "LM weight tensor %s (absent or wrong size)", miss.c_str());
        return lm_fail(LDS_ENOMEM, "LM weight upload failed");
    }
    *out = lm;
    return LDS_OK;
}
extern "C" void lds_lm_destroy(lds_lm* lm) { delete lm; }

namespace {
struct LmArena {
    char* base; size_t cap, used = 0; bool ok = true;
    LmArena(void* p, size_t n) : base((char*)p), cap(n) {}
    float* f(size_t n) {
        const size_t bytes = (n * sizeof(float) + 255) & ~(size_t)255;
        if (base && used + bytes > cap) ok = false;
        char* p = base ? base + used : nullptr;
        used += bytes;
        return (float*)p;
    }
};
struct LmWs { float *x, *y, *z, *qkv, *ctx, *ff, *kv, *logits, *kc_tmp, *vc_tmp; int* flags; std::vector<float*> kc, vc, ckc, cvc; };
// N = rows processed at once (B * L for the prefill, B for a decode step)
void lm_plan(const lds_lm* lm, LmArena& A, int B, int L, int cap, LmWs& w) {
    const lds_lm_cfg& c = lm->cfg;
    const size_t N = (size_t)B * (L > 1 ? L : 1), H = c.hidden;
    w.x = A.f(N * H); w.y = A.f(N * H); w.z = A.f(N * H); w.qkv = A.f(N * 3 * H); w.ctx = A.f(N * H); w.ff = A.f(N * c.inter); w.kv = A.f(N * 2 * H);
    w.logits = A.f((size_t)B * c.sem_vocab);
    w.kc_tmp = A.f(N * H); w.vc_tmp = A.f(N * H);
    w.flags = (int*)A.f((size_t)cap + B + 64);
    w.kc.clear(); w.vc.clear(); w.ckc.clear(); w.cvc.clear();
    for (int i = 0; i < c.dec_layers; ++i) {
        w.kc.push_back(A.f((size_t)B * H * cap)); w.vc.push_back(A.f((size_t)B * H * cap));
        w.ckc.push_back(A.f((size_t)B * H * L)); w.cvc.push_back(A.f((size_t)B * H * L));
    }
}
hipError_t lm_attention(const float* q, int ldq, const float* kc, const float* vc, int cap, int B, int Lq, int Lk, const int* klen, const lds_lm_cfg& c, float* out,
                        hipStream_t st) {
    const int total = B * Lq * c.heads;
    const size_t lds = (size_t)(((Lk + 63) & ~63) + 4 * 34) * sizeof(float);
    hipLaunchKernelGGL(lm_attn_kernel, dim3(total), dim3(256), lds, st, q, ldq, kc, vc, cap, Lq, Lk, klen, c.hidden, c.heads, out);
    return hipGetLastError();
}
// rows [N][hidden] together with the LayerNorm that is still owed to them (ln == nullptr: already normalised)
struct LmRows { float* p; const LmLN* ln; };

template <int EPI>
hipError_t lm_dln(const LmLinear& W, const LmRows& X, int ldx, const LmRows* R, float eps, float* Y, int ldy, int N, hipStream_t st, const LmRope* rope = nullptr) {
    if (W.K % 256 || ((EPI == 4 || EPI == 5) && W.M % 64)) return hipErrorInvalidValue;
    if (EPI == 5 && !R) return hipErrorInvalidValue;
    if (EPI == 4 && (!rope || W.M % 192 || (W.M / 3 / rope->heads) % 2)) return hipErrorInvalidValue;
    auto kern = lm_linear_dln_kernel<EPI>;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t e = ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), attr_done);
    if (e != hipSuccess) return e;
    const size_t lds = ((size_t)W.K * 8 + (size_t)15 * 64 * 8 + (EPI == 5 ? (size_t)W.M * 8 : 0) + 16 * 8) * sizeof(float);
    const LmRope ro = rope ? *rope : LmRope{nullptr, nullptr, nullptr, 0, 1, 1, 0};
    hipLaunchKernelGGL(kern, dim3((W.M + 63) / 64, (N + 7) / 8), dim3(1024), lds, st, (const float*)X.p, ldx, X.ln ? X.ln->g : nullptr, X.ln ? X.ln->b : nullptr, W.wt, W.b,
                       R ? (const float*)R->p : nullptr, (R && R->ln) ? R->ln->g : nullptr, (R && R->ln) ? R->ln->b : nullptr, eps, Y, ldy, N, W.K, W.M, ro);
    return hipGetLastError();
}

// one BERT-style post-LN layer over N = B * L rows; self-attention over [pos0, pos0 + L) appended to the cache (kc, vc).
// Every LayerNorm is deferred to the readers of its rows (lm_linear_dln_kernel): `x` comes in, and goes out, as rows + owed LayerNorm.
// The three row buffers of the workspace rotate: input -> a (attention block) -> c (cross-attention block) -> output in the input's buffer.
// self_klen / cross_klen: per-batch-row key counts of the padding mask (device int32 [B]) or null
int lm_layer(const lds_lm* lm, const LmStack& s, const LmLayer& Ly, const LmWs& w, LmRows& x, int B, int L, int pos0, float* kc, float* vc, int cap,
             const float* ckc, const float* cvc, int Lenc, const int* self_klen, const int* cross_klen, hipStream_t st) {
    const lds_lm_cfg& c = lm->cfg;
    const int H = c.hidden, N = B * L;
    if (Ly.self.o.M != H || Ly.ff2.M != H || (Ly.has_cross && (Ly.cross.o.M != H || Ly.cross.q.K != H))) return lm_fail(LDS_EINVAL, "layer shapes");
    float* free1 = (x.p == w.x) ? w.y : w.x;
    float* free2 = (x.p == w.z) ? w.y : w.z;
    if (free1 == free2) free2 = w.x;
    {
        const LmRope rope{s.table, kc, vc, pos0, L, c.heads, cap};      // rotary embedding and cache append in the projection's epilogue
        LM_HIP(lm_dln<4>(Ly.self.qkv, x, H, nullptr, c.eps, w.qkv, 3 * H, N, st, &rope));
    }
    LM_HIP(lm_attention(w.qkv, 3 * H, kc, vc, cap, B, L, pos0 + L, self_klen, c, w.ctx, st));
    const LmRows ctx{w.ctx, nullptr};
    LM_HIP(lm_dln<5>(Ly.self.o, ctx, H, &x, c.eps, free1, H, N, st));               // a = W_o ctx + LN(x)
    LmRows cur{free1, &Ly.self.ln};
    if (Ly.has_cross) {
        LM_HIP(lm_dln<0>(Ly.cross.q, cur, H, nullptr, c.eps, w.qkv, H, N, st));
        LM_HIP(lm_attention(w.qkv, H, ckc, cvc, Lenc, B, L, Lenc, cross_klen, c, w.ctx, st));
        LM_HIP(lm_dln<5>(Ly.cross.o, ctx, H, &cur, c.eps, free2, H, N, st));        // c = W_o' ctx' + LN(a)
        cur = LmRows{free2, &Ly.cross.ln};
    }
    LM_HIP(lm_dln<1>(Ly.ff1, cur, H, nullptr, c.eps, w.ff, c.inter, N, st));
    const LmRows ff{w.ff, nullptr};
    LM_HIP(lm_dln<5>(Ly.ff2, ff, c.inter, &cur, c.eps, x.p, H, N, st));              // g = W_2 f + LN(cur), into the input's buffer (no longer read)
    x.ln = &Ly.ln_ff;
    return LDS_OK;
}
}  // namespace

extern "C" int lds_lm_workspace_bytes(const lds_lm* lm, int B, int L, int max_length, size_t* out) {
    if (!lm || !out || B <= 0 || L <= 0 || max_length < 2) return lm_fail(LDS_EINVAL, "bad argument");
    LmArena A(nullptr, 0);
    LmWs w;
    lm_plan(lm, A, B, L, max_length, w);
    *out = A.used;
    return LDS_OK;
}

// phone, tone [B,L] int64 (dev), spk_id [B,L] int64 or NULL, enc_len [B] int32 (dev) or NULL = the padding mask's key counts -> enc [B,L,hidden] (dev)
extern "C" int lds_lm_encode(lds_lm* lm, const int64_t* phone, const int64_t* tone, const int64_t* spk_id, const int32_t* enc_len, float* enc, void* ws,
                             size_t ws_bytes, int B, int L, void* stream) {
    if (!lm || !phone || !tone || !enc || !ws || B <= 0 || L <= 0) return lm_fail(LDS_EINVAL, "bad argument");
    const lds_lm_cfg& c = lm->cfg;
    if (L > c.max_pos) return lm_fail(LDS_EINVAL, "sequence longer than max_position_embeddings");
    hipStream_t st = (hipStream_t)stream;
    LmArena A(ws, ws_bytes);
    LmWs w;
    lm_plan(lm, A, B, L, 2, w);
    if (!A.ok) return lm_fail(LDS_ENOMEM, "LM workspace too small: need %zu bytes", A.used);
    const int N = B * L, H = c.hidden;
    hipLaunchKernelGGL(lm_embed_kernel, dim3(N), dim3(256), 0, st, lm->enc.word, lm->enc.type, lm->spk, lm->enc.ln_emb.g, lm->enc.ln_emb.b, phone, 1, tone,
                       lm->spk ? spk_id : nullptr, 1, c.eps, H, c.text_vocab, c.type_vocab, c.n_spk_rows > 0 ? c.n_spk_rows : 1, w.x);
    LM_HIP(hipGetLastError());
    LmRows cx{w.x, nullptr};
    for (const LmLayer& Ly : lm->enc.layers) {
        // the encoder's "cache" is just this layer's keys / values for all L positions
        int r = lm_layer(lm, lm->enc, Ly, w, cx, B, L, 0, w.kc_tmp, w.vc_tmp, L, nullptr, nullptr, 0, enc_len, nullptr, st);
        if (r != LDS_OK) return r;
    }
    if (cx.ln) {      // the states leave the library normalised
        hipLaunchKernelGGL(lm_ln_rows_kernel, dim3(N), dim3(256), 0, st, (const float*)cx.p, cx.ln->g, cx.ln->b, c.eps, H, enc);
        LM_HIP(hipGetLastError());
    } else {
        LM_HIP(hipMemcpyAsync(enc, cx.p, sizeof(float) * N * H, hipMemcpyDeviceToDevice, st));
    }
    return LDS_OK;
}

// enc [B,L,hidden] (dev); uniforms [max_length-1][B] (dev, sampling only); tokens [B][max_length] int64 (dev): BOS then the generated ids,
// positions past the returned length are unspecified; logits_out optional [max_length-1][B][vocab] (dev).  *n_tokens_host = sequence length incl. BOS.
extern "C" int lds_lm_generate(lds_lm* lm, const float* enc, const int32_t* enc_len, int B, int L, int max_length, int do_sample, int top_k, float top_p,
                               float temperature, float repetition_penalty, const float* uniforms, int64_t* tokens, float* logits_out, int* n_tokens_host,
                               void* ws, size_t ws_bytes, void* stream) {
    if (!lm || !enc || !tokens || !n_tokens_host || !ws || B <= 0 || L <= 0 || max_length < 2) return lm_fail(LDS_EINVAL, "bad argument");
    const lds_lm_cfg& c = lm->cfg;
    if (do_sample && (!uniforms || top_k < 0 || top_k > kMaxTopK || top_k > c.sem_vocab || !(top_p > 0.f) || !(temperature > 0.f)))
        return lm_fail(LDS_EINVAL, "sampling needs uniforms, 0 <= top_k <= %d (0 = no top-k filter), top_p > 0, temperature > 0", kMaxTopK);
    if (do_sample && top_k == 0 && (size_t)c.sem_vocab * sizeof(float) > 60 * 1024) return lm_fail(LDS_EINVAL, "top_k = 0 needs the vocabulary row in LDS (<= 15360 entries)");
    if (max_length > c.max_pos) return lm_fail(LDS_EINVAL, "max_length exceeds max_position_embeddings");
    // the encoder states come from lds_lm_encode (L <= max_pos); the cross-attention stages one score per encoder position in LDS
    if (L < 1 || L > c.max_pos || (size_t)(((L + 63) & ~63) + 4 * 34) * sizeof(float) > 64 * 1024)
        return lm_fail(LDS_EINVAL, "encoder length %d out of range (1 .. min(max_position_embeddings = %d, 16000))", L, c.max_pos);
    hipStream_t st = (hipStream_t)stream;
    LmArena A(ws, ws_bytes);
    LmWs w;
    lm_plan(lm, A, B, L, max_length, w);
    if (!A.ok) return lm_fail(LDS_ENOMEM, "LM workspace too small: need %zu bytes", A.used);
    const int H = c.hidden, V = c.sem_vocab;
    int* unfinished = w.flags;                 // [B]
    int* any_unf = w.flags + B;                // [max_length]: 1 when a sequence is still running after step s
    // cross-attention keys / values of every decoder layer, once
    for (int i = 0; i < c.dec_layers; ++i) {
        {
            const LmRows er{const_cast<float*>(enc), nullptr};
            LM_HIP(lm_dln<0>(lm->dec.layers[i].cross.kv, er, H, nullptr, c.eps, w.kv, 2 * H, B * L, st));
        }
        hipLaunchKernelGGL(lm_kv_pack_kernel, dim3((unsigned)(((long long)B * L * H + 255) / 256)), dim3(256), 0, st, w.kv, B, L, H, c.heads, w.ckc[i], w.cvc[i], L);
        LM_HIP(hipGetLastError());
    }
    {
        std::vector<int> init(B + max_length, 0);
        for (int b = 0; b < B; ++b) init[b] = 1;
        LM_HIP(hipMemcpyAsync(unfinished, init.data(), sizeof(int) * (B + max_length), hipMemcpyHostToDevice, st));
        std::vector<int64_t> bos((size_t)B * max_length, (int64_t)c.sem_pad);
        for (int b = 0; b < B; ++b) bos[(size_t)b * max_length] = c.sem_bos;
        LM_HIP(hipMemcpyAsync(tokens, bos.data(), sizeof(int64_t) * B * max_length, hipMemcpyHostToDevice, st));
        LM_HIP(hipStreamSynchronize(st));      // the two host staging vectors go out of scope
    }
    const float inv_temp = do_sample ? 1.0f / temperature : 1.0f;
    int n_tokens = max_length;
    std::vector<int> host_flags(max_length, 0);
    int checked = 0;
    for (int step = 0; step + 1 < max_length; ++step) {
        // token at position `step` -> logits -> token at position step + 1
        if (step == 0) {      // the BOS embedding; every later step's input row is written by the previous step's token-choice kernel
            hipLaunchKernelGGL(lm_embed_kernel, dim3(B), dim3(256), 0, st, lm->dec.word, lm->dec.type, (const float*)nullptr, lm->dec.ln_emb.g, lm->dec.ln_emb.b,
                               (const int64_t*)tokens, max_length, (const int64_t*)nullptr, (const int64_t*)nullptr, 0, c.eps, H, c.sem_vocab, 1, 1, w.x);
            LM_HIP(hipGetLastError());
        }
        LmRows cx{w.x, nullptr};
        for (int i = 0; i < c.dec_layers; ++i) {
            int r = lm_layer(lm, lm->dec, lm->dec.layers[i], w, cx, B, 1, step, w.kc[i], w.vc[i], max_length, w.ckc[i], w.cvc[i], L, nullptr, enc_len, st);
            if (r != LDS_OK) return r;
        }
        // LM head: h = GELU(W_t LN(x)) stored un-normalised, logits = W_d LN(h)
        LM_HIP(lm_dln<1>(lm->head_t, cx, H, nullptr, c.eps, w.ctx, H, B, st));
        const LmRows hrows{w.ctx, &lm->head_ln};
        float* lg = logits_out ? logits_out + (size_t)step * B * V : w.logits;
        LM_HIP(lm_dln<0>(lm->head_d, hrows, H, nullptr, c.eps, lg, V, B, st));
        if (do_sample && top_k == 0)
            hipLaunchKernelGGL(lm_sample_full_kernel, dim3(B), dim3(256), sizeof(float) * V, st, lg, V, top_p, inv_temp, repetition_penalty, uniforms + (size_t)step * B,
                               tokens, max_length, step, unfinished, c.sem_eos, c.sem_pad, any_unf, lm->dec.word, lm->dec.type, lm->dec.ln_emb.g, lm->dec.ln_emb.b,
                               c.eps, H, w.x);
        else if (V <= 256 * 9)
            hipLaunchKernelGGL(lm_sample_kernel<9>, dim3(B), dim3(256), 0, st, lg, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty,
                               do_sample ? uniforms + (size_t)step * B : nullptr, tokens, max_length, step, unfinished, c.sem_eos, c.sem_pad, any_unf,
                               lm->dec.word, lm->dec.type, lm->dec.ln_emb.g, lm->dec.ln_emb.b, c.eps, H, w.x);
        else if (V <= 256 * 17)      // (the reference's 4096-entry semantic codebook + 3 special ids)
            hipLaunchKernelGGL(lm_sample_kernel<17>, dim3(B), dim3(256), 0, st, lg, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty,
                               do_sample ? uniforms + (size_t)step * B : nullptr, tokens, max_length, step, unfinished, c.sem_eos, c.sem_pad, any_unf,
                               lm->dec.word, lm->dec.type, lm->dec.ln_emb.g, lm->dec.ln_emb.b, c.eps, H, w.x);
        else
            hipLaunchKernelGGL(lm_sample_kernel<32>, dim3(B), dim3(256), 0, st, lg, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty,
                               do_sample ? uniforms + (size_t)step * B : nullptr, tokens, max_length, step, unfinished, c.sem_eos, c.sem_pad, any_unf,
                               lm->dec.word, lm->dec.type, lm->dec.ln_emb.g, lm->dec.ln_emb.b, c.eps, H, w.x);
        LM_HIP(hipGetLastError());
        // EOS poll every 8 steps: the loop ends after the step in which the last running sequence emitted EOS
        if ((step & 7) == 7 || step + 2 == max_length) {
            LM_HIP(hipMemcpyAsync(host_flags.data() + checked, any_unf + checked, sizeof(int) * (step + 1 - checked), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            bool done = false;
            for (int s = checked; s <= step; ++s)
                if (!host_flags[s]) { n_tokens = s + 2; done = true; break; }
            checked = step + 1;
            if (done) break;
        }
    }
    *n_tokens_host = n_tokens;
    return LDS_OK;
}

// Test entry (include/lds_test.h): ONE token choice per row of `logits` [B][V] with the generate loop's own kernels, after a history of n_hist tokens
// per row (hist [B][n_hist] int64, dev; what the repetition penalty looks at).  out [B] int64 (dev).  top_k 0 = no top-k filter.
extern "C" int lds_test_lm_sample(const float* logits, int B, int V, int do_sample, int top_k, float top_p, float temperature, float repetition_penalty,
                                  const float* uniforms, const int64_t* hist, int n_hist, int64_t* out, void* stream) {
    if (!logits || !out || B <= 0 || V <= 0 || V > 256 * 32 || n_hist < 1 || !hist || (do_sample && !uniforms) || top_k < 0 || top_k > kMaxTopK) return lm_fail(LDS_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int cap = n_hist + 1;
    int64_t* seq = nullptr;
    int* flags = nullptr;
    if (hipMalloc(&seq, sizeof(int64_t) * B * cap) != hipSuccess || hipMalloc(&flags, sizeof(int) * (B + cap)) != hipSuccess) return lm_fail(LDS_ENOMEM, "alloc");
    std::vector<int> init(B + cap, 0);
    for (int b = 0; b < B; ++b) init[b] = 1;
    int rc = LDS_OK;
    do {
        if (hipMemcpy(flags, init.data(), sizeof(int) * (B + cap), hipMemcpyHostToDevice) != hipSuccess) { rc = lm_fail(LDS_EHIP, "copy"); break; }
        if (hipMemcpy2DAsync(seq, sizeof(int64_t) * cap, hist, sizeof(int64_t) * n_hist, sizeof(int64_t) * n_hist, B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            rc = lm_fail(LDS_EHIP, "copy"); break;
        }
        const float inv_temp = do_sample ? 1.0f / temperature : 1.0f;
        const int step = n_hist - 1;
        const float* nf = nullptr;
        if (do_sample && top_k == 0)
            hipLaunchKernelGGL(lm_sample_full_kernel, dim3(B), dim3(256), sizeof(float) * V, st, logits, V, top_p, inv_temp, repetition_penalty, uniforms, seq, cap, step,
                               flags, -1, -1, flags + B, nf, nf, nf, nf, 0.f, 0, (float*)nullptr);
        else if (V <= 256 * 9)
            hipLaunchKernelGGL(lm_sample_kernel<9>, dim3(B), dim3(256), 0, st, logits, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty, uniforms, seq, cap,
                               step, flags, -1, -1, flags + B, nf, nf, nf, nf, 0.f, 0, (float*)nullptr);
        else if (V <= 256 * 17)
            hipLaunchKernelGGL(lm_sample_kernel<17>, dim3(B), dim3(256), 0, st, logits, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty, uniforms, seq, cap,
                               step, flags, -1, -1, flags + B, nf, nf, nf, nf, 0.f, 0, (float*)nullptr);
        else
            hipLaunchKernelGGL(lm_sample_kernel<32>, dim3(B), dim3(256), 0, st, logits, V, do_sample, do_sample ? top_k : 1, top_p, inv_temp, repetition_penalty, uniforms, seq, cap,
                               step, flags, -1, -1, flags + B, nf, nf, nf, nf, 0.f, 0, (float*)nullptr);
        if (hipGetLastError() != hipSuccess) { rc = lm_fail(LDS_EHIP, "launch"); break; }
        if (hipMemcpy2DAsync(out, sizeof(int64_t), seq + n_hist, sizeof(int64_t) * cap, sizeof(int64_t), B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            rc = lm_fail(LDS_EHIP, "copy"); break;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = lm_fail(LDS_EHIP, "sync");
    } while (0);
    (void)hipFree(seq);
    (void)hipFree(flags);
    return rc;
}
