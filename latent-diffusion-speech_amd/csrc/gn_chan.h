// Chan's parallel combination of (count, mean, M2) partials across a wave in a fixed order (DPP row shifts + row broadcasts): the way
// every consumer of the GroupNorm partials [B][C/16][ceil(T/32)] combines them (gn_stream in k4p_ops.hip; the GroupNorm fold of
// conv_dma / conv_bf3), so that all of them see the same statistics bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace lds {

template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
// an empty side (count 0) is the identity
static __device__ __forceinline__ void chan(float& n, float& mean, float& m2, float nb, float mb, float qb) {
    const float nn = n + nb;
    const float r = (nn > 0.f) ? __builtin_amdgcn_rcpf(nn) : 0.f;      // counts are small integers: v_rcp_f32 is within 1 ulp
    const float d = mb - mean;
    mean += d * (nb * r);
    m2 += qb + d * d * (n * nb * r);
    n = nn;
}
template <int CTRL, int ROW_MASK, bool BOUND>
static __device__ __forceinline__ void chan_step(float& n, float& mean, float& m2) {
    // lanes outside ROW_MASK (and row starts with BOUND) receive zeros = an empty partial
    // (the three values are complete, and two wait states have passed, before any lane reads them across the wave: the compiler may not weave
    //  a step's arithmetic into the next step's cross-lane reads -- section 14 of DESIGN.md is about a schedule of this code that was not stable)
    asm volatile("s_nop 1" : "+v"(n), "+v"(mean), "+v"(m2));
    const float nb = dpp_get<CTRL, ROW_MASK, BOUND>(n), mb = dpp_get<CTRL, ROW_MASK, BOUND>(mean), qb = dpp_get<CTRL, ROW_MASK, BOUND>(m2);
    chan(n, mean, m2, nb, mb, qb);
}

// (mean, variance) of group g of batch element b over all T frames, from the partials `gp` of a C-channel tensor; every lane of the
// wave takes part, the result is wave-uniform.  cg16 = 16-channel blocks per group.  Two phases so that a caller can put other work
// between the request and the use: gn_part_load = this lane's partial of round p0 (an empty one beyond the group's last),
// gn_group_finish = the combination (further rounds are loaded there; P <= 64 partials -- every level of the UNet -- need none).
struct GnPart { float n, mean, m2; };
// Tv: valid frames of this batch element (ragged batches; = T otherwise): blocks beyond it are empty partials whatever the buffer holds.
// The load itself is UNCONDITIONAL (an out-of-range lane reads the group's first partial, a valid address) and only the count carries the mask:
// a load inside a divergent branch is waited for at the end of that branch, and the fold's kernels want these requests in flight across the
// issue of their first operand tiles (round 4: the ISA of the folded proj_in showed `s_waitcnt vmcnt(0)` right behind each request).  An
// empty partial is n = 0 with a finite (mean, M2) that every consumer multiplies by its count or masks by it.
static __device__ __forceinline__ GnPart gn_part_load(const float2* __restrict__ gp, int b, int C, int T, int cg16, int g, int lane, int p0, int Tv) {
    const int nT = (T + 31) >> 5, P = cg16 * nT;
    const int pi = p0 + lane;
    const bool in = pi < P;
    const int pc = in ? pi : 0;
    const int kk = (int)(((float)pc + 0.5f) * (1.0f / (float)nT)), tb = pc - kk * nT, kb = g * cg16 + kk;      // pc / nT without the integer-division sequence
    const int nv = (Tv - tb * 32 < 32) ? Tv - tb * 32 : 32;
    const float2 pr = gp[((long long)b * (C >> 4) + kb) * nT + tb];
    GnPart r;
    r.n = (in && nv > 0) ? 16.0f * (float)nv : 0.f;
    r.mean = pr.x; r.m2 = pr.y;
    return r;
}
static __device__ __forceinline__ void gn_group_finish(const float2* __restrict__ gp, int b, int C, int T, int cg16, int g, int lane, GnPart first, float& mu, float& var, int Tv) {
    const int nT = (T + 31) >> 5, P = cg16 * nT;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    chan(n, mean, m2, first.n, first.mean, (first.n > 0.f) ? first.m2 : 0.f);      // (an empty partial's M2 is whatever its clamped load returned)
    for (int p0 = 64; p0 < P; p0 += 64) {
        const GnPart q = gn_part_load(gp, b, C, T, cg16, g, lane, p0, Tv);
        chan(n, mean, m2, q.n, q.mean, (q.n > 0.f) ? q.m2 : 0.f);
    }
    chan_step<0x111, 0xf, true>(n, mean, m2);
    chan_step<0x112, 0xf, true>(n, mean, m2);
    chan_step<0x114, 0xf, true>(n, mean, m2);
    chan_step<0x118, 0xf, true>(n, mean, m2);
    chan_step<0x142, 0xa, false>(n, mean, m2);
    chan_step<0x143, 0xc, false>(n, mean, m2);
    mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mean), 63));
    var = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m2), 63)) /
          __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n), 63));
}

// ---- the GroupNorm FOLD's statistics (conv_dma.hip / conv_bf3.hip gnf_prepare) -----------------------------------------------------------
// The same (mean, variance) from the same partials, combined as two plain wave sums instead of a Chan tree:
//     mean = sum_i n_i mean_i / N,      var = sum_i (M2_i + n_i (mean_i - mean)^2) / N,      N = 16 cg16 Tv  (known without a reduction)
// Every cross-lane step is ONE add of a value that was complete before the step (no reciprocal, no compare / select, no packed arithmetic
// between the steps), each behind an explicit pad of wait states.  Why: round 4 found the Chan tree's result inside conv_bf3<32,64,..,BK 64,
// F16, GNF> -- and only there, and only in workgroups that share their CU with another one -- differing between launches on identical inputs
// (the lane-63 chain took one step with a zero weight; every recompilation that moved the code hid it).  The order of
// the additions is fixed, so the result is a function of the partials alone.
static __device__ __forceinline__ float gnf_wave_sum(float v) {      // total in every lane's copy of lane 63 -> returned wave-uniform
#define LDS_GNF_STEP(CTRL, RM, BC)                                                                                                        \
    {                                                                                                                                     \
        asm volatile("s_nop 3" : "+v"(v));                                                                                                \
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, BC));                    \
    }
    LDS_GNF_STEP(0x111, 0xf, true)       // row_shr:1
    LDS_GNF_STEP(0x112, 0xf, true)       // row_shr:2
    LDS_GNF_STEP(0x114, 0xf, true)       // row_shr:4
    LDS_GNF_STEP(0x118, 0xf, true)       // row_shr:8    lane 15 of every row = the row's sum
    LDS_GNF_STEP(0x142, 0xa, false)      // row_bcast:15 into rows 1 and 3
    LDS_GNF_STEP(0x143, 0xc, false)      // row_bcast:31 into rows 2 and 3: lane 63 = the wave's sum
#undef LDS_GNF_STEP
    asm volatile("s_nop 3" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// `first` = this lane's partial of round 0 (gn_part_load, requested early); groups of more than 64 partials (T > 1024 at 32 channels per group)
// load the further rounds here, once per pass
static __device__ __forceinline__ void gnf_group_stats(const float2* __restrict__ gp, int b, int C, int T, int cg16, int g, int lane, GnPart first, float& mu,
                                                       float& var, int Tv) {
    const int nT = (T + 31) >> 5, P = cg16 * nT;
    const float rN = __builtin_amdgcn_rcpf(16.0f * (float)cg16 * (float)Tv);      // (1 ulp; the IEEE division is ~25 dependent instructions in front of the first MFMA)
    float s1 = first.n * first.mean;
    for (int p0 = 64; p0 < P; p0 += 64) {
        const GnPart q = gn_part_load(gp, b, C, T, cg16, g, lane, p0, Tv);
        s1 = fmaf(q.n, q.mean, s1);
    }
    mu = gnf_wave_sum(s1) * rN;
    float d = first.mean - mu;
    float s2 = (first.n > 0.f) ? fmaf(first.n * d, d, first.m2) : 0.f;      // (an empty partial carries n = 0 and the finite values of a clamped load)
    for (int p0 = 64; p0 < P; p0 += 64) {
        const GnPart q = gn_part_load(gp, b, C, T, cg16, g, lane, p0, Tv);
        d = q.mean - mu;
        s2 += (q.n > 0.f) ? fmaf(q.n * d, d, q.m2) : 0.f;
    }
    var = gnf_wave_sum(s2) * rN;
}

// LayerNorm statistics of one output column from the producer's np partials (mean, M2) over 32 channels each (conv_dma.hip / conv_bf3.hip
// ln_columns; reference attention.py:83,102,118).  Equal counts make the combination two plain sums in a fixed order,
//   mean = sum(mean_i) / np,   var = (sum(M2_i) + 32 sum((mean_i - mean)^2)) / (32 np),
// ~4 vector instructions per partial instead of the 10 of a running Chan update with its two reciprocals.
// Up to 16 partials (512 channels) stay in registers between the two sums; more are read twice.
static __device__ __forceinline__ void ln_column_stats(const float2* src, long long stride, int np, float eps, bool ok, float& mu, float& rs) {
    constexpr int CH = 16;
    const float cnt = 32.f * (float)np;
    float s1 = 0.f, s2 = 0.f;
    if (np <= CH) {
        float2 pr[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) pr[e] = src[(long long)(e < np ? e : 0) * stride];
#pragma unroll
        for (int e = 0; e < CH; ++e) s1 += (e < np) ? pr[e].x : 0.f;
        mu = s1 * __builtin_amdgcn_rcpf((float)np);      // (v_rcp_f32 / v_rsq_f32, 1 ulp: the IEEE division and square-root sequences are ~25 dependent
                                                         //  instructions each, in front of the launch's first MFMA)
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float d = pr[e].x - mu;
            s2 += (e < np) ? fmaf(32.f * d, d, pr[e].y) : 0.f;
        }
    } else {
        for (int e = 0; e < np; ++e) s1 += src[(long long)e * stride].x;
        mu = s1 * __builtin_amdgcn_rcpf((float)np);
        for (int e = 0; e < np; ++e) {
            const float2 q = src[(long long)e * stride];
            const float d = q.x - mu;
            s2 += fmaf(32.f * d, d, q.y);
        }
    }
    rs = ok ? __builtin_amdgcn_rsqf(fmaf(s2, __builtin_amdgcn_rcpf(cnt), eps)) : 0.f;
}

}  // namespace lds
