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
    const float nb = dpp_get<CTRL, ROW_MASK, BOUND>(n), mb = dpp_get<CTRL, ROW_MASK, BOUND>(mean), qb = dpp_get<CTRL, ROW_MASK, BOUND>(m2);
    chan(n, mean, m2, nb, mb, qb);
}

// (mean, variance) of group g of batch element b over all T frames, from the partials `gp` of a C-channel tensor; every lane of the
// wave takes part, the result is wave-uniform.  cg16 = 16-channel blocks per group.  Two phases so that a caller can put other work
// between the request and the use: gn_part_load = this lane's partial of round p0 (an empty one beyond the group's last),
// gn_group_finish = the combination (further rounds are loaded there; P <= 64 partials -- every level of the UNet -- need none).
struct GnPart { float n, mean, m2; };
// Tv: valid frames of this batch element (ragged batches; = T otherwise): blocks beyond it are empty partials whatever the buffer holds
static __device__ __forceinline__ GnPart gn_part_load(const float2* __restrict__ gp, int b, int C, int T, int cg16, int g, int lane, int p0, int Tv) {
    const int nT = (T + 31) >> 5, P = cg16 * nT;
    const int pi = p0 + lane;
    GnPart r{0.f, 0.f, 0.f};
    if (pi < P) {
        const int kk = (int)(((float)pi + 0.5f) * (1.0f / (float)nT)), tb = pi - kk * nT, kb = g * cg16 + kk;      // pi / nT without the integer-division sequence
        const int nv = (Tv - tb * 32 < 32) ? Tv - tb * 32 : 32;
        if (nv > 0) {
            const float2 pr = gp[((long long)b * (C >> 4) + kb) * nT + tb];
            r.n = 16.0f * (float)nv; r.mean = pr.x; r.m2 = pr.y;
        }
    }
    return r;
}
static __device__ __forceinline__ void gn_group_finish(const float2* __restrict__ gp, int b, int C, int T, int cg16, int g, int lane, GnPart first, float& mu, float& var, int Tv) {
    const int nT = (T + 31) >> 5, P = cg16 * nT;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    chan(n, mean, m2, first.n, first.mean, first.m2);
    for (int p0 = 64; p0 < P; p0 += 64) {
        const GnPart q = gn_part_load(gp, b, C, T, cg16, g, lane, p0, Tv);
        chan(n, mean, m2, q.n, q.mean, q.m2);
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

}  // namespace lds
