// Self-attention for the K4P UNet path (reference Attention + AttnProcessor2_0, attention_processor.py:980-1052).
// Same algorithm as attention.hip (swapped QK^T so each lane owns one query column, probabilities fed from the
// accumulator registers straight into the P.V MFMAs), but all operands are vector LDS reads:
//   q, k arrive in K4P (k4p.h): the 4 floats at one (row, frame) are the head-dim values 8q+2j+h, j=0..3, i.e.
//   the A (keys) / B (queries) operands of four consecutive v_mfma_f32_32x32x2_f32 -> one ds_read_b128 per 4 MFMAs;
//   v arrives frame-major [d][frames] (the QKV conv stores that third plain), staged with pitch 68 so that one
//   conflict-free ds_read_b128 = V[d][4 consecutive keys] = the A operands of the four P.V MFMAs that consume
//   accumulator registers 4g..4g+3.
// Output is written in K4P (two 8-byte stores per 8-channel block, pad frames included).
#include "k4p.h"
#include "kernels.h"

#include <math.h>

namespace lds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int D, int NW>
__global__ void __launch_bounds__(NW * 64) attention_k4p_kernel(const float* __restrict__ qk, const float* __restrict__ vp, float* __restrict__ out,
                                                                int C, int T, float scale2) {
    constexpr int DQ = D / 8;              // 8-channel blocks per head
    constexpr int DT = (D + 31) / 32;
    constexpr int KB = 64, VP = 68;
    constexpr int KSZ = DQ * 2 * KB * 4, VSZ = DT * 32 * VP;
    constexpr int NKL = DQ * 2 * KB / (NW * 64), NVL = D * (KB / 4) / (NW * 64);   // 16-byte entries per thread and tile
    static_assert(NKL * NW * 64 == DQ * 2 * KB && NVL * NW * 64 == D * (KB / 4), "tile must split evenly over the threads");
    __shared__ __attribute__((aligned(16))) float Ks[2 * KSZ];   // two stages of [kq][h][key][4]
    __shared__ __attribute__((aligned(16))) float Vs[2 * VSZ];   // two stages of [d][key], pitch 68
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    // XCD-aware order: the query blocks of one (batch, head) share K and V, so they take consecutive slots of one XCD
    // (workgroups are dispatched round-robin over the 8 XCDs, each with its own L2)
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int id = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const int per = total >> 3, rem = total & 7, xcd = id & 7;
    const int L = xcd * per + (xcd < rem ? xcd : rem) + (id >> 3);
    const int qblk = L % gx, hd = (L / gx) % gy, b = L / (gx * gy);
    const int tq = qblk * (NW * 32) + wave * 32 + c;
    const int Tp = T + 2;
    const float* qb = qk + ((long long)b * 2 * C + (long long)hd * D) * Tp;            // q rows of this head
    const float* kb = qk + ((long long)b * 2 * C + C + (long long)hd * D) * Tp;        // k rows
    const float* vb = vp + ((long long)b * C + (long long)hd * D) * T;

    f32x4 qv[DQ];
#pragma unroll
    for (int kq = 0; kq < DQ; ++kq) {
        qv[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tq < T) qv[kq] = *reinterpret_cast<const f32x4*>(qb + ((long long)(kq * 2 + h) * Tp + tq + 1) * 4);
    }
    for (int i = tid; i < (DT * 32 - D) * VP; i += NW * 64) Vs[D * VP + i] = Vs[VSZ + D * VP + i] = 0.f;   // head-dim padding rows (D = 48)

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bool vec = (T & 3) == 0;

    // software pipeline: the next key tile travels global -> registers while the current one is consumed from LDS
    f32x4 kreg[NKL], vreg[NVL];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NKL; ++i) {                                   // K: straight 16-byte copies of K4P entries
            const int idx = tid + i * NW * 64, row = idx / KB, key = idx - row * KB;
            kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k0 + key < T) kreg[i] = *reinterpret_cast<const f32x4*>(kb + ((long long)row * Tp + k0 + key + 1) * 4);
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {                                   // V: rows of 64 keys
            const int idx = tid + i * NW * 64, d = idx / (KB / 4), j = (idx - d * (KB / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (vec && k0 + j + 3 < T) {
                v = *reinterpret_cast<const f32x4*>(vb + (long long)d * T + k0 + j);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (k0 + j + e < T) ? vb[(long long)d * T + k0 + j + e] : 0.f;
            }
            vreg[i] = v;
        }
    };
    auto stash = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NKL; ++i) {
            const int idx = tid + i * NW * 64;
            *reinterpret_cast<f32x4*>(Ks + stage * KSZ + idx * 4) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int idx = tid + i * NW * 64, d = idx / (KB / 4), j = (idx - d * (KB / 4)) * 4;
            *reinterpret_cast<f32x4*>(Vs + stage * VSZ + d * VP + j) = vreg[i];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();

    int stage = 0;
    for (int k0 = 0; k0 < T; k0 += KB, stage ^= 1) {
        const bool more = k0 + KB < T;
        if (more) fetch(k0 + KB);
        const float* Kc = Ks + stage * KSZ;
        const float* Vc = Vs + stage * VSZ;
#pragma unroll 1
        for (int kt = 0; kt < KB / 32; ++kt) {
            const int kbase = k0 + kt * 32;
            if (kbase >= T) break;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int kq = 0; kq < DQ; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Kc + ((kq * 2 + h) * KB + kt * 32 + c) * 4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj], qv[kq][jj], s, 0, 0, 0);
            }
            // scores are kept in log2 units (scale2 = log2(e)/sqrt(d)) so every exponential is one v_exp_f32; all of this
            // VALU work is paid in matrix time on gfx950 (the fp32 MFMA shares the vector ALU), so it is kept minimal
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                s[r] = (key < T) ? s[r] * scale2 : -INFINITY;
                mt = fmaxf(mt, s[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ls += s[r]; }
            if (__any(m_new != m_run)) {             // wave-uniform: rescale only when some query's running maximum moved
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m_run = m_new;
            }
            l_run += ls;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(Vc + (i * 32 + c) * VP + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], s[4 * g + e], o[i], 0, 0, 0);
                }
            }
        }
        if (more) stash(stage ^ 1);       // that stage was last read one iteration ago, before the barrier below
        __syncthreads();
    }
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    if (tq < T) {
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
                    *reinterpret_cast<f32x2*>(ob + off) = f32x2{o[i][4 * g + hh] / l, o[i][4 * g + 2 + hh] / l};
                    if (tq == 0) *reinterpret_cast<f32x2*>(ob + off - 4) = f32x2{0.f, 0.f};
                    if (tq == T - 1) *reinterpret_cast<f32x2*>(ob + off + 4) = f32x2{0.f, 0.f};
                }
            }
    }
}

template <int D>
static hipError_t launch_dk(const float* qk, const float* v, float* out, int B, int C, int T, int heads, hipStream_t s) {
    const float scale = 1.4426950408889634f / sqrtf((float)D);    // log2(e) / sqrt(d)
    if (T > 64) hipLaunchKernelGGL((attention_k4p_kernel<D, 4>), dim3((T + 127) / 128, heads, B), dim3(256), 0, s, qk, v, out, C, T, scale);
    else hipLaunchKernelGGL((attention_k4p_kernel<D, 2>), dim3((T + 63) / 64, heads, B), dim3(128), 0, s, qk, v, out, C, T, scale);
    return hipGetLastError();
}

hipError_t launch_attention_k4p(const float* qk, const float* v, float* out, int B, int C, int T, int heads, hipStream_t s) {
    if (C % heads) return hipErrorInvalidValue;
    ProfScope ps(s, "attention", 4.0 * B * (double)T * T * C, 4.0 * 4.0 * B * C * T);
    switch (C / heads) {
        case 32: return launch_dk<32>(qk, v, out, B, C, T, heads, s);
        case 48: return launch_dk<48>(qk, v, out, B, C, T, heads, s);
        case 64: return launch_dk<64>(qk, v, out, B, C, T, heads, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lds
