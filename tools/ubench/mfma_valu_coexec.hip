// Does v_mfma_f32_32x32x2_f32 share the SIMD's VALU with ordinary vector ops?
// One workgroup per CU, 8 waves: waves 0-3 (one per SIMD) run MFMA chains, waves 4-7 run FMA / transcendental
// chains.  Time each role alone and both together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ void __launch_bounds__(512) k(float* out, int iters, int do_mfma, int valu_kind, int valu_iters) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f32x16 acc[NCH];
            for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
            float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            }
            for (int c = 0; c < NCH; ++c) r += acc[c][0] + acc[c][7];
        }
    } else {
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        if (valu_kind == 1) {
            for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 0.9999f, 0.5f); x3 = fmaf(x3, 0.9999f, 0.25f); }
            }
        } else if (valu_kind == 2) {
            for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { x0 = __builtin_amdgcn_exp2f(x0 * 0.01f); x1 = __builtin_amdgcn_rcpf(x1 + 1.5f); x2 = __builtin_amdgcn_exp2f(x2 * 0.01f); x3 = __builtin_amdgcn_rcpf(x3 + 1.5f); }
            }
        } else if (valu_kind == 3) {   // integer / address-like ops
            int i0 = threadIdx.x, i1 = i0 + 7;
            for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u) { i0 = i0 * 3 + i1; i1 = (i1 << 1) ^ i0; }
            }
            x0 = (float)(i0 + i1);
        }
        r = x0 + x1 + x2 + x3;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int NCH>
float run(float* d, int iters, int do_mfma, int kind, int viters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(512), 0, 0, d, iters, do_mfma, kind, viters);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(512), 0, 0, d, iters, do_mfma, kind, viters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 4000;   // x8 MFMAs per chain
    for (int nch : {1, 2, 4}) {
        auto R = [&](int m, int kind, int vi) { return nch == 1 ? run<1>(d, iters / nch, m, kind, vi) : nch == 2 ? run<2>(d, iters / nch, m, kind, vi) : run<4>(d, iters / nch, m, kind, vi); };
        float tm = R(1, 0, 0);
        double tf = 256.0 * 4 * iters * 8 * (2.0 * 32 * 32 * 2) / (tm * 1e-6) / 1e12;
        printf("chains %d: mfma only %.1f us (%.1f TFLOP/s, %.1f cycles/mfma @2.4GHz)\n", nch, tm, tf, tm * 1e-6 * 2.4e9 / (iters * 8));
        for (int kind : {1, 2, 3}) {
            int vi = 2000;
            float tv = R(0, kind, vi);
            float tb = R(1, kind, vi);
            printf("   valu kind %d: alone %.1f us, with mfma %.1f us (sum %.1f, max %.1f)\n", kind, tv, tb, tm + tv, tm > tv ? tm : tv);
        }
    }
    return 0;
}
