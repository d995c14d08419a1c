// What limits an LDS-DMA-fed fp32 MFMA main loop on MI355X?  4 waves per workgroup (one per SIMD), OCC workgroups per CU.
// Per "K-step" a wave issues NDMA global_load_lds_dwordx4 (1 KB each), NLDS ds_read_b128 and 32 v_mfma_f32_32x32x2_f32
// (two accumulator chains) -- the ratios of conv_dma's 64x64 / BK64 tile when NDMA = 8, NLDS = 16.  Each variant reports
// TFLOP/s and the shader clock during the kernel (s_memtime ticks per s_memrealtime 100 MHz tick).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_feed mfma_feed.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NDMA, int NLDS, bool BAR, bool SPREAD = false>
__global__ void __launch_bounds__(256) feed(const float* __restrict__ src, long long win_floats, int iters, float* out, long long* clk, int share) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 stages x 32 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // share = 1: like a conv tile -- half of the traffic is a window common to all workgroups (weights), half a window
    // shared by 4 workgroups that sit on one XCD (activations)
    const float* base = src + (long long)blockIdx.x * win_floats;
    const float* baseW = share ? src : base;
    const float* baseX = (share ? src + (long long)(1 + (blockIdx.x & 7) + 8 * (blockIdx.x >> 5)) * win_floats : base) + (share >> 1) * 4;   // share & 2: 16-byte misaligned rows
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
    long long off = 0;
    for (int it = 0; it < iters; ++it) {
        float* st = smem + (SPREAD ? it % 3 : (it & 1)) * 8192;      // SPREAD: 3 stages, the one read was filled two steps ago
        if (NDMA && !SPREAD) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((d < NDMA / 2 ? baseW : baseX) + off + ((wave * NDMA + d) * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(st + ((wave * NDMA + d) * 64) * 4), 16, 0, 0);
            }
            off += 4 * NDMA * 256;
            if (off + 4 * NDMA * 256 > win_floats) off = 0;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");       // the previous step's loads have landed
        }
        if (BAR && !SPREAD) __builtin_amdgcn_s_barrier();
        const float* rd = smem + (SPREAD ? (it + 1) % 3 : ((it + 1) & 1)) * 8192;
        if (NDMA && SPREAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        if (SPREAD && BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (SPREAD && g < NDMA) {
                if (g == 0) { off += 4 * NDMA * 256; if (off + 4 * NDMA * 256 > win_floats) off = 0; }     // one DMA per MFMA group instead of a burst at the top of the step
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((g < NDMA / 2 ? baseW : baseX) + off + ((wave * NDMA + g) * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(st + ((wave * NDMA + g) * 64) * 4), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (NLDS) {
                if (g * 2 < NLDS) a = *reinterpret_cast<const f32x4*>(rd + (g * 256 + lane) * 4);
                if (g * 2 + 1 < NLDS) b = *reinterpret_cast<const f32x4*>(rd + (g * 256 + 64 + lane) * 4 + 2048 * (wave & 1));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc0, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    if (r == 123.456f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int NDMA, int NLDS, bool BAR, bool SPREAD = false>
void run(const char* name, const float* src, long long win, int occ, float* out, long long* clk, int share = 0) {
    const int iters = 4000, grid = 256 * occ;
    hipFuncSetAttribute(reinterpret_cast<const void*>(feed<NDMA, NLDS, BAR, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = SPREAD ? 98304 : 65536;      // 2 workgroups per CU at most, like conv_dma's BK64 tile (1 for the 3-stage variant)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed<NDMA, NLDS, BAR, SPREAD>), dim3(grid), dim3(256), lds, 0, src, win, iters, out, clk, share);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((feed<NDMA, NLDS, BAR, SPREAD>), dim3(grid), dim3(256), lds, 0, src, win, iters, out, clk, share);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
    printf("%-44s occ %d  %7.1f TFLOP/s  shader clock %6.0f MHz  (%.2f ms)\n", name, occ, fl / (ms * 1e-3) / 1e12, 100.0 * h[0] / h[1], ms);
}

int main() {
    float *src, *out; long long* clk;
    const size_t cap = (size_t)512 * (1 << 16) * 4 + (1 << 20);
    hipMalloc(&src, cap); hipMemset(src, 0, cap);
    hipMalloc(&out, 4096); hipMalloc(&clk, 64);
    const long long wins[3] = {1 << 13, 1 << 14, 1 << 16};      // 32 KB, 64 KB (L2-resident in total), 256 KB (128 MB in total) per workgroup
    for (long long win : wins) {
        printf("---- window %lld KB per workgroup ----\n", win * 4 / 1024);
        for (int occ = 1; occ <= 2; ++occ) {
            run<0, 0, false>("MFMA only", src, win, occ, out, clk);
            run<0, 16, false>("MFMA + 16 ds_read_b128", src, win, occ, out, clk);
            run<8, 0, false>("MFMA + 8 DMA (32 KB/WG/step)", src, win, occ, out, clk);
            run<4, 0, false>("MFMA + 4 DMA", src, win, occ, out, clk);
            run<2, 0, false>("MFMA + 2 DMA", src, win, occ, out, clk);
            run<8, 16, true>("MFMA + 16 ds_read + 8 DMA + barrier", src, win, occ, out, clk);
            run<4, 16, true>("MFMA + 16 ds_read + 4 DMA + barrier", src, win, occ, out, clk);
            run<4, 8, true>("MFMA + 8 ds_read + 4 DMA + barrier", src, win, occ, out, clk);
            run<2, 8, true>("MFMA + 8 ds_read + 2 DMA + barrier", src, win, occ, out, clk);
            run<8, 16, true>("... 16 ds_read + 8 DMA + barrier, shared", src, win, occ, out, clk, 1);
            run<4, 16, true>("... 16 ds_read + 4 DMA + barrier, shared", src, win, occ, out, clk, 1);
            run<8, 16, true>("... 8 DMA, shared, half the rows 16 B misaligned", src, win, occ, out, clk, 3);
            run<8, 16, true, true>("... 16 ds_read + 8 DMA + barrier, DMA spread", src, win, occ, out, clk, 1);
            run<4, 16, true, true>("... 16 ds_read + 4 DMA + barrier, DMA spread", src, win, occ, out, clk, 1);
        }
    }
    return 0;
}
