/*
 * liblds.so -- single-op test and micro-benchmark entry points.  NOT part of the drop-in boundary (include/lds.h): they exist so
 * that tests/ can check each kernel alone against the oracle and tools/ can time one launch.  Every call here allocates its own
 * temporaries and synchronises the stream before returning.
 */
#ifndef LDS_TEST_H
#define LDS_TEST_H

#include "lds.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- single-op entry points (used by the parity tests to check each kernel alone) ----------- */
typedef struct {                        /* the generic convolution (vocoder / front end path, conv_gemm)   */
    const float* x1; const float* x2;   /* dev inputs [B,C1,Tsrc], [B,C2,Tsrc] (x2 may be NULL)  */
    int C1, C2, Tsrc;
    const float* w;                     /* host, reference layout [Co, C1+C2, K]                  */
    const float* bias;                  /* host [Co] or NULL                                      */
    int Co, K, pad, dil;
    int act_in;                         /* 0 none, 2 LeakyReLU(slope) on the input                */
    float slope;
    const float* res;                   /* dev [B,Co,To] or NULL                                  */
    int epilogue;                       /* 0 none, 2 tanh                                         */
    int tile;                           /* 0 auto, else BM*1000+BN                                */
} lds_conv_test;
int lds_test_conv(const lds_conv_test* a, float* out, int B, void* stream);
/* ---- the UNet's K4P path, one op at a time (plain tensors in/out; layout conversion happens on the device) ---- */
typedef struct {
    const float* x1; const float* x2;   /* dev inputs [B,C1,T], [B,C2,T] (x2 may be NULL)                 */
    int C1, C2, T;
    const float* w; const float* bias;  /* host, reference layout [Co, C1+C2, K] / [Co]                   */
    int Co, K, stride, pad, ups;
    const float* res;                   /* dev [B,Cout,To] or NULL                                        */
    int epilogue;                       /* 0 none, 1 GEGLU                                                */
    int plain_out;                      /* 1: the kernel writes frame-major output directly               */
    int v_split;                        /* QKV: last third of the channels frame-major (1) or in attention's VT layout with head dim v_split (> 1) */
    int cfg;                            /* 0 auto, else BM*1000000 + BN*1000 + BK*10 + NST                */
} lds_dconv_test;
int lds_test_dconv(const lds_dconv_test* a, float* out, float* lnpart, int B, void* stream);
int lds_bench_dconv(const lds_dconv_test* a, float* out, int B, int iters, float* ms_out, char* cfg_out, size_t cfg_cap,
                    void* stream);
int lds_test_gn_apply(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                      const float* beta, const float* scale_shift, int silu, float* out, int B, void* stream);
/* average time of one streaming-GroupNorm launch on zero-filled tensors (tools/bench_gn.py) */
int lds_bench_gn_stream(int C1, int C2, int T, int B, int iters, float* ms_out, void* stream);
/* conv 1x1 (C -> Cm, GroupNorm partials from its epilogue) -> proj(GroupNorm(mid)) (Cm -> Co) as ONE launch with the normalisation folded into the
 * projection (csrc/kernels.h DmaConvArgs::gnf_part); tile_batch > 0: the latency mode's tile / cluster-split choices at that batch */
int lds_test_gn_fold_k4p(const float* x, const float* w1, const float* bias1, const float* gamma, const float* beta, float eps, int groups,
                         const float* w2, const float* bias2, float* mid, float* out, int B, int C, int Cm, int Co, int T, int cfg, int tile_batch,
                         void* stream);
/* the same through either kernel family (fmt -1 = exact fp32 K4P, 0 = three bf16 planes, 1 = two fp16 planes); the folded launch runs `reps` times
 * on the same inputs into out[rep][B][Co][T]: every repetition must equal the first bit for bit, also when workgroups share a CU */
int lds_test_gn_fold_split(const float* x, const float* w1, const float* bias1, const float* gamma, const float* beta, float eps, int groups,
                           const float* w2, const float* bias2, float* mid, float* out, int B, int C, int Cm, int Co, int T, int cfg, int tile_batch,
                           int fmt, int reps, void* stream);
/* the latency mode's cluster split-K hand-off (csrc/conv_dma.hip cluster_join): a K-tap (1 / 3) convolution Ci -> Co with the tile and cluster choices at
 * tile_batch = B, launched `reps` times back to back alternating between the inputs xa / xb (a stale partial is then the other input's) into
 * out[rep][B][Co][T]; ref_a / ref_b = the same tile shapes with one workgroup per tile; fmt -1 exact fp32, 0 / 1 split planes; cfg_out = the cluster
 * launch's configuration ("... KS<S> ...": S workgroups per tile) */
int lds_test_cluster_join(const float* xa, const float* xb, const float* w, const float* bias, int Ci, int Co, int K, int T, int B, int fmt, int reps,
                          float* out, float* ref_a, float* ref_b, char* cfg_out, size_t cfg_cap, void* stream);
/* mid = conv1x1(x) (+bias) written together with the epilogue's GroupNorm partial statistics; out = GroupNorm(mid)(+SiLU) by the
 * streaming pass that combines those partials -- the statistics path of the UNet (cfg: conv_dma tile code, 0 = auto) */
int lds_test_gn_chain_k4p(const float* x, const float* w1, const float* bias1, const float* gamma, const float* beta, float eps,
                          int groups, int silu, float* mid, float* out, int B, int C, int Co, int T, int cfg, void* stream);
int lds_test_ln_chain_k4p(const float* x, const float* w1, const float* w2, const float* gamma, const float* beta,
                          float eps, float* mid, float* out, int B, int C, int Co, int T, void* stream);
/* one residual step of a vocoder ResBlock1 on the K4P / LDS-DMA path: out = c2(lrelu(c1(lrelu(x)))) + x (reference models.py:186-192);
 * mode 0 plain output, 1 raw + LeakyReLU'd K4P outputs (returned plain in out / out_act), 2 running sum out = (acc + y) / div */
int lds_test_voc_step(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, int C, int T, int K, int dil, int mode,
                      const float* acc, float div, float* out, float* out_act, int B, void* stream);
/* qkv dev [B,3C,T] -> out dev [B,C,T]; softmax(QK^T/sqrt(C/heads))V per head */
int lds_test_attention_k4p(const float* qkv, float* out, int B, int C, int T, int heads, void* stream);
/* the same with both products on the fp16 matrix pipe, operands split in registers into two fp16 terms (the split-fp16 GEMM mode's attention) */
int lds_test_attention_f16math(const float* qkv, float* out, int B, int C, int T, int heads, void* stream);
/* the latency mode's configuration choice (judged at the actual batch: 32-query workgroups whose four waves share the keys); f16math 0 / 1 */
int lds_test_attention_latency(const float* qkv, float* out, int B, int C, int T, int heads, int f16math, void* stream);
int lds_test_conv_transpose(const float* x, const float* w /*host [Ci,Co,K]*/, const float* bias,
                            float* out, int B, int Ci, int Co, int T, int K, int stride, int pad,
                            float in_slope, void* stream);

/* the same single-op entry as lds_test_dconv / lds_bench_dconv through the split-bf16 kernel (conv_bf3.hip): plain tensors are
 * converted to K8B3 on the device; nprod = bf16 products per fp32 product (6 = the product path; 3 and 9 exist for the error study
 * of tools/split_bf16_probe.py on the 128 x 128 x BK32 1x1 tile only) */
int lds_test_dconv_bf3(const lds_dconv_test* a, float* out, float* lnpart, int B, int nprod, void* stream);
int lds_bench_dconv_bf3(const lds_dconv_test* a, float* out, int B, int iters, int nprod, float* ms_out, char* cfg_out, size_t cfg_cap,
                        void* stream);
/* lds_bench_dconv with consecutive launches rotating through the tile configurations cfgs[0 .. n): the cost of running code the previous
 * launch did not run (tools/bench_icache.py) */
int lds_bench_dconv_alt(const lds_dconv_test* a, float* out, int B, int iters, const int* cfgs, int n, float* ms_out, void* stream);
/* the same through either split-plane format: fmt 0 = three bf16 planes, 1 = two fp16 planes (csrc/k8b3.h); nprod 0 = the format's default */
int lds_test_dconv_split(const lds_dconv_test* a, float* out, float* lnpart, int B, int nprod, int fmt, void* stream);
int lds_bench_dconv_split(const lds_dconv_test* a, float* out, int B, int iters, int nprod, int fmt, float* ms_out, char* cfg_out, size_t cfg_cap,
                          void* stream);
int lds_test_split_roundtrip(const float* x, float* out, int B, int C, int T, int fmt, void* stream);
int lds_test_gn_apply_split(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                            const float* beta, const float* scale_shift, int silu, float* out, int B, int fmt, void* stream);
/* tuning only (tools/tune_split_rules.py): bit mask of alternative tile rules of the split-GEMM launcher, 0 = the shipped rules */
/* 0: the transformer blocks' GroupNorm runs as its own pass instead of folded into proj_in (A/B measurements and tests); default 1 */
int lds_debug_set_gn_fold(int on);
/* the narrow vocoder stages' residual steps: 1 (default) = one fused launch per step (csrc/voc_pair.hip), 0 = two convolution launches */
int lds_debug_set_voc_pair(int on);
/* experiment: 1 = a convolution's weights are read (one dword per 64-byte line) by a small launch immediately before the convolution's own:
 * the upper bound of what a weight prefetcher could give (tools/touch_weights_probe.py) */
int lds_debug_set_touch_weights(int on);
/* one residual step of ResBlock1 at 16 / 32 channels through the fused kernel: out = (acc ? acc : 0) + c2(lrelu(c1(lrelu(x)))) + x, / div
 * (reference models.py:186-192, 250-259); x, acc, out dev [B][C][T], weights host [C][C][K]; lengths host int32 [B] or null */
int lds_test_voc_pair(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, int C, int T, int K, int dil,
                      const float* acc, float div, const int32_t* lengths, float* out, int B, void* stream);
int lds_debug_set_split_rule(int rule);
/* plain [B,C,T] -> K8B3 -> plain: must return the input bit for bit (the three-term split is lossless) */
int lds_test_k8b3_roundtrip(const float* x, float* out, int B, int C, int T, void* stream);
/* GroupNorm(+scale/shift)(+SiLU) through the K8B3 streaming pass (statistics from gn_partials_bf3) */
int lds_test_gn_apply_bf3(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                          const float* beta, const float* scale_shift, int silu, float* out, int B, void* stream);

/* ONE token choice per row of logits [B][V] (dev) with the kernels of lds_lm_generate, after a history hist [B][n_hist] int64 (dev; what the repetition
 * penalty looks at); uniforms [B] (dev); out [B] int64 (dev).  top_k 0 = no top-k filter (HF: None / 0), else 1 .. 64 with ties of the k-th kept. */
int lds_test_lm_sample(const float* logits, int B, int V, int do_sample, int top_k, float top_p, float temperature, float repetition_penalty,
                       const float* uniforms, const int64_t* hist, int n_hist, int64_t* out, void* stream);

/* ---- debugging aids (tests/test_gpu_poison.py, tools/diag_trace.py) ----------------------------------------------------------
 * lds_debug_fill_u32: every 32-bit word of a device buffer = pattern.  Tests fill a caller workspace with NaN patterns (0x7fc07fc0 is a NaN
 * as fp32 and as two fp16 / bf16 halves) before a call: a kernel that reads a slot no kernel of THAT call wrote turns it into a NaN (or, behind
 * a select, into a result that differs from the zero-filled run's).
 * lds_debug_trace(1): clear the records and, while on, make every lds_unet_forward synchronise after each stage and keep a host copy of the
 * stage's output tensor (names "down0.res1.conv1", "mid.tfm.att2", ...; activation tensors in the handle's layout: K4P fp32, or split planes
 * [B][C/8][planes][T+2][8] 16-bit); lds_debug_trace(0) stops recording and keeps the records for lds_debug_trace_get.  Process-global. */
int lds_debug_fill_u32(void* dev, size_t n_words, uint32_t pattern, void* stream);
/* the workspace plan of lds_unet_forward(B, T) as text, one slot per line: "name offset bytes" */
int lds_debug_unet_plan(const lds_unet* u, int B, int T, char* buf, size_t cap);
int lds_debug_trace(int on);
int lds_debug_trace_count(void);
int lds_debug_trace_get(int i, char* name, size_t name_cap, const void** data, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LDS_TEST_H */
