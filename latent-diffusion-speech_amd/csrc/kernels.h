// Internal kernel interfaces of liblds (gfx950 only).  Activations: fp32 [B,C,T], T contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <string>

namespace lds {

// ---------------------------------------------------------------------------------------------
// conv_gemm (vocoder / front end): out[b,co,t] = epi( bias[co] + sum_{tap,ci} Wp[tap][ci][co] * f(X[b,ci, t + tap*dil - pad]) )
// f = optional LeakyReLU applied while staging; out-of-range taps read exact zeros.
// The MFMA is v_mfma_f32_32x32x2_f32 (exact fp32); A = packed weights (M = co), B = activations (N = t).
// ---------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_LRELU = 2 };      // activation applied to the input while staging (conv_gemm)
enum { EPI_NONE = 0, EPI_GEGLU = 1, EPI_TANH = 2 };

struct ConvArgs {
    // input: virtual channel-concat of two sources (x2 for ci >= C1)
    const float* x1; const float* x2;
    int C1, C2;
    int Tsrc;            // stored row length of the sources
    int Tin;             // logical input length (2*Tsrc when ups)
    long long xb1, xb2;  // batch strides (elements)
    // packed weights [KT][Ci][Mp]
    const float* w;
    int Mp, Co, Ci, KT, stride, dil, pad, ups;
    int act_in; float slope;
    // epilogue
    const float* bias;       // [Mp] packed-row bias or null
    const float* bias_bc;    // [B][Co] extra per-(batch,channel) bias or null
    const float* res;        // [B][Cout][To] residual or null
    int epi;
    int accum;               // 1: out = (out + y) / out_div  (MRF running sum), 0: out = y / out_div
    float out_div;           // 1.0 normally
    float* out;
    int Cout;                // channels of `out` (Co, or Co/phases for transposed conv)
    int To;                  // columns computed (N)
    // transposed-conv scatter: row m = co*phases + phi -> out[b][co][n*phases + phi - tpad] (phases=1: plain)
    int phases, tpad, Tout;  // Tout = row length of out
    int B;
    // Ragged batches (vocoder): valid frames per batch element of the OUTPUT tensor (device int32 [B]; output frames at and beyond are
    // written as zeros) and of the INPUT (frames at and beyond read as zeros); null = the buffers' lengths
    const int* vlen; const int* vlen_in;
};

// tile: 0 = auto, else BM*1000+BN in {128128, 64064, 128064, 64128, 32128}
hipError_t launch_conv_gemm(const ConvArgs& a, int tile, hipStream_t s);
const char* conv_gemm_last_config();

// conv_small (conv_small.hip): the vocoder's 16- / 32-channel resblock convolutions on v_mfma_f32_16x16x4_f32, weights resident in
// registers, plain tensors.  Same ConvArgs semantics (LeakyReLU on the input, bias, residual, running sum, division).
bool conv_small_applies(const ConvArgs& a);
hipError_t launch_conv_small(const ConvArgs& a, hipStream_t s);
bool conv_mono_applies(const ConvArgs& a);      // one output channel, k 7: the generator's conv_post as a stream
hipError_t launch_conv_mono(const ConvArgs& a, hipStream_t s);
const char* conv_small_last_config();

// voc_pair (voc_pair.hip): one residual step of the vocoder's ResBlock1 at 16 / 32 channels as one launch over plain [B][C][T] tensors:
// out = (accum ? out : 0) + c2(lrelu(c1(lrelu(x)))) + x, divided by out_div; c1 dilated by `dil`, both k = KT, "same" zero padding;
// frames at and beyond vlen[b] (device int32 [B], null = T) are outside the utterance: the intermediate reads as zero there and the
// output is written as zeros.  x and out must not alias (neighbouring tiles read x's halo).
struct VocPairArgs {
    const float* x; float* out;
    const float* w1; const float* b1; int Mp1;      // packed weights [KT][C/8][2][Mp][4] and packed-row biases (model.hip pack_conv)
    const float* w2; const float* b2; int Mp2;
    int C, KT, dil, B, T;
    int accum; float out_div; float slope;
    const int* vlen;
};
bool voc_pair_applies(int C, int KT, int dil);
hipError_t launch_voc_pair(const VocPairArgs& a, hipStream_t s);
const char* voc_pair_last_config();

// Weight packers (host side): reference layout -> [KT][Ci][Mp]
size_t packed_conv_elems(int Co, int Ci, int K, int* Mp_out);

// ---------------------------------------------------------------------------------------------
// conv_dma: the UNet's VALU-free GEMM over K4P activations (k4p.h, conv_dma.hip)
// ---------------------------------------------------------------------------------------------
struct DmaConvArgs {
    const float* x1; const float* x2;   // K4P sources [B][C1][Tsrc], [B][C2][Tsrc] (virtual channel concat)
    int C1, C2, Tsrc;
    const float* w;                     // packed weights [KT][Ci/8][2][Mp][4]
    const float* bias;                  // packed-row bias [Mp] or null
    int Mp, Co, Ci, KT, stride, pad, ups;
    const float* res;                   // K4P residual [B][Cout(K4P part)][To] or null
    int epi;                            // EPI_NONE | EPI_GEGLU
    float* out;                         // K4P [B][min(Cout, plain_from)][To], or plain [B][Cout][To] when out_plain
    int out_plain;
    int plain_from; float* out2;        // output channels >= plain_from go frame-major to out2 [B][Cout-plain_from][To] ...
    int vt_D;                           // ... or, when vt_D > 0, in attention's VT layout [B][(Cout-plain_from)/vt_D][ceil(To/4)][vt_D][4]
    float2* lnpart_out;                 // optional [B][C/32][To] per-frame (mean, M2) partials over 32-channel tiles
    float2* gnpart_out;                 // optional [B][C/16][ceil(To/32)] (mean, M2) of every (16 channels x 32 frames) block of the K4P output (GroupNorm statistics for gn_stream)
    // LayerNorm of the INPUT folded into the epilogue (weights pre-multiplied by gamma on the host):
    //   y[m,t] = rstd_t * (acc[m,t] - mean_t * ln_c1[m]) + ln_c2[m],  c1 = sum_c W[m,c]*gamma_c,  c2 = sum_c W[m,c]*beta_c + bias[m]
    // mean_t / rstd_t are combined per column from the producer's partials ln_part [B][ln_np][Tsrc]
    const float2* ln_part; int ln_np; float ln_eps; const float* ln_c1; const float* ln_c2;
    // GroupNorm of the INPUT (affine, no activation: the transformer's `norm` in front of proj_in, reference transformer_1d.py:256-266)
    // folded into a 1x1 convolution over ONE source (weights pre-multiplied by gamma on the host):
    //   y[m,t] = sum_g rstd_g * (sum_{c in g} Wg[m,c] x[c,t]) - sum_g rstd_g mean_g gnf_cg[g][m] + gnf_c2[m]
    //   gnf_cg[g][m] = sum_{c in g} W[m,c] gamma_c,  gnf_c2[m] = sum_c W[m,c] beta_c + bias[m]
    // The channels of a group are contiguous and every wave's share of a K-step lies inside one group: its activation operands are
    // multiplied by that group's rstd on their way from LDS to the MFMA (two packed multiplies per four MFMAs); the per-(batch, group)
    // statistics are combined from the producer's partials gnf_part [B][Ci/16][ceil(Tsrc/32)] (the same ones gn_stream reads) at kernel start.
    const float2* gnf_part; int gnf_groups; float gnf_eps; const float* gnf_cg; const float* gnf_c2;
    int Cout, To, B;
    // vocoder extensions (HiFi-VAEGAN MRF, reference models.py:161-262); the UNet leaves them at dil 1, xpad = opad = 1, rest 0 / 1.0
    int voc;                            // 1: vocoder kernel family (the fields below are honoured)
    int dil;                            // tap spacing in frames (1, 3, 5)
    int xpad, opad;                     // zero frames on each side of every K4P row of the inputs / of out, res, acc_in, out_act (>= pad)
    float act_slope;                    // != 0: LeakyReLU(slope) of the final value is what the next convolution reads ...
    float* out_act;                     // ... written here while `out` keeps the raw value (null: `out` receives the activated value)
    const float* acc_in;                // K4P running sum added before the division: out = (acc_in + y) / out_div
    float out_div;
    // polyphase ConvTranspose (vocoder upsamplers; weights packed by pack_convT: row m = co * phases + phase, KT = K / stride taps):
    // column n of row m is output frame n * phases + phase - ph_tpad of channel co, stored to K4P tensors with ph_Tout frames
    int ph_log2, ph_tpad, ph_Tout, ph_Cout;      // ph_Tout > 0 selects the mode; phases = 1 << ph_log2; ph_Cout = Co / phases
    // conv_bf3 only (split-bf16 path, k8b3.h): x1 / x2 / res / out are K8B3 tensors; out_f32 = 1 keeps the K4P-range output channels
    // in fp32 K4P instead (q / k rows for the attention kernel)
    int out_f32;
    float acc_scale;                    // != 0: the accumulators are multiplied by it first (fp16-plane weights are stored times a power of two)
    // Batch size the tile / split choice is judged at.  0 = the nominal per-GPU batch (16): the choice then never depends on the actual
    // batch and an utterance's result is bit-identical for any batch split.  > 0 (lds_unet_set_latency_mode): the actual batch, so that
    // one or two utterances spread over the chip -- same tolerances against the oracle, not bit-identical with batched results.
    int tile_batch;
    // Ragged batches (k4p.h ragged_len): per-utterance lengths at the UNet's input resolution and the levels of this launch's input and
    // output tensors; output frames at and beyond an utterance's length are written as zeros.  nullptr: no masking.
    const int* lens; int lvl_in, lvl_out;
    // ... and of the vocoder's stages, whose lengths multiply instead of halving: valid OUTPUT frames per batch element (device int32 [B];
    // of the scattered tensor in polyphase mode), written as zeros beyond; null = none
    const int* vlen;
    // Cluster split-K (latency mode only; conv_dma.hip cluster_join): ksplit = S > 1 workgroups share an output tile, each reducing 1/S
    // of the K-steps; kpart = scratch for the partial tiles (tiles x 4 waves x S x 1024 floats), kcount = one zeroed counter per tile,
    // left zeroed.  The launchers choose S (conv_dma_cluster_split) when tile_batch > 0 and kpart / kcount are given.
    int ksplit;
    float* kpart; unsigned* kcount;
    long long kpart_cap; int kcount_cap;      // capacities (floats / counters)
};
// cfg: 0 = auto, else BM*1000000 + BN*1000 + BK*10 + NST
hipError_t launch_conv_dma(const DmaConvArgs& a, int cfg, hipStream_t s);
// Resnet tail in one launch (reference resnet.py:636-641): out = conv2_k3(h) + shortcut_1x1([x ; skip]) + bias.  `a3` carries the k 3
// convolution's sources / weights (its epilogue fields are ignored), `a1` the 1x1 shortcut's sources / weights AND the epilogue
// (bias = the two biases added on the host, GroupNorm partials, output).  Both reductions run into the same accumulators: one
// launch and one K4P round trip less per resnet that changes its width.  hipErrorNotSupported: no fused variant for these shapes
// (the caller launches the two convolutions separately).
hipError_t launch_conv_dma_pair(const DmaConvArgs& a3, const DmaConvArgs& a1, hipStream_t s);
bool conv_dma_pair_applies(const DmaConvArgs& a3, const DmaConvArgs& a1);
const char* conv_dma_last_config();

// ---------------------------------------------------------------------------------------------
// conv_bf3: the same operator on the bf16 matrix pipe with fp32-equivalent split operands (conv_bf3.hip, k8b3.h).  Same argument
// struct; activations (x1, x2, res, out) are K8B3 tensors, weights packed [KT][Ci/8][3][Mp][8] bf16 (pack_conv_bf3, model.hip).
// nprod: bf16 products per fp32 product (6 = product path; 3 / 9 exist for tools/split_bf16_probe.py's error study only).
// ---------------------------------------------------------------------------------------------
// fmt: FMT_BF16X3 (three bf16 planes, 6 products) or FMT_F16X2 (two fp16 planes, 3 products; k8b3.h); nprod 0 = the format's default
hipError_t launch_conv_bf3(const DmaConvArgs& a, int cfg, int nprod, int fmt, hipStream_t s);
hipError_t launch_conv_bf3_pair(const DmaConvArgs& a3, const DmaConvArgs& a1, int fmt, hipStream_t s);
bool conv_bf3_pair_applies(const DmaConvArgs& a3, const DmaConvArgs& a1);
const char* conv_bf3_last_config();
void conv_bf3_set_debug_rule(int r);      // tuning only (lds_debug_set_split_rule)
// split-plane helpers (k8b3_ops.hip): plain [B][C][T] <-> K8B3 / K8H2 (channels [c_off, c_off + C) of a tensor with Ctot channels); fmt as above
hipError_t launch_to_k8b3(const float* in, void* out, int B, int C, int T, int Ctot, int c_off, hipStream_t s, int fmt = 0, const int* lens = nullptr);
hipError_t launch_from_k8b3(const void* in, float* out, int B, int C, int T, hipStream_t s, int fmt = 0);
// GroupNorm(+scale/shift)(+SiLU) of the virtual concat [x1;x2], split planes in and out; same statistics path as launch_gn_stream
hipError_t launch_gn_stream_bf3(const void* x1, const void* x2, int C1, int C2, int T, int groups, float eps, const float* gamma,
                                const float* beta, const float* scale_shift, int ss_stride, int ss_off, int silu, const float2* gp1,
                                const float2* gp2, void* y, int B, hipStream_t s, int fmt = 0, const int* lens = nullptr, int lvl = 0);
hipError_t launch_gn_partials_bf3(const void* x, int C, int T, float2* gp, int B, hipStream_t s, int fmt = 0);
hipError_t launch_resample_k8b3(const void* in, void* out, int B, int C, int Tin, int Tout, hipStream_t s, int fmt = 0, const int* lens = nullptr, int lvl_in = 0, int lvl_out = 0);

// ---------------------------------------------------------------------------------------------
// K4P helpers (k4p_ops.hip)
// ---------------------------------------------------------------------------------------------
// plain [B][C][T] -> channels [c_off, c_off+C) of a K4P tensor with Ctot channels (pads of those rows zeroed)
hipError_t launch_to_k4p(const float* in, float* out, int B, int C, int T, int Ctot, int c_off, hipStream_t s, const int* lens = nullptr);
hipError_t launch_from_k4p(const float* in, float* out, int B, int C, int T, hipStream_t s);
hipError_t launch_from_k4p_pad(const float* in, float* out, int B, int C, int T, int pad, hipStream_t s);
// plain [B][C][T] -> attention's VT layout [B][C/D][ceil(T/4)][D][4] (tail keys zeroed); the UNet gets this layout straight
// from the QKV convolution's epilogue, this kernel serves the stand-alone attention entry point
hipError_t launch_plain_to_vt(const float* in, float* out, int B, int C, int T, int D, hipStream_t s);
// GroupNorm of the virtual concat [x1;x2] (K4P) -> y (K4P, C1+C2 channels):
//   y = act(((x - mean_g) * rstd_g * gamma + beta) * (1 + scale) + shift), act = SiLU if silu
// One streaming pass (1 read + 1 write, one workgroup per (batch, 8-channel block)): the group statistics come from
// the per-(16 channels x 32 frames) (mean, M2) partials gp1 / gp2 [B][C/16][ceil(T/32)] that the producers of x1 / x2
// wrote in their epilogues (DmaConvArgs::gnpart_out) and are combined per workgroup with Chan's formula in a fixed order
// while the tensor loads are in flight.
hipError_t launch_gn_stream(const float* x1, const float* x2, int C1, int C2, int T, int groups, float eps,
                            const float* gamma, const float* beta, const float* scale_shift, int ss_stride, int ss_off,
                            int silu, const float2* gp1, const float2* gp2, float* y, int B, hipStream_t s, const int* lens = nullptr, int lvl = 0);
// the same partials computed from a K4P tensor by a stand-alone pass (tensors not produced by conv_dma; test entry points)
hipError_t launch_gn_partials(const float* x, int C, int T, float2* gp, int B, hipStream_t s);
// nearest-neighbour resample along frames (K4P -> K4P), reference F.interpolate(size=Tout)
// plain [B][C][T] -> K4P with `pad` zero frames per side: raw copy and (slope != 0) LeakyReLU copy (either output may be null)
hipError_t launch_to_k4p_act(const float* in, float* raw, float* act, float slope, int B, int C, int T, int pad, hipStream_t s, const int* vlen = nullptr);
// zero the `pad` frames on both sides of every row of a K4P tensor (tensors whose writers only store real frames)
hipError_t launch_k4p_zero_pads(float* x, int B, int C, int T, int pad, hipStream_t s);
// lens: per-utterance lengths (ragged_len) at levels lvl_in / lvl_out of the two tensors
hipError_t launch_resample_k4p(const float* in, float* out, int B, int C, int Tin, int Tout, hipStream_t s, const int* lens = nullptr, int lvl_in = 0, int lvl_out = 0);
// self-attention: q,k in K4P (tensor qk [B][2C][T]: q channels 0..C-1, k channels C..2C-1), v in the VT layout
// [B][heads][ceil(T/4)][D][4] (key tail zeroed); out K4P [B][C][T]
// tile_batch: as DmaConvArgs::tile_batch (0 = nominal batch of 16)
hipError_t launch_attention_k4p(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, hipStream_t s, int tile_batch = 0,
                                const int* lens = nullptr, int lvl = 0);
// the same with the output written as a K8B3 tensor (split-bf16 path: the output feeds the to_out projection)
hipError_t launch_attention_k4p_f16math(const float* qk, const float* vt, float* out, int B, int C, int T, int heads, hipStream_t s, int tile_batch = 0);
hipError_t launch_attention_k4p_out_bf3(const float* qk, const float* vt, void* out, int B, int C, int T, int heads, hipStream_t s, int fmt = 0, int tile_batch = 0,
                                        const int* lens = nullptr, int lvl = 0);

// ---------------------------------------------------------------------------------------------
// Small dense layers with N = batch columns (time embedding path)
// out[b][m] = sum_k W[m][k] * g(in[b][k]) + bias[m];  g = identity | SiLU | sinusoid(t[b])
// ---------------------------------------------------------------------------------------------
enum { IN_PLAIN = 0, IN_SILU = 1, IN_SINUSOID = 2 };
hipError_t launch_small_linear(const float* W, const float* bias, const float* in, int in_stride, int in_mode,
                               const float* freqs, float* out, int out_stride, int out_silu, int M, int K, int B, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Elementwise sampler updates over n = B*M*T elements
// ---------------------------------------------------------------------------------------------
enum {
    EW_X0 = 1,        // out = (a - c0*b) / c1                         (x0 = (x - sigma*eps)/alpha)
    EW_AXPBY = 2,     // out = c0*a - c1*b                             (DPM first-order / UniPC x_t_)
    EW_DPM2 = 3,      // out = c0*a - c1*b - c2*(c3*(b - c))           (DPM-Solver++ 2M)
    EW_UNIPC_PRED = 4,// out = a - c0*(c1*((c - b)/c2))                (a = x_t_, b = m0, c = m_prev1)
    EW_UNIPC_CORR = 5,// out = a - c0*(c1*((d - b)/c2) + c3*(c - b))   (a = x_t_, b = m0, c = m_t, d = m_prev1)
    EW_UNIPC_CORR1 = 6,// out = a - c0*(c3*(c - b))                    (order-1 corrector)
    EW_DDPM = 7,      // x0 = clamp(c0*a - c1*b, -1, 1); out = c2*x0 + c3*a + c4*c   (a = x, b = eps, c = noise)
    EW_DDIM = 8,      // out = c0*(a/c1 + c2*b)
    EW_PLMS_PRED = 9, // out = a + c0*(c1*a - c2*b)
    EW_LIN4 = 10,     // out = (c0*a + c1*b + c2*c + c3*d) / c4
    EW_COPY = 11
};
hipError_t launch_ew(int op, float* out, const float* a, const float* b, const float* c, const float* d,
                     float c0, float c1, float c2, float c3, float c4, long long n, hipStream_t s);
hipError_t launch_dpm_step(float* x, const float* eps, float* m0, const float* m1, float sigma, float alpha, int second, float c4, float c5, float c6,
                           float c7, long long n, hipStream_t s);      // EW_X0 + EW_AXPBY / EW_DPM2 in one pass
hipError_t launch_fill(float* p, float v, long long n, hipStream_t s);
hipError_t launch_touch_lines(const float* p, long long bytes, hipStream_t s);      // experiment: lds_debug_set_touch_weights
struct FloatList64 { float f[64]; };
// dst[0..n) = host[0..n), n <= 64, carried in the kernel arguments (the host array is read during the call only)
hipError_t launch_set_list(float* dst, const float* host, int n, hipStream_t s);
hipError_t launch_transpose(const float* in, float* out, int B, int R, int C, float scale, hipStream_t s);
hipError_t launch_gather_rows(const float* table, const int64_t* idx, int idx_off, float* out, int B, int C,
                              int nrows, hipStream_t s);
hipError_t launch_resample_frames(const float* in, float* out, int B, int Tin, int Tout, int C, float step, hipStream_t s);
hipError_t launch_resample_nearest(const float* in, float* out, int B, int C, int Tin, int Tout, hipStream_t s);


// ---------------------------------------------------------------------------------------------
// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
// ---------------------------------------------------------------------------------------------
// Called by a launcher inside an open ProfScope of this thread: two fresh events for hipExtLaunchKernelGGL's start / stop slots, which
// the runtime binds to the kernel's own begin / end timestamps (no extra packets on the stream, so neighbouring launches still
// overlap their dispatch as in an un-instrumented run).  false: no scope is open / the profiler is off -- launch normally.
bool prof_attach_events(hipEvent_t* start, hipEvent_t* stop);
// records the thread-local message returned by lds_last_error() and returns `code` (model.hip)
int set_error(int code, const char* fmt, ...);

struct ProfScope {
    bool on;
    hipStream_t s;
    int idx = -1;
    bool attached = false;      // the launch itself carries the two events (prof_attach_events): the destructor records nothing
    // attachable: the launcher inside the scope will bind the events to its dispatch (prof_attach_events): no event is recorded here
    ProfScope(hipStream_t st, const char* name, double flops, double bytes, bool attachable = false);
    ~ProfScope();
    void rename(const std::string& n);
};

// hipFuncAttributeMaxDynamicSharedMemorySize must be raised once per (kernel, device); `done` is that kernel's per-device bit mask
inline hipError_t ensure_max_dynamic_lds(const void* kern, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      // idempotent: a concurrent second call is harmless
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace lds
