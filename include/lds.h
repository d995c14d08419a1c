/*
 * liblds.so -- C ABI of the MI355X-native latent-diffusion speech sampler.
 *
 * The reference (bfloat16/latent-diffusion-speech) is pure Python on PyTorch and has no
 * FFI layer of its own; its drop-in boundary is the module API
 *   diffusion.unit2mel.Unit2Mel / load_model_vocoder   (reference diffusion/unit2mel.py:18-88)
 *   diffusion.diffusion.GaussianDiffusion.forward      (reference diffusion/diffusion.py:189-343)
 *   diffusion.vocoder.Vocoder.infer                    (reference diffusion/vocoder.py:32-33)
 * which the modules under latent-diffusion-speech_amd/diffusion/ re-expose unchanged.  Those modules keep
 * tensors, streams and checkpoints in PyTorch and call the entry points below through ctypes
 * (latent-diffusion-speech_amd/lds/native.py); INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer marked "dev" is device memory owned by the caller (a torch tensor);
 *     "host" pointers are read during the call only.
 *   - activations are fp32, channel-major [B, C, T] with the frame axis contiguous.
 *   - `stream` is a hipStream_t passed as void*; the forward / run calls only enqueue work on it and
 *     never synchronise the device (host arrays they are handed -- the sampler's table -- are copied into
 *     the launches' kernel arguments during the call).  Exceptions, documented at the function:
 *     lds_lm_generate polls for EOS, *_create upload weights, lds_prof_summary reads events.
 *     Handles are immutable after create EXCEPT for the two mode switches of the denoiser (lds_unet_set_gemm_mode,
 *     lds_unet_set_latency_mode): calls on different streams / threads are safe as long as each has its own
 *     workspace and nobody switches a mode meanwhile -- a switch while another thread is inside a forward or
 *     sampler call of the same handle is refused with LDS_EBUSY (the handle counts the calls in progress).
 *   - return value 0 = ok, negative LDS_E* on error; lds_last_error() gives the message
 *     (thread-local).  Nothing throws or aborts.
 */
#ifndef LDS_H
#define LDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDS_OK 0
#define LDS_EINVAL (-1)    /* bad argument / unsupported shape */
#define LDS_ENOMEM (-2)    /* workspace too small or device allocation failed */
#define LDS_EHIP (-3)      /* a HIP runtime call failed */
#define LDS_EMISSING (-4)  /* a required weight tensor was not supplied */
#define LDS_EBUSY (-5)     /* a mode switch while another thread is inside a forward / sampler call of the same handle */

typedef struct lds_unet lds_unet;
typedef struct lds_embed lds_embed;
typedef struct lds_vocoder lds_vocoder;

const char* lds_last_error(void);
int lds_version(void);

/* ---- denoiser: UNet1DConditionModel (reference diffusion/unet1d/unet_1d_condition.py:61-1036,
 *      configured as in reference diffusion/unit2mel.py:62-71) ------------------------------ */
typedef struct {
    int out_dims;              /* mel channels M (x / eps channels)                       */
    int n_hidden;              /* condition channels H; UNet in_channels = M + H          */
    int n_layers;              /* layers_per_block                                        */
    int n_heads;               /* attention heads (reference attention_head_dim)          */
    int norm_groups;           /* GroupNorm groups (8)                                    */
    int n_blocks;              /* len(block_out_channels), <= 8                           */
    int block_out_channels[8];
} lds_unet_cfg;

/* names[i] are the reference's state_dict keys of UNet1DConditionModel (e.g.
 * "down_blocks.0.resnets.0.conv1.weight"), host_ptrs[i] fp32 host arrays in the reference's
 * own layouts, numel[i] their element counts.  Weights are re-laid-out and uploaded once. */
int lds_unet_create(const lds_unet_cfg* cfg, int n_tensors, const char* const* names,
                    const float* const* host_ptrs, const int64_t* numel, lds_unet** out);
void lds_unet_destroy(lds_unet* u);
int lds_unet_workspace_bytes(const lds_unet* u, int B, int T, size_t* out);

/* One denoiser evaluation = `self.denoise_fn(cat([x, cond], dim=-2), t).sample`
 * (reference diffusion/diffusion.py:105-106,223-226).
 * x dev [B,M,T], cond dev [B,H,T], t dev [B] (fp32, may be fractional), eps dev [B,M,T]. */
int lds_unet_forward(lds_unet* u, const float* x, const float* cond, const float* t, float* eps,
                     void* ws, size_t ws_bytes, int B, int T, void* stream);

/* ---- sampler: GaussianDiffusion.forward(infer=True) loops (reference diffusion/diffusion.py:
 *      214-341; DPM-Solver++ dpm_solver_pytorch.py:1171-1213; UniPC uni_pc.py:590-658) -------- */
#define LDS_METHOD_DPM_SOLVER_PP 1
#define LDS_METHOD_UNIPC 2
#define LDS_METHOD_DDPM 3
#define LDS_METHOD_DDIM 4
#define LDS_METHOD_PLMS 5
#define LDS_TABLE_STRIDE 16

/* `table` (host, n_rows x LDS_TABLE_STRIDE floats) holds the per-step scalar coefficients,
 * computed by the caller from the noise schedule in fp32 (layout per method: see
 * latent-diffusion-speech_amd/diffusion/diffusion.py and lds_sampler_run in csrc/model.hip).  The table is read on the
 * host during the call (it may be freed afterwards); no device synchronisation is performed.
 * x dev [B,M,T] is x_T on entry and the sample on return; cond dev [B,H,T];
 * noise dev [n_rows,B,M,T] for DDPM (one draw per step), else NULL. */
int lds_sampler_run(lds_unet* u, int method, int n_rows, const float* table, const float* cond,
                    float* x, const float* noise, void* ws, size_t ws_bytes, int B, int T,
                    void* stream);
int lds_sampler_workspace_bytes(const lds_unet* u, int B, int T, size_t* out);

/* Ragged batches (the reference's 22_infer_tts.py:76-114 synthesises sentences of different lengths; a padded torch batch would change
 * every GroupNorm statistic and attention row): `lengths` (host int32 [B], 1 <= lengths[b] <= T, B <= 64) are the utterances' own frame
 * counts inside buffers of T frames.  Every kernel stops an utterance's statistics and attention keys at its length and writes zeros
 * beyond it (the convolutions' zero padding, as when the utterance runs alone), and the decoder resamples every utterance to its own skip
 * lengths.  Frames [0, lengths[b]) of utterance b equal the utterance run alone at its own length within the stated tolerances (not bit
 * for bit: tile shapes follow the buffer length); frames beyond are unspecified in x / zero in eps.  Every GEMM mode. */
int lds_unet_forward_ragged(lds_unet* u, const float* x, const float* cond, const float* t, const int32_t* lengths, float* eps, void* ws,
                            size_t ws_bytes, int B, int T, void* stream);
int lds_sampler_run_ragged(lds_unet* u, int method, int n_rows, const float* table, const float* cond, float* x, const float* noise,
                           const int32_t* lengths, void* ws, size_t ws_bytes, int B, int T, void* stream);

/* ---- front end: Unit2Mel.forward's condition (reference diffusion/unit2mel.py:79-82) ---------
 * cond[b,:,t] = unit_embed(units[b,t,:]) + spk_embed[spk_id[b]-1]                               */
int lds_embed_create(int input_channel, int n_hidden, int n_spk, const float* unit_w,
                     const float* unit_b, const float* spk_w, lds_embed** out);
void lds_embed_destroy(lds_embed* e);
int lds_embed_workspace_bytes(const lds_embed* e, int B, int T, size_t* out);
/* units dev [B,T,input_channel], spk_id dev [B] int64 (1-based, may be NULL when n_spk<=1),
 * cond dev [B,n_hidden,T]. */
int lds_embed_forward(lds_embed* e, const float* units, const int64_t* spk_id, float* cond,
                      void* ws, size_t ws_bytes, int B, int T, void* stream);

/* out[b,c,r] = in[b,r,c] / div  (the [B,T,M] <-> [B,M,T] layout changes at the module
 * boundary: reference diffusion/diffusion.py:190,342-343, hifi_vaegan.py:54). */
int lds_transpose(const float* in, float* out, int B, int R, int C, float div, void* stream);
/* ---- token -> unit-embedding step in front of the path (reference 22_infer_tts.py:43-52,100-110) -------------
 * out[i,:] = table[idx[i],:]: the k-means codebook lookup `semantic_embedding(semantic_token)` (nn.Embedding over
 * cluster_centers_ [n_rows, C]); an index outside [0, n_rows) yields a NaN row (nn.Embedding raises). */
int lds_gather_rows(const float* table, const int64_t* idx, float* out, int n_idx, int C, int n_rows, void* stream);
/* out[b,i,:] = in[b, min((int)floorf(i*step), Tin-1), :]: F.interpolate(mode='nearest') of units_forced_alignment
 * (reference tools/tools.py:193-223) on frame-major units [B,Tin,C] -> [B,Tout,C]; step = 1/scale_factor (fp32). */
int lds_resample_frames(const float* in, float* out, int B, int Tin, int Tout, int C, float step, void* stream);
/* out = c0*a + c1*b over n elements (q_sample of shallow diffusion, reference diffusion.py:169-171) */
int lds_axpby(float* out, const float* a, const float* b, float c0, float c1, int64_t n, void* stream);

/* ---- vocoder: HiFi-VAEGAN Generator (reference encoder/hifi_vaegan/modules/models.py:224-272,
 *      hifi_vaegan.py:52-65) ------------------------------------------------------------------- */
typedef struct {
    int inter_channels;             /* latent / mel channels                                */
    int upsample_initial_channel;
    int n_ups;                      /* <= 8 */
    int upsample_rates[8];
    int upsample_kernel_sizes[8];
    int resblock;                   /* 1 or 2 */
    int n_kernels;                  /* <= 4 */
    int resblock_kernel_sizes[4];
    int n_dil;                      /* dilations per resblock, <= 4 */
    int resblock_dilation_sizes[4][4];
} lds_vocoder_cfg;

/* names: the reference Generator's state_dict keys; weight-norm pairs (`*.weight_g`,
 * `*.weight_v`) are folded here like remove_weight_norm() (reference hifi_vaegan.py:61);
 * already-folded `*.weight` tensors are accepted too. */
int lds_vocoder_create(const lds_vocoder_cfg* cfg, int n_tensors, const char* const* names,
                       const float* const* host_ptrs, const int64_t* numel, lds_vocoder** out);
void lds_vocoder_destroy(lds_vocoder* v);
int lds_vocoder_workspace_bytes(const lds_vocoder* v, int B, int T, size_t* out);
/* z dev [B,C,T] -> wav dev [B,1,T*prod(upsample_rates)] */
int lds_vocoder_forward(lds_vocoder* v, const float* z, float* wav, void* ws, size_t ws_bytes,
                        int B, int T, void* stream);
/* Ragged batch (see lds_sampler_run_ragged): lengths host int32 [B] (B <= 64) = the utterances' own frame counts inside z [B,C,T].  Every
 * stage of the generator writes zeros beyond an utterance's (up-sampled) length -- the zero padding its convolutions see when it runs
 * alone -- and z itself is read as zeros there; samples [0, lengths[b] * prod(rates)) of wav[b] equal the utterance decoded alone within the
 * stated tolerance, the samples beyond are zeros. */
int lds_vocoder_forward_ragged(lds_vocoder* v, const float* z, const int32_t* lengths, float* wav, void* ws, size_t ws_bytes,
                               int B, int T, void* stream);

/* ---- text2semantic: RoFormer encoder prefill + cached autoregressive decode (reference text2semantic/roformer/roformer.py:59-255
 *      over HF transformers RoFormerModel / RoFormerForCausalLM + GenerationMixin; called from 22_infer_tts.py:76-98) ------------- */
typedef struct lds_lm lds_lm;
typedef struct {
    int hidden, heads, inter;          /* 256, 8, 512 (reference configs/config.yaml:60-83)                          */
    int enc_layers, dec_layers;        /* 4, 1                                                                        */
    int text_vocab, type_vocab;        /* phone symbols + 3, tones + 1                                                */
    int sem_vocab;                     /* semantic_kmeans_num + 3                                                     */
    int n_spk_rows;                    /* rows of spk_emb (n_spk + 1), 0 = no speaker embedding                       */
    int max_pos;                       /* rows of the sinusoid tables                                                 */
    float eps;                         /* layer_norm_eps                                                              */
    int sem_bos, sem_eos, sem_pad;
} lds_lm_cfg;
/* names = the reference Roformer.state_dict() keys (text_encoder.*, semantic_decoder.*, spk_emb.weight), fp32 host arrays */
int lds_lm_create(const lds_lm_cfg* cfg, int n_tensors, const char* const* names, const float* const* host_ptrs, const int64_t* numel,
                  lds_lm** out);
void lds_lm_destroy(lds_lm* lm);
int lds_lm_workspace_bytes(const lds_lm* lm, int B, int L, int max_length, size_t* out);
/* phone, tone, spk_id: dev int64 [B,L] (spk_id may be NULL) -> enc dev [B,L,hidden] = encoder_hidden_states (roformer.py:196-204).
 * enc_len: dev int32 [B] or NULL -- the padding mask of a right-padded batch (reference roformer.py:182,209-214: attention_mask), given as
 * the number of real positions per row: keys at positions >= enc_len[b] get probability 0 in every encoder self-attention. */
int lds_lm_encode(lds_lm* lm, const int64_t* phone, const int64_t* tone, const int64_t* spk_id, const int32_t* enc_len, float* enc, void* ws,
                  size_t ws_bytes, int B, int L, void* stream);
/* Roformer.generate (roformer.py:179-240): greedy (do_sample 0) or RepetitionPenalty -> Temperature -> TopK -> TopP -> one draw per
 * step.  enc_len as above (roformer.py:229-236: encoder_attention_mask on the decoder's cross-attention) or NULL.  top_k as HF's
 * TopKLogitsWarper: 1 .. 64 keeps the k largest scores AND every score tied with the k-th (up to 64 survivors in all); 0 (HF: None / 0) applies
 * no top-k filter -- softmax, the nucleus cut and the draw then run over the whole vocabulary (a slower, optional path).
 * uniforms dev [max_length-1][B]: the draw is the inverse-CDF rule over the vocabulary order with these numbers (torch's own
 * multinomial stream cannot be reproduced outside torch).  tokens dev int64 [B][max_length] (BOS first; finished sequences padded);
 * logits_out optional dev [max_length-1][B][sem_vocab]; *n_tokens_host = length of the returned sequences incl. BOS.  The call
 * synchronises the stream every 8 steps to poll for EOS. */
int lds_lm_generate(lds_lm* lm, const float* enc, const int32_t* enc_len, int B, int L, int max_length, int do_sample, int top_k, float top_p,
                    float temperature, float repetition_penalty, const float* uniforms, int64_t* tokens, float* logits_out, int* n_tokens_host,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- per-launch HIP-event timing for bench.py's roofline leg (off by default) ------------------
 * lds_prof_enable(1) clears and starts recording one event pair per kernel launch on the launch
 * stream; lds_prof_summary synchronises them and writes a JSON list of
 * {name,count,ms,flops,bytes} aggregates (algorithmic flops/bytes per kernel family). */
int lds_prof_enable(int on);
int lds_prof_summary(char* buf, size_t cap);

/* ---- exact-fp32 vs split-operand GEMMs on the 16-bit matrix pipe (csrc/conv_bf3.hip, csrc/k8b3.h) ------------------------------------
 * mode 0 (default, what bench.py's `value` is measured in): every convolution / linear layer of the UNet on the exact-fp32 MFMA (a k-ordered
 * fmaf chain).
 * mode 1 (three bf16 terms per operand, lossless, six products) was a whole-UNet mode in round 3 and is REMOVED: it met the tolerances and ran no
 * faster than mode 0; lds_unet_set_gemm_mode(1) returns LDS_EINVAL.  The kernel format survives for the single-op probes
 * (include/lds_test.h lds_test_dconv_split, tools/split_bf16_probe.py).
 * mode 2 (opt-in, experimental, never a default): two fp16 terms per operand -- 22 significand bits, NARROWER than the reference's fp32 -- three
 * products, fp32 accumulate; weights carry a per-layer power-of-two scale.  PRECONDITION on the activations: every tensor between kernels must
 * stay below 65,504 in magnitude (an overflow becomes an infinity, then a NaN: loud) and should live at a scale of 2^-3 or more: below that
 * the second fp16 term is subnormal and the tensor keeps fewer than 22 bits (down to 11 at 6e-5), silently.  Attention probabilities are
 * exempt (formed times 2^12 inside the kernel).  UNet1DConditionModel.check_split_f16_ranges (Python) reports every tensor's range for given
 * inputs and raises outside [2^-3, 2^15].  The first switch packs the split weights (host work + upload); later switches only flip the flag.
 * Switching modes is not thread-safe against concurrent forwards on the same handle (one mode per handle lifetime is the supported use). */
#define LDS_GEMM_F32 0
#define LDS_GEMM_SPLIT_BF16 1      /* removed: LDS_EINVAL */
#define LDS_GEMM_SPLIT_F16 2
int lds_unet_set_gemm_mode(lds_unet* u, int mode);
int lds_unet_get_gemm_mode(const lds_unet* u);

/* ---- latency mode (off by default): the one-sentence caller (reference 22_infer_tts.py:100-114 synthesises one utterance per call) ----
 * By default every tile / split choice is made at the nominal per-GPU batch of 16, so that an utterance's result is bit-identical
 * alone, inside any batch and for any shard count -- at the price that one utterance occupies 1/16 of the chip.  With the mode on the
 * choices follow the ACTUAL batch: smaller tiles, and for deep reductions a cluster of up to 16 workgroups per output tile, each
 * reducing a share of the K range and the last one to arrive summing the partial tiles in a fixed order (deterministic; csrc/conv_dma.hip
 * cluster_join).  Same tolerances against the reference; results are NOT bit-identical with the default mode's.  The workspace grows by
 * 16 MB (lds_unet_workspace_bytes / lds_sampler_workspace_bytes report it).  Applies to every GEMM mode. */
int lds_unet_set_latency_mode(lds_unet* u, int on);
int lds_unet_get_latency_mode(const lds_unet* u);

#ifdef __cplusplus
}
#endif
#endif /* LDS_H */
