"""`Roformer` / `get_model` with the reference's names, constructor arguments, attributes, state_dict keys and `generate` signature
(reference text2semantic/roformer/roformer.py:8-255).  The reference drives HF transformers' RoFormerModel (phone/tone encoder) and
RoFormerForCausalLM (+ cross-attention) through GenerationMixin; here the encoder prefill, the key/value-cached decode loop and the
token choice run in liblds (csrc/lm.hip).  torch keeps the parameters and draws the sampling uniforms.

Scope: phone mode (the 'text' mode fetches a BERT tokenizer from the hub), inference only (`forward` = teacher-forced training
path), num_beams = 1, no n-gram blocking, no end gate, no padding mask -- what 22_infer_tts.py:83-98 uses."""
import numpy as np
import torch
from torch import nn

from lds import arch, native
from lds.paramtree import ParamTree


def _get(cfg, key, default=None):
    """read a field from a dict, an HF config object or any namespace"""
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def get_model(n_spk, **kwargs):
    return Roformer(
        encoder_config=dict(kwargs["model"]["encoder"], is_decoder=False),
        decoder_config=dict(kwargs["model"]["decoder"], is_decoder=True),
        mode=kwargs["model"]["mode"],
        semantic_kmeans_num=kwargs["model"]["semantic_kmeans_num"],
        codebook_path=kwargs["model"]["codebook_path"],
        n_spk=n_spk,
        use_flash_attn=kwargs["train"]["use_flash_attn"])


class Roformer(ParamTree):
    def __init__(self, encoder_config, decoder_config, mode="phone", semantic_kmeans_num=10000, codebook_path="pretrain/semantic_codebook.pt",
                 n_spk=1, use_flash_attn=False, **kwargs):
        if "text" in mode:
            raise NotImplementedError("mode 'text' needs BertTokenizer.from_pretrained (network); phone mode is built")
        if "phone" not in mode:
            raise ValueError(f"unknown mode {mode!r}")
        for k in ("hidden_size", "num_attention_heads", "intermediate_size"):
            if _get(encoder_config, k) != _get(decoder_config, k):
                raise NotImplementedError("encoder and decoder widths must match (reference configs/config.yaml:60-83)")
        if _get(encoder_config, "hidden_act", "gelu") != "gelu" or _get(decoder_config, "hidden_act", "gelu") != "gelu":
            raise NotImplementedError("only hidden_act 'gelu' is built")
        # one sinusoid table and one LayerNorm epsilon serve both stacks (lds_lm_cfg): configurations whose stacks differ are refused
        # rather than run with the encoder's values (HF RoFormerConfig defaults: 1536 positions, eps 1e-12)
        for k, dflt in (("max_position_embeddings", 1536), ("layer_norm_eps", 1e-12)):
            if _get(encoder_config, k, dflt) != _get(decoder_config, k, dflt):
                raise NotImplementedError(f"encoder and decoder must agree on {k}")
        cfg = arch.roformer_config(
            n_spk=n_spk, semantic_kmeans_num=semantic_kmeans_num, hidden_size=_get(encoder_config, "hidden_size"),
            num_attention_heads=_get(encoder_config, "num_attention_heads"), intermediate_size=_get(encoder_config, "intermediate_size"),
            encoder_layers=_get(encoder_config, "num_hidden_layers"), decoder_layers=_get(decoder_config, "num_hidden_layers"),
            max_position_embeddings=_get(encoder_config, "max_position_embeddings", 1536),
            layer_norm_eps=float(_get(encoder_config, "layer_norm_eps", 1e-12)))
        super().__init__(arch.roformer_param_shapes(cfg), seed=0)
        self.cfg = cfg
        self.mode, self.n_spk, self.use_flash_attn = mode, n_spk, use_flash_attn
        self.BOS, self.EOS, self.PAD = cfg["text_bos"], cfg["text_eos"], cfg["text_pad"]
        self.num_tones = arch.NUM_TONES
        self.semantic_bos_token_id, self.semantic_eos_token_id, self.semantic_pad_token_id = cfg["sem_bos"], cfg["sem_eos"], cfg["sem_pad"]
        with torch.no_grad():
            for k, v in arch.roformer_init_state(cfg, 0).items():
                self._leaf(k).copy_(torch.from_numpy(v))
        # weight tying like RoFormerForCausalLM: the LM head shares the decoder's word embeddings and the output-only bias
        pred = self.semantic_decoder.cls.predictions
        pred.decoder.weight = self.semantic_decoder.roformer.embeddings.word_embeddings.weight
        pred.decoder.bias = pred.bias
        try:      # reference roformer.py:110-115: seed the token embeddings with the k-means centres when the widths agree
            from cluster import get_cluster_model
            self.quantizer = get_cluster_model(codebook_path)
            centers = self.quantizer.cluster_centers_
            if self.semantic_decoder.roformer.embeddings.word_embeddings.weight.shape[1] == centers.shape[1]:
                self.semantic_decoder.roformer.embeddings.word_embeddings.weight.data[:semantic_kmeans_num] = torch.from_numpy(centers.copy())
        except Exception:
            pass
        self.spk_emb_enabled = n_spk is not None and n_spk > 1
        self._native = None

    def _leaf(self, key):
        node = self
        parts = key.split(".")
        for p in parts[:-1]:
            node = node._modules[p]
        return node._parameters[parts[-1]]

    # parameter changes drop the packed copy (see UNet1DConditionModel)
    def _apply(self, fn, *a, **k):
        self._native = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._native = None
        return super()._load_from_state_dict(*a, **k)

    def native(self):
        if self._native is None:
            self._native = native.LM(self.cfg, dict(self.state_dict()))
        return self._native

    def forward(self, *a, **k):
        raise NotImplementedError("teacher-forced training forward is out of scope for the MI355X inference build")

    @staticmethod
    def _mask_to_lengths(attention_mask):
        """HF padding mask [B,L] (1 = real, 0 = pad) of a RIGHT-padded batch -> int32 [B] real lengths (what the kernels take)"""
        if attention_mask is None:
            return None
        m = attention_mask.to(torch.int64)
        n = m.sum(-1)
        L = m.shape[-1]
        if not bool(((torch.arange(L, device=m.device)[None] < n[:, None]).to(torch.int64) == m).all()) or bool((n < 1).any()):
            raise NotImplementedError("only right-padded batches are built: every attention_mask row must be ones followed by zeros")
        return n.to(torch.int32).contiguous()

    @torch.no_grad()
    def encode(self, phone, tone, spk_id=None, attention_mask=None):
        """encoder_hidden_states [B,L,hidden] of reference roformer.py:196-214 (rows of padded positions are computed but never read)"""
        return self.native().encode(phone, tone, spk_id if self.spk_emb_enabled else None, self._mask_to_lengths(attention_mask))

    @torch.no_grad()
    def generate(self, phone, tone, attention_mask=None, use_cache=None, max_length=1024, do_sample=True, temperature=1.0, top_k=5, top_p=0.8,
                 repetition_penalty=1.2, num_beams=1, no_repeat_ngram_size=0, early_stopping=True, spk_id=None, end_gate_threshold=None,
                 return_logits=False, **kwargs):
        if num_beams != 1 or no_repeat_ngram_size != 0 or end_gate_threshold is not None:
            raise NotImplementedError("beam search, n-gram blocking and the end gate are not built (22_infer_tts.py uses none of them)")
        # top_k as HF's TopKLogitsWarper: None / 0 = no filter (the draw runs over the whole vocabulary), k keeps the k largest scores and every
        # score tied with the k-th (22_infer_tts.py:83-98 passes top_k = 5); more than 64 survivors are not built
        top_k = 0 if top_k is None else int(top_k)
        if do_sample and not (0 <= top_k <= 64):
            raise NotImplementedError("sampling is built for top_k = None / 0 (no filter) or 1 <= top_k <= 64")
        enc_len = self._mask_to_lengths(attention_mask)      # reference roformer.py:209-236: the encoder's mask and the cross-attention's
        if not phone.is_cuda:
            raise RuntimeError("Roformer.generate needs tensors on a HIP device (no CPU fallback)")
        enc = self.native().encode(phone, tone, spk_id if self.spk_emb_enabled else None, enc_len)
        B = enc.shape[0]
        uniforms = torch.rand(max_length - 1, B, device=enc.device) if do_sample else None      # one draw per step and sequence
        toks, logits = self.native().generate(enc, max_length, do_sample, top_k, top_p, temperature, repetition_penalty, uniforms, return_logits, enc_len)
        return (toks, logits) if return_logits else toks
