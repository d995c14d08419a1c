"""text2semantic RoFormer, numpy fp32 (reference text2semantic/roformer/roformer.py:59-255 driving HF transformers
RoFormerModel / RoFormerForCausalLM + GenerationMixin; pinned to transformers 5.15.0, the version the fixtures were produced
with: tests/golden/make_fixtures.py -> roformer.npz).  `w` = the reference module's state_dict as float32 arrays, `cfg` =
lds.arch.roformer_config (passed in; this package never imports the product).

Encoder input (roformer.py:196-204, called with inputs_embeds): LN_e(LN_e(word[phone] + type[tone]) + spk[spk_id] + type[0]).
Layers are BERT-style post-LayerNorm with rotary position embedding on q, k of the self-attention (not on cross-attention,
rotary_value False).  The decoder is evaluated token by token with a key/value cache exactly as generate() does."""
import math

import numpy as np
from scipy.special import erf

f32 = np.float32


def layer_norm(x, g, b, eps):
    xr = x.astype(np.float64)
    mu = xr.mean(-1, keepdims=True)
    var = xr.var(-1, keepdims=True)
    return (((xr - mu) / np.sqrt(var + eps)) * g + b).astype(f32)


def gelu(x):
    return (0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))).astype(f32)


def linear(x, w, b):
    return (np.matmul(x, w.T) + b).astype(f32)


def rotary(x, table, pos0):
    """x [B,H,L,d]; table [n_pos, d] = [sin | cos]; positions pos0 .. pos0+L-1 (modeling_roformer.py:220-245)"""
    L, d = x.shape[2], x.shape[3]
    sc = table[pos0:pos0 + L]
    sin, cos = np.repeat(sc[:, : d // 2], 2, axis=1), np.repeat(sc[:, d // 2:], 2, axis=1)
    rot = np.stack([-x[..., 1::2], x[..., ::2]], axis=-1).reshape(x.shape)
    return (x * cos + rot * sin).astype(f32)


def heads(x, H):
    B, L, C = x.shape
    return x.reshape(B, L, H, C // H).transpose(0, 2, 1, 3)


def attend(q, k, v, key_len=None):
    """softmax(q k^T / sqrt(d)) v; q [B,H,Lq,d], k/v [B,H,Lk,d] -> [B,Lq,H*d].  key_len [B] = the padding mask of a right-padded batch
    (reference roformer.py:209-236: HF adds a -inf bias to the keys at positions >= key_len[b]; their probability is exactly 0)"""
    s = np.matmul(q, k.transpose(0, 1, 3, 2)).astype(f32) / f32(math.sqrt(q.shape[-1]))
    if key_len is not None:
        masked = np.arange(k.shape[2])[None, :] >= np.asarray(key_len)[:, None]          # [B, Lk]
        s = np.where(masked[:, None, None, :], f32(-np.inf), s).astype(f32)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s).astype(f32)
    p = (p / p.sum(-1, keepdims=True)).astype(f32)
    o = np.matmul(p, v).astype(f32)
    B, H, L, d = o.shape
    return o.transpose(0, 2, 1, 3).reshape(B, L, H * d)


def _self_attention(w, p, cfg, x, table, pos0, cache=None, key_len=None):
    H = cfg["heads"]
    q = rotary(heads(linear(x, w[p + "self.query.weight"], w[p + "self.query.bias"]), H), table, pos0)
    k = rotary(heads(linear(x, w[p + "self.key.weight"], w[p + "self.key.bias"]), H), table, pos0)
    v = heads(linear(x, w[p + "self.value.weight"], w[p + "self.value.bias"]), H)
    if cache is not None:
        if "k" in cache:
            k, v = np.concatenate([cache["k"], k], axis=2), np.concatenate([cache["v"], v], axis=2)
        cache["k"], cache["v"] = k, v
    ctx = attend(q, k, v, key_len)
    return layer_norm(linear(ctx, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + x, w[p + "output.LayerNorm.weight"],
                      w[p + "output.LayerNorm.bias"], cfg["eps"])


def _cross_attention(w, p, cfg, x, enc_kv, enc_len=None):
    q = heads(linear(x, w[p + "self.query.weight"], w[p + "self.query.bias"]), cfg["heads"])
    ctx = attend(q, enc_kv[0], enc_kv[1], enc_len)
    return layer_norm(linear(ctx, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + x, w[p + "output.LayerNorm.weight"],
                      w[p + "output.LayerNorm.bias"], cfg["eps"])


def _ffn(w, p, cfg, x):
    h = gelu(linear(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
    return layer_norm(linear(h, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + x, w[p + "output.LayerNorm.weight"],
                      w[p + "output.LayerNorm.bias"], cfg["eps"])


def encoder_forward(w, cfg, phone, tone, spk_id=None, enc_len=None):
    """phone, tone [B,L] int; spk_id [B,L] int or None; enc_len [B] = real positions of each right-padded row or None
    -> encoder_hidden_states [B,L,hidden] (rows at padded positions are computed like HF does; nothing reads them)"""
    p = "text_encoder."
    g, b = w[p + "embeddings.LayerNorm.weight"], w[p + "embeddings.LayerNorm.bias"]
    e = layer_norm(w[p + "embeddings.word_embeddings.weight"][phone] + w[p + "embeddings.token_type_embeddings.weight"][tone], g, b, cfg["eps"])
    if spk_id is not None and "spk_emb.weight" in w:
        e = (e + w["spk_emb.weight"][spk_id]).astype(f32)
    x = layer_norm(e + w[p + "embeddings.token_type_embeddings.weight"][0], g, b, cfg["eps"])       # embeddings(inputs_embeds=...)
    table = w[p + "encoder.embed_positions.weight"]
    for i in range(cfg["enc_layers"]):
        q = p + f"encoder.layer.{i}."
        x = _self_attention(w, q + "attention.", cfg, x, table, 0, key_len=enc_len)
        x = _ffn(w, q, cfg, x)
    return x


def cross_kv(w, cfg, enc):
    out = []
    for i in range(cfg["dec_layers"]):
        p = f"semantic_decoder.roformer.encoder.layer.{i}.crossattention."
        out.append((heads(linear(enc, w[p + "self.key.weight"], w[p + "self.key.bias"]), cfg["heads"]),
                    heads(linear(enc, w[p + "self.value.weight"], w[p + "self.value.bias"]), cfg["heads"])))
    return out


def decoder_step(w, cfg, tok, pos, caches, enc_kv, enc_len=None):
    """one incremental decoder evaluation: tok [B] at position `pos` -> logits [B, vocab] (modeling_roformer.py:883-950)"""
    p = "semantic_decoder.roformer."
    x = layer_norm(w[p + "embeddings.word_embeddings.weight"][tok][:, None] + w[p + "embeddings.token_type_embeddings.weight"][0],
                   w[p + "embeddings.LayerNorm.weight"], w[p + "embeddings.LayerNorm.bias"], cfg["eps"])
    table = w[p + "encoder.embed_positions.weight"]
    for i in range(cfg["dec_layers"]):
        q = p + f"encoder.layer.{i}."
        x = _self_attention(w, q + "attention.", cfg, x, table, pos, caches[i])
        x = _cross_attention(w, q + "crossattention.", cfg, x, enc_kv[i], enc_len)
        x = _ffn(w, q, cfg, x)
    c = "semantic_decoder.cls.predictions."
    t = layer_norm(gelu(linear(x, w[c + "transform.dense.weight"], w[c + "transform.dense.bias"])), w[c + "transform.LayerNorm.weight"],
                   w[c + "transform.LayerNorm.bias"], cfg["eps"])
    return linear(t, w[c + "decoder.weight"], w[c + "decoder.bias"])[:, 0]


def pick_token(logits, do_sample, top_k, u):
    """greedy argmax, or TopKLogitsWarper(top_k) -> softmax -> inverse-CDF draw with the uniform u (vocabulary order)"""
    if not do_sample:
        return int(np.argmax(logits))
    kth = np.sort(logits)[-top_k]
    s = np.where(logits < kth, -np.inf, logits).astype(f32)
    pr = np.exp(s - s.max()).astype(f32)
    pr = (pr / pr.sum()).astype(f32)
    c = np.cumsum(pr.astype(f32), dtype=f32)
    return int(min(np.searchsorted(c, f32(u), side="right"), len(c) - 1))


def pick_token_hf(logits, history, top_k, top_p, temperature, repetition_penalty, u):
    """One draw the way HF's GenerationMixin._sample forms it (transformers logits_process.py): RepetitionPenaltyLogitsProcessor over the distinct
    tokens of `history` -> TemperatureLogitsWarper -> TopKLogitsWarper (top_k None / 0: no filter; else scores BELOW the k-th largest are removed, so
    ties of the k-th survive) -> TopPLogitsWarper (ascending order, drop while the cumulative probability stays <= 1 - top_p, keep the largest) ->
    softmax -> inverse-CDF draw in vocabulary order with the uniform u (float32 sequential sums, like the kernels)."""
    s = np.asarray(logits, dtype=f32).copy()
    if repetition_penalty != 1.0:
        for t in sorted(set(int(t) for t in history)):
            s[t] = s[t] * f32(repetition_penalty) if s[t] < 0 else s[t] / f32(repetition_penalty)
    if temperature != 1.0:
        s = (s * f32(1.0 / temperature)).astype(f32)
    if top_k:
        # the survivors in descending order of score (equal scores: lower id first -- HF's own order among exactly equal scores is whatever
        # torch.sort returns), softmax over them, the nucleus cut from the tail, the draw in vocabulary order
        kth = np.sort(s)[-int(top_k)]
        ids = np.nonzero(~(s < kth))[0]
        ids = ids[np.lexsort((ids, -s[ids]))]
        p = np.exp(s[ids] - s[ids][0]).astype(f32)
        tot = f32(0)
        for v in p:
            tot = f32(tot + v)
        kept = len(ids)
        if top_p < 1.0:
            tail, cut = f32(0), kept
            for q in range(kept - 1, 0, -1):
                tail = f32(tail + f32(p[q] / tot))
                if tail <= f32(1.0) - f32(top_p):
                    cut = q
                else:
                    break
            kept = cut
            tot = f32(0)
            for v in p[:kept]:
                tot = f32(tot + v)
        ids, p = ids[:kept], (p[:kept] / tot).astype(f32)
        o = np.argsort(ids)
        c = np.cumsum(p[o], dtype=f32)
        return int(ids[o][min(int(np.searchsorted(c, f32(u), side="right")), kept - 1)])
    # no top-k filter: the whole vocabulary; the nucleus cut by VALUE (probabilities equal to the last dropped one go with it)
    p = np.exp(s - s.max()).astype(f32)
    p = (p / np.cumsum(p, dtype=f32)[-1]).astype(f32)
    if top_p < 1.0:
        order = np.argsort(p, kind="stable")                      # ascending
        cum = np.cumsum(p[order].astype(np.float64))
        drop = cum <= 1.0 - top_p
        drop[-1] = False
        if drop.any():
            cutv = p[order][drop].max()
            if cutv < p.max():
                p = np.where(p <= cutv, f32(0), p)
    z = np.cumsum(p, dtype=f32)[-1]
    c = np.cumsum((p / z).astype(f32), dtype=f32)
    k = int(np.searchsorted(c, f32(u), side="right"))
    if k >= len(c):
        k = int(np.nonzero(p > 0)[0][-1])
    return k


def generate(w, cfg, enc, max_length, do_sample=False, top_k=5, uniforms=None, enc_len=None):
    """GenerationMixin greedy / sampling loop as Roformer.generate drives it (roformer.py:179-240): starts from BOS, stops when every
    sequence has produced EOS or at max_length, finished sequences are padded.  Returns (tokens [B, n], logits [n-1, B, vocab])."""
    B = enc.shape[0]
    kv = cross_kv(w, cfg, enc)
    caches = [dict() for _ in range(cfg["dec_layers"])]
    seq = np.full((B, 1), cfg["sem_bos"], dtype=np.int64)
    unfinished = np.ones(B, dtype=bool)
    all_logits = []
    step = 0
    while seq.shape[1] < max_length and unfinished.any():
        lg = decoder_step(w, cfg, seq[:, -1], seq.shape[1] - 1, caches, kv, enc_len)
        all_logits.append(lg)
        nxt = np.array([pick_token(lg[b], do_sample, top_k, None if uniforms is None else uniforms[step, b]) for b in range(B)], dtype=np.int64)
        nxt = np.where(unfinished, nxt, cfg["sem_pad"])
        seq = np.concatenate([seq, nxt[:, None]], axis=1)
        unfinished &= nxt != cfg["sem_eos"]
        step += 1
    return seq, np.stack(all_logits)
