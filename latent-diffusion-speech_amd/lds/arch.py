"""Architecture enumeration for the hot path: parameter names and shapes.

The denoiser is the reference's ``UNet1DConditionModel`` as configured by
``Unit2Mel`` (reference diffusion/unit2mel.py:62-71): four down blocks
(3x CrossAttnDown + Down), a cross-attn mid block, four up blocks
(Up + 3x CrossAttnUp), ``layers_per_block`` resnets per down block (+1 per up
block), 8 GroupNorm groups, 8 attention heads, scale/shift time conditioning.
Key names follow the reference modules' ``state_dict`` (reference
diffusion/unet1d/unet_1d_condition.py:151-607, unet_1d_blocks.py:861-1096,
1985-2206, resnet.py:461-589, transformer_1d.py:41-224, attention.py:26-128)
so that a reference checkpoint's ``ckpt['model']`` loads unchanged.

The vocoder decoder is the reference's HiFi-VAEGAN ``Generator`` (reference
encoder/hifi_vaegan/modules/models.py:224-247) whose checkpoint carries
weight-norm pairs ``weight_g`` / ``weight_v``.

tests/golden/manifest_*.json hold the key->shape lists captured from the
reference modules; tests/test_arch.py checks this enumeration against them.
"""
from collections import OrderedDict

TIME_EMBED_DIM_MULT = 4


def unet_config(out_dims=80, n_hidden=256, block_out_channels=(256, 384, 512, 512),
                n_layers=2, n_heads=8, norm_groups=8):
    boc = tuple(int(c) for c in block_out_channels)
    return dict(
        in_channels=out_dims + n_hidden, out_channels=out_dims, x_channels=out_dims,
        cond_channels=n_hidden, block_out_channels=boc, layers_per_block=int(n_layers),
        heads=int(n_heads), groups=int(norm_groups), time_embed_dim=boc[0] * TIME_EMBED_DIM_MULT,
        time_proj_dim=boc[0])


def _resnet(d, p, cin, cout, temb):
    d[p + "norm1.weight"] = (cin,)
    d[p + "norm1.bias"] = (cin,)
    d[p + "conv1.weight"] = (cout, cin, 3)
    d[p + "conv1.bias"] = (cout,)
    d[p + "time_emb_proj.weight"] = (2 * cout, temb)
    d[p + "time_emb_proj.bias"] = (2 * cout,)
    d[p + "norm2.weight"] = (cout,)
    d[p + "norm2.bias"] = (cout,)
    d[p + "conv2.weight"] = (cout, cout, 3)
    d[p + "conv2.bias"] = (cout,)
    if cin != cout:
        d[p + "conv_shortcut.weight"] = (cout, cin, 1)
        d[p + "conv_shortcut.bias"] = (cout,)


def _transformer(d, p, c):
    d[p + "norm.weight"] = (c,)
    d[p + "norm.bias"] = (c,)
    d[p + "proj_in.weight"] = (c, c, 1)
    d[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    for i in (1, 2):
        d[b + f"norm{i}.weight"] = (c,)
        d[b + f"norm{i}.bias"] = (c,)
        d[b + f"attn{i}.to_q.weight"] = (c, c)
        d[b + f"attn{i}.to_k.weight"] = (c, c)
        d[b + f"attn{i}.to_v.weight"] = (c, c)
        d[b + f"attn{i}.to_out.0.weight"] = (c, c)
        d[b + f"attn{i}.to_out.0.bias"] = (c,)
    d[b + "norm3.weight"] = (c,)
    d[b + "norm3.bias"] = (c,)
    d[b + "ff.net.0.proj.weight"] = (8 * c, c)
    d[b + "ff.net.0.proj.bias"] = (8 * c,)
    d[b + "ff.net.2.weight"] = (c, 4 * c)
    d[b + "ff.net.2.bias"] = (c,)
    d[p + "proj_out.weight"] = (c, c, 1)
    d[p + "proj_out.bias"] = (c,)


def unet_blocks(cfg):
    """Structural description shared by the parameter enumeration, the numpy
    oracle and the native plan builder: a list of dicts in execution order."""
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    nb = len(boc)
    down = []
    skip_ch = [boc[0]]
    cout = boc[0]
    for i, c in enumerate(boc):
        cin, cout = cout, c
        last = i == nb - 1
        res = [(cin if j == 0 else cout, cout) for j in range(L)]
        down.append(dict(kind="down", idx=i, attn=not last, resnets=res, downsample=not last, ch=cout))
        skip_ch += [cout] * L
        if not last:
            skip_ch.append(cout)
    mid = dict(kind="mid", ch=boc[-1])
    rev = list(reversed(boc))
    up = []
    stack = list(skip_ch)
    cout = rev[0]
    for i, c in enumerate(rev):
        prev = cout
        cout = c
        last = i == nb - 1
        res = []
        for j in range(L + 1):
            sk = stack.pop()
            hin = prev if j == 0 else cout
            res.append((hin, sk, cout))  # (hidden in, skip in, out)
        up.append(dict(kind="up", idx=i, attn=i != 0, resnets=res, upsample=not last, ch=cout))
    return down, mid, up


def unet_param_shapes(cfg):
    d = OrderedDict()
    boc = cfg["block_out_channels"]
    temb = cfg["time_embed_dim"]
    d["conv_in.weight"] = (boc[0], cfg["in_channels"], 3)
    d["conv_in.bias"] = (boc[0],)
    d["time_embedding.linear_1.weight"] = (temb, cfg["time_proj_dim"])
    d["time_embedding.linear_1.bias"] = (temb,)
    d["time_embedding.linear_2.weight"] = (temb, temb)
    d["time_embedding.linear_2.bias"] = (temb,)
    down, mid, up = unet_blocks(cfg)
    for blk in down:
        p = f"down_blocks.{blk['idx']}."
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                _transformer(d, p + f"attentions.{j}.", blk["ch"])
        for j, (cin, cout) in enumerate(blk["resnets"]):
            _resnet(d, p + f"resnets.{j}.", cin, cout, temb)
        if blk["downsample"]:
            d[p + "downsamplers.0.conv.weight"] = (blk["ch"], blk["ch"], 3)
            d[p + "downsamplers.0.conv.bias"] = (blk["ch"],)
    for blk in up:
        p = f"up_blocks.{blk['idx']}."
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                _transformer(d, p + f"attentions.{j}.", blk["ch"])
        for j, (hin, sk, cout) in enumerate(blk["resnets"]):
            _resnet(d, p + f"resnets.{j}.", hin + sk, cout, temb)
        if blk["upsample"]:
            d[p + "upsamplers.0.conv.weight"] = (blk["ch"], blk["ch"], 3)
            d[p + "upsamplers.0.conv.bias"] = (blk["ch"],)
    c = mid["ch"]
    _transformer(d, "mid_block.attentions.0.", c)
    _resnet(d, "mid_block.resnets.0.", c, c, temb)
    _resnet(d, "mid_block.resnets.1.", c, c, temb)
    d["conv_norm_out.weight"] = (boc[0],)
    d["conv_norm_out.bias"] = (boc[0],)
    d["conv_out.weight"] = (cfg["out_channels"], boc[0], 3)
    d["conv_out.bias"] = (cfg["out_channels"],)
    return d


DIFFUSION_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2")


def unit2mel_param_shapes(input_channel, n_spk, cfg, timesteps=1000):
    """Full ``Unit2Mel.state_dict()`` key->shape (reference diffusion/unit2mel.py:52-71,
    diffusion/diffusion.py:64-85)."""
    d = OrderedDict()
    nh = cfg["cond_channels"]
    d["unit_embed.weight"] = (nh, input_channel)
    d["unit_embed.bias"] = (nh,)
    if n_spk is not None and n_spk > 1:
        d["spk_embed.weight"] = (n_spk, nh)
    for b in DIFFUSION_BUFFERS:
        d["decoder." + b] = (timesteps,)
    d["decoder.spec_min"] = (1, 1, 1)
    d["decoder.spec_max"] = (1, 1, 1)
    for k, s in unet_param_shapes(cfg).items():
        d["decoder.denoise_fn." + k] = s
    return d


# Synthetic HiFi-GAN-V1-shaped vocoder config (the real one lives only in
# the absent checkpoint pretrain/hifi-vaegan/decoder.pth["config"]).
SYNTHETIC_VOCODER_H = dict(
    sampling_rate=44100, hop_size=512, inter_channels=80, resblock="1",
    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4, 4],
    upsample_initial_channel=512)


def generator_param_shapes(h):
    """``Generator.state_dict()`` before remove_weight_norm (reference
    encoder/hifi_vaegan/modules/models.py:224-247, 161-222)."""
    d = OrderedDict()

    def wn(p, wshape, nbias):
        d[p + "bias"] = (nbias,)
        d[p + "weight_g"] = (wshape[0], 1, 1)
        d[p + "weight_v"] = tuple(wshape)

    c0 = h["upsample_initial_channel"]
    wn("conv_pre.", (c0, h["inter_channels"], 7), c0)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        wn(f"ups.{i}.", (c0 // 2 ** i, c0 // 2 ** (i + 1), k), c0 // 2 ** (i + 1))
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i in range(len(h["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, (k, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            if h["resblock"] == "1":
                for m in range(len(dil)):
                    wn(p + f"convs1.{m}.", (ch, ch, k), ch)
                for m in range(len(dil)):
                    wn(p + f"convs2.{m}.", (ch, ch, k), ch)
            else:
                for m in range(len(dil)):
                    wn(p + f"convs.{m}.", (ch, ch, k), ch)
    wn("conv_post.", (1, ch, 7), 1)
    return d


def get_encoder_out_channels(encoder):
    """Reference tools/tools.py:257-264 (`get_encdoer_out_channels`)."""
    table = {"whisper_large_v3": 1280, "contentvec768l12": 768, "xlsr_53_56k": 1024}
    if encoder in table:
        return table[encoder]
    raise ValueError(f"[x] Unknown encoder: {encoder}")


# ---- text2semantic RoFormer (reference text2semantic/roformer/roformer.py:8-160; HF transformers RoFormerModel /
# RoFormerForCausalLM): phone-mode encoder (4 layers) + causal decoder with cross-attention (1 layer) --------------------
PHONE_SYMBOLS = 108      # len(symbols) in reference text/symbols.py:1-33 (pad + sorted phone set + punctuation)
NUM_TONES = 11           # reference text/symbols.py:36 (6 zh + 1 ja + 4 en)


def roformer_config(n_spk=323, semantic_kmeans_num=4096, hidden_size=256, num_attention_heads=8, intermediate_size=512,
                    encoder_layers=4, decoder_layers=1, max_position_embeddings=3072, layer_norm_eps=1e-12):
    """Shapes of `get_model(n_spk, **config['text2semantic'])` in phone mode (reference configs/config.yaml:56-83)."""
    return dict(n_spk=n_spk, semantic_kmeans_num=semantic_kmeans_num, hidden=hidden_size, heads=num_attention_heads,
                inter=intermediate_size, enc_layers=encoder_layers, dec_layers=decoder_layers, max_pos=max_position_embeddings,
                eps=float(layer_norm_eps), text_vocab=PHONE_SYMBOLS + 3, type_vocab=NUM_TONES + 1,
                text_bos=PHONE_SYMBOLS, text_eos=PHONE_SYMBOLS + 1, text_pad=PHONE_SYMBOLS + 2,
                sem_vocab=semantic_kmeans_num + 3, sem_bos=semantic_kmeans_num, sem_eos=semantic_kmeans_num + 1,
                sem_pad=semantic_kmeans_num + 2)


def _roformer_attention(d, p, h):
    for n in ("query", "key", "value"):
        d[p + f"self.{n}.weight"] = (h, h)
        d[p + f"self.{n}.bias"] = (h,)
    d[p + "output.dense.weight"] = (h, h)
    d[p + "output.dense.bias"] = (h,)
    d[p + "output.LayerNorm.weight"] = (h,)
    d[p + "output.LayerNorm.bias"] = (h,)


def _roformer_stack(d, p, cfg, vocab, type_vocab, layers, cross):
    h, it = cfg["hidden"], cfg["inter"]
    d[p + "embeddings.word_embeddings.weight"] = (vocab, h)
    d[p + "embeddings.token_type_embeddings.weight"] = (type_vocab, h)
    d[p + "embeddings.LayerNorm.weight"] = (h,)
    d[p + "embeddings.LayerNorm.bias"] = (h,)
    d[p + "encoder.embed_positions.weight"] = (cfg["max_pos"], h // cfg["heads"])
    for i in range(layers):
        q = p + f"encoder.layer.{i}."
        _roformer_attention(d, q + "attention.", h)
        if cross:
            _roformer_attention(d, q + "crossattention.", h)
        d[q + "intermediate.dense.weight"] = (it, h)
        d[q + "intermediate.dense.bias"] = (it,)
        d[q + "output.dense.weight"] = (h, it)
        d[q + "output.dense.bias"] = (h,)
        d[q + "output.LayerNorm.weight"] = (h,)
        d[q + "output.LayerNorm.bias"] = (h,)


def roformer_param_shapes(cfg):
    """`Roformer.state_dict()` key -> shape (reference roformer.py:59-125 over HF RoFormerModel / RoFormerForCausalLM).
    `*.embed_positions.weight` is the fixed sinusoid table, `cls.predictions.decoder.{weight,bias}` are tied to the word
    embeddings / `cls.predictions.bias`."""
    d = OrderedDict()
    h = cfg["hidden"]
    _roformer_stack(d, "text_encoder.", cfg, cfg["text_vocab"], cfg["type_vocab"], cfg["enc_layers"], False)
    _roformer_stack(d, "semantic_decoder.roformer.", cfg, cfg["sem_vocab"], 1, cfg["dec_layers"], True)
    d["semantic_decoder.cls.predictions.bias"] = (cfg["sem_vocab"],)
    d["semantic_decoder.cls.predictions.transform.dense.weight"] = (h, h)
    d["semantic_decoder.cls.predictions.transform.dense.bias"] = (h,)
    d["semantic_decoder.cls.predictions.transform.LayerNorm.weight"] = (h,)
    d["semantic_decoder.cls.predictions.transform.LayerNorm.bias"] = (h,)
    d["semantic_decoder.cls.predictions.decoder.weight"] = (cfg["sem_vocab"], h)
    d["semantic_decoder.cls.predictions.decoder.bias"] = (cfg["sem_vocab"],)
    if cfg["n_spk"] is not None and cfg["n_spk"] > 1:
        d["spk_emb.weight"] = (cfg["n_spk"] + 1, h)
    return d


ROFORMER_TIED = {"semantic_decoder.cls.predictions.decoder.weight": "semantic_decoder.roformer.embeddings.word_embeddings.weight",
                 "semantic_decoder.cls.predictions.decoder.bias": "semantic_decoder.cls.predictions.bias"}


def roformer_sinusoid_table(n_pos, dim):
    """HF RoFormerSinusoidalPositionalEmbedding.create_weight: [sin(pos * w_i) | cos(pos * w_i)], w_i = 10000^(-2i/dim),
    evaluated in float64 and rounded to fp32 once."""
    import numpy as np
    pos = np.arange(n_pos, dtype=np.float64)[:, None]
    inv = 1.0 / np.power(10000.0, 2.0 * np.arange(dim // 2, dtype=np.float64) / dim)
    ang = pos * inv[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(np.float32)


def roformer_init_state(cfg, seed=0, init_weights=None):
    """Build-owned seeded weights for the LM (no checkpoint ships): per-key integer RNG, tied tensors made identical, the
    position tables computed.  `init_weights` = the lds.init_weights module when this file is loaded outside the package."""
    if init_weights is None:
        from . import init_weights
    import numpy as np
    shapes = roformer_param_shapes(cfg)
    st = init_weights.init_state(shapes, seed)
    for k in shapes:
        if k.endswith("embed_positions.weight"):
            st[k] = roformer_sinusoid_table(*shapes[k])
        elif "LayerNorm.weight" in k:
            st[k] = init_weights.uniform(k, shapes[k], seed, 0.8, 1.2)
        elif "embeddings.weight" in k or k == "spk_emb.weight":
            st[k] = init_weights.uniform(k, shapes[k], seed, -1.0, 1.0)
    for dst, src in ROFORMER_TIED.items():
        st[dst] = st[src]
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in st.items()}
