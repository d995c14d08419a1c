"""text2semantic module shell without a GPU: the reference's constructor / generate surface (reference
text2semantic/roformer/roformer.py:8-58,59-77,179-197), state_dict keys, tying, and the loud failures."""
import inspect
import json
import os

import pytest
import torch
import yaml

from conftest import GOLDEN


@pytest.fixture(scope="module")
def lm():
    from text2semantic.utils import get_language_model
    return get_language_model(**yaml.safe_load(open(os.path.join(GOLDEN, "config_lm_like_reference.yaml"))))


def test_roformer_api_surface(lm):
    from text2semantic.roformer.roformer import Roformer, get_model
    g = inspect.signature(Roformer.generate)
    assert list(g.parameters)[1:17] == ["phone", "tone", "attention_mask", "use_cache", "max_length", "do_sample", "temperature", "top_k", "top_p",
                                        "repetition_penalty", "num_beams", "no_repeat_ngram_size", "early_stopping", "spk_id", "end_gate_threshold",
                                        "return_logits"]
    assert g.parameters["top_p"].default == 0.8 and g.parameters["repetition_penalty"].default == 1.2 and g.parameters["max_length"].default == 1024
    assert list(inspect.signature(get_model).parameters) == ["n_spk", "kwargs"]
    i = inspect.signature(Roformer.__init__)
    assert list(i.parameters)[1:8] == ["encoder_config", "decoder_config", "mode", "semantic_kmeans_num", "codebook_path", "n_spk", "use_flash_attn"]
    assert (lm.BOS, lm.EOS, lm.PAD, lm.num_tones) == (108, 109, 110, 11)
    assert (lm.semantic_bos_token_id, lm.semantic_eos_token_id, lm.semantic_pad_token_id) == (4096, 4097, 4098)
    man = json.load(open(os.path.join(GOLDEN, "manifest_roformer.json")))
    sd = lm.state_dict()
    assert set(sd) == set(man) and all(list(sd[k].shape) == man[k] for k in man)
    assert sd["semantic_decoder.cls.predictions.decoder.bias"].data_ptr() == sd["semantic_decoder.cls.predictions.bias"].data_ptr()
    # a reference-format checkpoint round-trips through load_state_dict
    lm.load_state_dict({k: v.clone() for k, v in sd.items()})


def test_roformer_loud_failures(lm):
    from text2semantic.roformer.roformer import Roformer
    from text2semantic.utils import get_language_model
    ph = torch.ones(1, 4, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lm.generate(ph, ph, max_length=8)
    with pytest.raises(NotImplementedError):
        lm.generate(ph, ph, num_beams=4)
    with pytest.raises(NotImplementedError, match="right-padded"):      # only ones-then-zeros masks are built
        lm.generate(ph, ph, attention_mask=torch.tensor([[1, 0, 1, 1]]))
    with pytest.raises(NotImplementedError, match="top_k"):
        lm.generate(ph, ph, top_k=65)      # (None / 0 = no filter and 1 .. 64 are built)
    assert lm._mask_to_lengths(torch.tensor([[1, 1, 1, 0], [1, 1, 1, 1]])).tolist() == [3, 4]
    with pytest.raises(NotImplementedError):
        lm(ph, ph, ph)
    cfg = yaml.safe_load(open(os.path.join(GOLDEN, "config_lm_like_reference.yaml")))
    cfg["text2semantic"]["model"]["mode"] = "text"
    with pytest.raises(NotImplementedError):
        get_language_model(**cfg)
    cfg["text2semantic"]["model"]["type"] = "gpt"
    with pytest.raises(ValueError):
        get_language_model(**cfg)
