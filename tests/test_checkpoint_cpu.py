"""Checkpoint format handling against what the UNMODIFIED reference derives from the same
config.json / tensor names (tests/golden/checkpoint_format.json <- oracle/gen_golden_ckpt.py)."""
import json
import os

import pytest

from fish_speech_amd.dual_ar import DualARConfig, remap_fish_qwen3_omni_keys

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_format.json")
FIELDS = ["vocab_size", "n_layer", "n_head", "n_local_heads", "head_dim", "dim", "intermediate_size", "rope_base",
          "norm_eps", "max_seq_len", "attention_qk_norm", "codebook_size", "num_codebooks", "semantic_begin_id",
          "semantic_end_id", "scale_codebook_embeddings", "norm_fastlayer_input", "n_fast_layer", "fast_dim",
          "fast_n_head", "fast_n_local_heads", "fast_head_dim", "fast_intermediate_size", "fast_attention_qk_norm"]


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_fish_qwen3_omni_config_equals_the_reference_model_args(gold):
    for name, c in gold["configs"].items():
        cfg = DualARConfig.from_fish_qwen3_omni(c["config_json"], im_end_id=7)
        for f in FIELDS:
            assert getattr(cfg, f) == c["model_args"][f], (name, f, getattr(cfg, f), c["model_args"][f])
        assert cfg.im_end_id == 7
    # ids injected from the tokenizer win over the config's (llama.py:499-505)
    c = gold["configs"]["s2_like"]["config_json"]
    cfg = DualARConfig.from_fish_qwen3_omni(c, 7, semantic_begin_id=11, semantic_end_id=4106)
    assert (cfg.semantic_begin_id, cfg.semantic_end_id) == (11, 4106)


def test_unsupported_checkpoint_options_are_refused(gold):
    base = gold["configs"]["s2_like"]["config_json"]
    for blk, key in (("text_config", "attention_qkv_bias"), ("text_config", "attention_o_bias"),
                     ("audio_decoder_config", "attention_qkv_bias")):
        bad = json.loads(json.dumps(base))
        bad[blk][key] = True
        with pytest.raises(ValueError):
            DualARConfig.from_fish_qwen3_omni(bad, 7)
    bad = json.loads(json.dumps(base))
    bad["text_config"]["tie_word_embeddings"] = False
    with pytest.raises(ValueError):
        DualARConfig.from_fish_qwen3_omni(bad, 7)


def test_tensor_name_remap_equals_the_reference(gold):
    before, after = gold["keys"]["before"], gold["keys"]["after"]
    got = remap_fish_qwen3_omni_keys({k: i for i, k in enumerate(before)})
    assert list(got.keys()) == after and list(got.values()) == list(range(len(before)))
    plain = {k: 0 for k in gold["keys_plain"]}
    assert list(remap_fish_qwen3_omni_keys(plain).keys()) == gold["keys_plain"]
