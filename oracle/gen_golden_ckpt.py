"""Writes tests/golden/checkpoint_format.json: what the UNMODIFIED reference makes of an S2-style checkpoint's
config.json (BaseModelArgs._from_fish_qwen3_omni, llama.py:99-143) and tensor names
(_remap_fish_qwen3_omni_keys, llama.py:229-246).  Authoring container only:  python -m oracle.gen_golden_ckpt"""
import dataclasses
import json
import os
from collections import OrderedDict

from oracle.refload import add_reference_to_path

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "checkpoint_format.json")

CONFIGS = {
    "s2_like": {
        "model_type": "fish_qwen3_omni", "semantic_start_token_id": 1000, "semantic_end_token_id": 5095,
        "text_config": {"vocab_size": 6000, "n_layer": 3, "n_head": 8, "n_local_heads": 2, "head_dim": 32, "dim": 192,
                        "intermediate_size": 512, "rope_base": 1000000, "norm_eps": 1e-6, "max_seq_len": 4096,
                        "attention_qk_norm": True, "tie_word_embeddings": True},
        "audio_decoder_config": {"vocab_size": 4096, "num_codebooks": 10, "n_layer": 2, "dim": 192, "n_head": 8,
                                 "n_local_heads": 2, "head_dim": 32, "intermediate_size": 512, "attention_qk_norm": False,
                                 "text_dim": 192},
    },
    "defaults": {   # optional keys absent: the reference's fallbacks apply
        "model_type": "fish_qwen3_omni",
        "text_config": {"vocab_size": 3000, "n_layer": 2, "n_head": 4, "dim": 128, "intermediate_size": 256},
        "audio_decoder_config": {"vocab_size": 64, "num_codebooks": 4, "n_layer": 1},
    },
}

KEYS = ["text_model.model.embeddings.weight", "text_model.model.layers.0.attention.wqkv.weight",
        "text_model.model.layers.2.feed_forward.w1.weight", "text_model.model.norm.weight",
        "audio_decoder.codebook_embeddings.weight", "audio_decoder.embeddings.weight", "audio_decoder.layers.1.attention.wo.weight",
        "audio_decoder.norm.weight", "audio_decoder.output.weight", "audio_decoder.layers.0.attention.q_norm.weight",
        "some.other.key"]


def main():
    add_reference_to_path()
    from fish_speech.models.text2semantic.llama import BaseModelArgs, _remap_fish_qwen3_omni_keys

    out = {"configs": {}, "keys": None}
    for name, data in CONFIGS.items():
        args = BaseModelArgs._from_fish_qwen3_omni(data)
        out["configs"][name] = {"config_json": data, "model_args": dataclasses.asdict(args)}
    remapped = _remap_fish_qwen3_omni_keys(OrderedDict((k, i) for i, k in enumerate(KEYS)))
    out["keys"] = {"before": KEYS, "after": list(remapped.keys())}
    untouched = _remap_fish_qwen3_omni_keys(OrderedDict([("layers.0.x", 0), ("fast_output.weight", 1)]))
    out["keys_plain"] = list(untouched.keys())
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
