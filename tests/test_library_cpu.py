"""CPU suite: the C-ABI library builds, loads and exports every symbol include/fishmi.h declares.
No compute entry point is called here (there is no GPU in this container)."""
import ctypes as C
import os
import re

from fish_speech_amd import _lib
from fish_speech_amd.build import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fishmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fmi_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build()
    assert os.path.exists(path)
    lib = C.CDLL(path)
    names = _declared_symbols()
    assert len(names) > 20
    for n in names:
        assert hasattr(lib, n), f"libfishmi.so lacks {n}"


def test_ctypes_binding_covers_the_header():
    assert set(_lib.EXPORTED_SYMBOLS) == set(_declared_symbols())
    lib = _lib.load()
    assert lib.fmi_version() == 1


def test_arena_size_of_s2_pro_shape():
    from fish_speech_amd.dual_ar import DualARConfig

    cfg = DualARConfig(vocab_size=155776, n_layer=36, n_head=32, n_local_heads=8, head_dim=128, dim=2560,
                       intermediate_size=9728, codebook_size=4096, num_codebooks=10, semantic_begin_id=151678,
                       semantic_end_id=155773, im_end_id=151645, max_seq_len=4096, attention_qk_norm=True)
    n = _lib.load().fmi_dualar_arena_bytes(C.byref(cfg.to_c()))
    assert 9.0e9 < n < 9.4e9  # 4.56 B bf16 parameters + live head rows + RoPE tables


def test_bad_config_is_rejected_not_crashing():
    from fish_speech_amd.dual_ar import DualARConfig

    cfg = DualARConfig(vocab_size=100, n_layer=1, n_head=3, n_local_heads=2, head_dim=48, dim=100,
                       intermediate_size=64, codebook_size=10, num_codebooks=4, semantic_begin_id=10,
                       semantic_end_id=20, im_end_id=5)
    assert _lib.load().fmi_dualar_arena_bytes(C.byref(cfg.to_c())) < 0
