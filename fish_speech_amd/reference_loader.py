"""Reference-audio side of the inference engine: `ReferenceLoader` (fish_speech/inference_engine/reference_loader.py:23-99
-- references by id from `references/<id>/` or by content hash, with the two in-memory caches) and `VQManager`
(fish_speech/inference_engine/vq_manager.py:16-53 -- `encode_reference`, `decode_vq_tokens`) over a `MiDAC`.

The DATA-PATH methods (`load_by_id`, `load_by_hash`, `load_audio`, `encode_reference`, `decode_vq_tokens`) keep upstream's
names, arguments, cache keys, validation and error types; NOT mirrored -- a stated gap, see INTEGRATION.md "What is not
mirrored" -- are the three reference-library management methods upstream's class also defines
(`list_reference_ids`, `add_reference`, `delete_reference`, reference_loader.py:155-260: directory listing / copy /
delete behind the HTTP server's routes, control plane per SURVEY.md 8).  What else differs is audio I/O:
torchaudio is not in this image, so `load_audio` reads WAV (scipy) and AIFF PCM (standard library) bytes / files and
resamples with a polyphase filter (host plumbing outside the hot path; mp3 / flac / ... references raise a clear error
instead of being mis-decoded).  `engine.StreamingTTSEngine` inherits both, like `TTSInferenceEngine` does."""
from __future__ import annotations

import io
import re
from hashlib import sha256
from pathlib import Path
from typing import Callable, List, Literal, Tuple

import numpy as np
import torch

AUDIO_EXTENSIONS = {".mp3", ".wav", ".flac", ".ogg", ".m4a", ".wma", ".aac", ".aiff", ".aif", ".aifc"}   # utils/file.py:8-19
_ID_PATTERN = re.compile(r"^[a-zA-Z0-9\-_ ]+$")          # reference_loader.py:20


def list_files(path, extensions=AUDIO_EXTENSIONS, recursive=False, sort=True) -> List[Path]:
    """utils/file.py:57-90."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Directory {path} does not exist.")
    files = [f for f in (path.rglob("*") if recursive else path.glob("*")) if f.is_file() and f.suffix in extensions]
    return sorted(files) if sort else files


def audio_to_bytes(file_path):
    """utils/file.py:41-46."""
    if not file_path or not Path(file_path).exists():
        return None
    with open(file_path, "rb") as f:
        return f.read()


def read_ref_text(ref_text):
    """utils/file.py:49-54."""
    path = Path(ref_text)
    if path.exists() and path.is_file():
        return path.read_text(encoding="utf-8")
    return ref_text


def _read_aiff(src):
    """AIFF / AIFF-C with uncompressed big-endian PCM through the standard library (reference_loader.py:133-141 accepts any
    container torchaudio decodes; AUDIO_EXTENSIONS lists .aiff / .aif / .aifc).  -> (sample_rate, int array (frames[, ch]))
    or None if `src` is not such a file."""
    import warnings

    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            import aifc
        if hasattr(src, "seek"):
            src.seek(0)
        with aifc.open(src, "rb") as f:
            if f.getcomptype() not in (b"NONE", b"sowt"):
                return None
            nch, width, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
            raw = f.readframes(n)
            little = f.getcomptype() == b"sowt"
    except Exception:  # noqa: BLE001 -- not an AIFF file (or a flavour aifc cannot read)
        return None
    if width == 3:      # 24-bit: widen to int32 (value << 8), so that the integer scaling below applies
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3)
        b = b[:, ::-1] if not little else b
        data = ((b[:, 0].astype(np.int32) | (b[:, 1].astype(np.int32) << 8) | (b[:, 2].astype(np.int8).astype(np.int32) << 16)) << 8)
    elif width in (1, 2, 4):
        dt = np.dtype({1: "i1", 2: "i2", 4: "i4"}[width]).newbyteorder("<" if little else ">")
        data = np.frombuffer(raw, dtype=dt).astype({1: np.int8, 2: np.int16, 4: np.int32}[width])
    else:
        return None
    return int(sr), (data.reshape(-1, nch) if nch > 1 else data)


class VQManager:
    """vq_manager.py:9-53 on a MiDAC (`self.decoder_model`)."""

    decoder_model = None
    load_audio: Callable

    def decode_vq_tokens(self, codes):
        return self.decoder_model.from_indices(codes[None])[0].squeeze()        # vq_manager.py:20

    def encode_reference(self, reference_audio, enable_reference_audio):
        if not (enable_reference_audio and reference_audio is not None):
            return None
        sample_rate = self.decoder_model.sample_rate
        content = self.load_audio(reference_audio, sample_rate)
        audios = torch.from_numpy(content).to(self.decoder_model.device)[None, None, :]
        audio_lengths = torch.tensor([audios.shape[2]], device=self.decoder_model.device, dtype=torch.long)
        return self.decoder_model.encode(audios, audio_lengths)[0][0]          # vq_manager.py:44


class ReferenceLoader:
    """reference_loader.py:23-260.  `references_root` (default "references", relative to the working directory like
    upstream) is the only addition: tests point it at a temporary directory."""

    references_root = Path("references")

    def __init__(self) -> None:
        self.ref_by_id: dict = {}
        self.ref_by_hash: dict = {}

    @staticmethod
    def _validate_id(id: str) -> None:
        if not _ID_PATTERN.match(id) or len(id) > 255:
            raise ValueError("Reference ID contains invalid characters or is too long. "
                             "Only alphanumeric, hyphens, underscores, and spaces are allowed.")

    def load_by_id(self, id: str, use_cache: Literal["on", "off"]) -> Tuple:
        self._validate_id(id)
        ref_folder = Path(self.references_root) / id
        ref_folder.mkdir(parents=True, exist_ok=True)
        ref_audios = list_files(ref_folder, AUDIO_EXTENSIONS, recursive=True, sort=False)
        if use_cache == "off" or id not in self.ref_by_id:
            prompt_tokens = [self.encode_reference(reference_audio=audio_to_bytes(str(a)), enable_reference_audio=True)
                             for a in ref_audios]
            prompt_texts = [read_ref_text(str(a.with_suffix(".lab"))) for a in ref_audios]
            self.ref_by_id[id] = (prompt_tokens, prompt_texts)
        else:
            prompt_tokens, prompt_texts = self.ref_by_id[id]
        return prompt_tokens, prompt_texts

    def load_by_hash(self, references: list, use_cache: Literal["on", "off"]) -> Tuple:
        """references: objects with `.audio` (bytes) and `.text` (ServeReferenceAudio)."""
        hashes = [sha256(ref.audio).hexdigest() for ref in references]
        prompt_tokens, prompt_texts = [], []
        for h, ref in zip(hashes, references):
            if use_cache == "off" or h not in self.ref_by_hash:
                prompt_tokens.append(self.encode_reference(reference_audio=ref.audio, enable_reference_audio=True))
                prompt_texts.append(ref.text)
                self.ref_by_hash[h] = (prompt_tokens[-1], ref.text)
            else:
                tok, text = self.ref_by_hash[h]
                prompt_tokens.append(tok)
                prompt_texts.append(text)
        return prompt_tokens, prompt_texts

    def load_audio(self, reference_audio, sr: int) -> np.ndarray:
        """bytes or a path -> mono float32 samples at `sr` (reference_loader.py:125-146, torchaudio replaced)."""
        from scipy.io import wavfile
        from scipy.signal import resample_poly

        if isinstance(reference_audio, (bytes, bytearray)) or len(reference_audio) > 255 or not Path(reference_audio).exists():
            src = io.BytesIO(reference_audio if isinstance(reference_audio, (bytes, bytearray)) else bytes(reference_audio))
        else:
            src = str(reference_audio)
        try:
            original_sr, data = wavfile.read(src)
        except ValueError as e:
            aiff = _read_aiff(src)          # the one other container the standard library decodes (.aiff / .aif / .aifc PCM)
            if aiff is None:
                raise ValueError("only RIFF/WAVE and AIFF (PCM) reference audio can be decoded without torchaudio "
                                 f"(scipy.io.wavfile: {e})") from e
            original_sr, data = aiff
        x = data.astype(np.float32)
        if data.dtype == np.uint8:           # 8-bit WAV PCM is unsigned: (x - 128) / 128, as torchaudio / soundfile decode it
            x = (x - 128.0) / 128.0
        elif np.issubdtype(data.dtype, np.integer):
            x /= float(2 ** (8 * data.dtype.itemsize - 1))   # 1 / 32768 for int16, not 1 / iinfo.max
        if x.ndim == 2:
            x = x.mean(axis=1)
        if original_sr != sr:
            g = int(np.gcd(int(original_sr), int(sr)))
            x = resample_poly(x, sr // g, original_sr // g).astype(np.float32)
        return np.ascontiguousarray(x.squeeze(), dtype=np.float32)
