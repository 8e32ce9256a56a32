"""whisper_amd — an MI355X-native (gfx950) inference path for OpenAI Whisper that keeps the reference's public
API: `load_model`, `model.transcribe`, `decode` / `DecodingTask`, `log_mel_spectrogram`.

    import whisper_amd as whisper
    model = whisper.load_model("large-v3")           # or a path to a {"dims", "model_state_dict"} checkpoint
    print(model.transcribe("audio.wav")["text"])

All arithmetic of the hot path runs in hand-written HIP kernels behind a C ABI (libwhisper_hip.so, see
include/whisper_hip.h); there is no CPU fallback — without the built library and a ROCm GPU, calls raise.
"""
from __future__ import annotations

import hashlib
import io
import os
import urllib.request
import warnings
from typing import List, Optional, Union

import torch

from .audio import load_audio, log_mel_spectrogram, pad_or_trim
from .decoding import (DecodingOptions, DecodingResult, decode, decode_many, detect_language, run_in_lanes,
                       run_interleaved)
from .model import ModelDimensions, Whisper
from .registry import ALIGNMENT_HEADS as _ALIGNMENT_HEADS
from .registry import MODEL_URLS as _MODELS
from .transcribe import transcribe, transcribe_batch
from . import launcher  # noqa: E402,F401  (multi-GPU layer: broadcast_weights, transcribe_sharded)

__version__ = "0.1.0"


def _fetch(url: str, root: str, in_memory: bool) -> Union[bytes, str]:
    """download (or reuse) a released checkpoint, verified against the sha256 embedded in its URL"""
    os.makedirs(root, exist_ok=True)
    want = url.split("/")[-2]
    target = os.path.join(root, os.path.basename(url))
    if os.path.exists(target) and not os.path.isfile(target):
        raise RuntimeError(f"{target} exists and is not a regular file")

    def digest_ok(data: bytes) -> bool:
        return hashlib.sha256(data).hexdigest() == want

    if os.path.isfile(target):
        with open(target, "rb") as f:
            data = f.read()
        if digest_ok(data):
            return data if in_memory else target
        warnings.warn(f"{target} exists, but the SHA256 checksum does not match; re-downloading the file")
    from tqdm import tqdm
    with urllib.request.urlopen(url) as src, open(target, "wb") as dst:
        with tqdm(total=int(src.info().get("Content-Length")), ncols=80, unit="iB", unit_scale=True,
                  unit_divisor=1024) as bar:
            for chunk in iter(lambda: src.read(8192), b""):
                dst.write(chunk)
                bar.update(len(chunk))
    with open(target, "rb") as f:
        data = f.read()
    if not digest_ok(data):
        raise RuntimeError("Model has been downloaded but the SHA256 checksum does not not match. "
                           "Please retry loading the model.")
    return data if in_memory else target


def available_models() -> List[str]:
    return list(_MODELS.keys())


def load_model(name: str, device: Optional[Union[str, torch.device]] = None, download_root: str = None,
               in_memory: bool = False) -> Whisper:
    """Load a Whisper model by official name or from a checkpoint path holding {"dims", "model_state_dict"}
    (same contract as reference whisper/__init__.py:103-161).  `device` defaults to the GPU; the checkpoint is
    read to host memory and packed into device blobs on first use of each precision."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if download_root is None:
        default = os.path.join(os.path.expanduser("~"), ".cache")
        download_root = os.path.join(os.getenv("XDG_CACHE_HOME", default), "whisper")

    if name in _MODELS:
        checkpoint_file = _fetch(_MODELS[name], download_root, in_memory)
        alignment_heads = _ALIGNMENT_HEADS[name]
    elif os.path.isfile(name):
        checkpoint_file = open(name, "rb").read() if in_memory else name
        alignment_heads = None
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")

    with (io.BytesIO(checkpoint_file) if in_memory else open(checkpoint_file, "rb")) as fp:
        checkpoint = torch.load(fp, map_location="cpu", weights_only=True)
    del checkpoint_file

    dims = ModelDimensions(**checkpoint["dims"])
    model = Whisper(dims, checkpoint["model_state_dict"], device=device)
    if alignment_heads is not None:
        model.set_alignment_heads(alignment_heads)
    return model
