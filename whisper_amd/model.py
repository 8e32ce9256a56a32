"""`Whisper` model object of whisper_amd.

Same attributes and bound methods as the reference's `whisper.model.Whisper` (model.py:252-345): `.dims`,
`.device`, `.is_multilingual`, `.num_languages`, `.encoder(mel)`, `.decoder(tokens, xa)`, `.logits`,
`.embed_audio`, `.forward`, `.alignment_heads`, `.set_alignment_heads`, `.detect_language`, `.transcribe`,
`.decode`, and the part of the nn.Module surface a caller of the reference touches on the model object:
`.parameters()` / `.named_parameters()` / `.state_dict()` / `.load_state_dict()` (reference names and shapes,
model.py:174-249), `.half()` / `.float()`, `.to()` / `.cuda()`, `.eval()` / `.train()` / `.requires_grad_()`.
It is not an nn.Module: the parameters live in one packed device blob consumed by the HIP kernels
(whisper_amd/hip.py -> libwhisper_hip.so).  The reference keeps fp32 master weights and lets the activation
dtype follow the input (model.py:44-50, decoding.py:645-646); here each precision is its own packed engine,
chosen by the dtype of the tensor you pass (fp16 mel -> fp16 engine, fp32 mel -> fp32 strict-parity engine) and
built lazily from the retained checkpoint tensors.
"""
from __future__ import annotations

import base64
import gzip
import threading
from collections import namedtuple
from dataclasses import dataclass
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import hip
from .decoding import decode as decode_function
from .decoding import detect_language as detect_language_function
from .transcribe import transcribe as transcribe_function
from .transcribe import transcribe_batch as transcribe_batch_function


@dataclass
class ModelDimensions:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


def expected_state_shapes(dims: ModelDimensions) -> Dict[str, Tuple[int, ...]]:
    """names and shapes of `Whisper(dims).state_dict()` in the reference (model.py:174-249: AudioEncoder, TextDecoder,
    ResidualAttentionBlock, MultiHeadAttention; `attn.key` has no bias, model.py:88; the decoder's causal `mask` and the
    `alignment_heads` are non-persistent buffers and absent)"""
    out: Dict[str, Tuple[int, ...]] = {}

    def linear(p, n_out, n_in, bias=True):
        out[p + ".weight"] = (n_out, n_in)
        if bias:
            out[p + ".bias"] = (n_out,)

    def lnorm(p, n):
        out[p + ".weight"] = (n,)
        out[p + ".bias"] = (n,)

    def block(p, n, cross):
        for a in (["attn", "cross_attn"] if cross else ["attn"]):
            linear(f"{p}.{a}.query", n, n)
            linear(f"{p}.{a}.key", n, n, bias=False)
            linear(f"{p}.{a}.value", n, n)
            linear(f"{p}.{a}.out", n, n)
            lnorm(f"{p}.{a}_ln", n)
        linear(f"{p}.mlp.0", 4 * n, n)
        linear(f"{p}.mlp.2", n, 4 * n)
        lnorm(f"{p}.mlp_ln", n)

    D, Dt = dims.n_audio_state, dims.n_text_state
    out["encoder.conv1.weight"], out["encoder.conv1.bias"] = (D, dims.n_mels, 3), (D,)
    out["encoder.conv2.weight"], out["encoder.conv2.bias"] = (D, D, 3), (D,)
    out["encoder.positional_embedding"] = (dims.n_audio_ctx, D)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", D, False)
    lnorm("encoder.ln_post", D)
    out["decoder.token_embedding.weight"] = (dims.n_vocab, Dt)
    out["decoder.positional_embedding"] = (dims.n_text_ctx, Dt)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", Dt, True)
    lnorm("decoder.ln", Dt)
    return out


_IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class _EncoderHandle:
    """callable `model.encoder` (AudioEncoder.forward, reference model.py:188-204)"""

    def __init__(self, owner: "Whisper"):
        self._owner = owner

    def __call__(self, x: Tensor) -> Tensor:
        owner = self._owner
        if x.dim() == 2:
            x = x[None]
        x = x.to(owner.device)
        if owner._half:
            x = x.half()                       # model.half(): fp16 engine whatever comes in
        elif x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        return owner.engine(x.dtype).encode(x)

    forward = __call__


_TASK_KEY = "_wh_task"       # entry of a kv_cache dict made by Whisper.install_kv_cache_hooks: the wh_task holding the caches


class _CacheHook:
    """what install_kv_cache_hooks returns in place of torch's RemovableHandle: remove() frees the device caches"""

    def __init__(self, cache: dict):
        self._cache = cache

    def remove(self) -> None:
        task = self._cache.pop(_TASK_KEY, None)
        if task is not None:
            task.close()


class _DecoderHandle:
    """callable `model.decoder` (TextDecoder.forward, reference model.py:227-249).  Without a cache: one teacher-forced
    pass.  With the dict returned by `model.install_kv_cache_hooks()`: incremental decoding as in the reference — the
    first call feeds all tokens, later calls only the new ones (model.py:234 `offset`), logits come back for every
    token fed — except that the keys / values live in a wh_task on the device, not as tensors in the dict."""

    def __init__(self, owner: "Whisper"):
        self._owner = owner

    def _incremental(self, x: Tensor, xa: Tensor, kv_cache: dict) -> Tensor:
        owner = self._owner
        task = kv_cache[_TASK_KEY]
        if task is None:
            engine = owner.engine(xa.dtype)
            n_rows, n_audio = x.shape[0], xa.shape[0]
            if n_rows % n_audio != 0:
                raise ValueError("token rows must be a multiple of the audio batch")
            task = engine.acquire_task(n_audio, n_rows // n_audio, owner.dims.n_text_ctx)
            task.set_audio(xa.contiguous())
            kv_cache[_TASK_KEY] = task
        if task.position == 0 or x.shape[1] > 1:
            return task.prefill(x.contiguous().long())
        return task.step(x[:, -1].long())[:, None]

    def __call__(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None) -> Tensor:
        owner = self._owner
        if owner._half:
            xa = xa.half()                     # model.half(): fp16 engine whatever comes in
        elif xa.dtype not in (torch.float16, torch.float32):
            xa = xa.float()
        if kv_cache is not None and _TASK_KEY in kv_cache:
            x, xa = x.to(owner.device), xa.to(owner.device)
            return self._incremental(x[None] if x.dim() == 1 else x, xa[None] if xa.dim() == 2 else xa, kv_cache)
        if kv_cache:
            # A dict the hooks of install_kv_cache_hooks() never filled.  The reference (model.py:233-249, 96-104) finds
            # none of its projection modules in it, so it computes every key / value afresh from `x` and only takes the
            # position offset from the first entry (`next(iter(kv_cache.values())).shape[1]`).  Offset 0 is then an
            # ordinary teacher-forced pass; a non-zero offset would embed x at shifted positions WITHOUT the earlier
            # tokens' keys — not a decoding step of anything, and not offered here.
            first = next(iter(kv_cache.values()))
            offset = int(first.shape[1]) if hasattr(first, "shape") and len(first.shape) > 1 else 0
            if offset != 0:
                raise NotImplementedError(
                    "kv_cache holds tensors that model.install_kv_cache_hooks() did not put there; incremental decoding "
                    "keeps its keys / values in a device-side task: cache, hooks = model.install_kv_cache_hooks()")
        x = x.to(owner.device)
        xa = xa.to(owner.device)
        if x.dim() == 1:
            x = x[None]
        if xa.dim() == 2:
            xa = xa[None]
        engine = owner.engine(xa.dtype)
        n_rows, n_audio = x.shape[0], xa.shape[0]
        if n_audio == 1 and n_rows > 1:
            group = n_rows
        elif n_rows % n_audio == 0:
            group = n_rows // n_audio
        else:
            raise ValueError("token rows must be a multiple of the audio batch")
        task = engine.acquire_task(n_audio, group, max(int(x.shape[1]), 8))
        try:
            task.set_audio(xa.contiguous())
            return task.prefill(x.contiguous().long())
        finally:
            task.close()

    forward = __call__


_ENGINE_LOCK = threading.Lock()


class Whisper:
    def __init__(self, dims: ModelDimensions, state_dict: Dict[str, Tensor], device=None):
        self.dims = dims
        self._state_dict = state_dict            # reference-format tensors (any device), kept to build engines
        self._device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self._engines: Dict[torch.dtype, hip.HipModel] = {}
        self._half = False                       # model.half() was called: every input runs on the fp16 engine
        self.training = False
        self.encoder = _EncoderHandle(self)
        self.decoder = _DecoderHandle(self)
        # default: every head of the upper half of the decoder (reference model.py:270-276)
        heads = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
        heads[dims.n_text_layer // 2:] = True
        self.alignment_heads = heads.to_sparse()

    # ---- engines ---------------------------------------------------------------------------------------
    def engine(self, dtype: torch.dtype) -> hip.HipModel:
        if dtype not in (torch.float16, torch.float32):
            raise TypeError(f"unsupported activation dtype {dtype}")
        eng = self._engines.get(dtype)
        if eng is None:
            with _ENGINE_LOCK:                     # threads of several lanes (decoding.run_in_lanes) may ask at once: pack once
                eng = self._engines.get(dtype)
                if eng is None:
                    hip.require_gpu(self._device)
                    code = hip.WH_F16 if dtype == torch.float16 else hip.WH_F32
                    blob = hip.pack_weights(self._state_dict, self.dims, code, self._device)
                    eng = hip.HipModel(self.dims, code, blob)
                    self._engines[dtype] = eng
        return eng

    def adopt_engine(self, dtype: torch.dtype, engine: hip.HipModel) -> None:
        """install an already-packed engine (multi-GPU load: the blob arrived by RCCL broadcast)"""
        self._engines[dtype] = engine

    # ---- reference surface -----------------------------------------------------------------------------
    def set_alignment_heads(self, dump: bytes):
        array = np.frombuffer(gzip.decompress(base64.b85decode(dump)), dtype=bool).copy()
        mask = torch.from_numpy(array).reshape(self.dims.n_text_layer, self.dims.n_text_head)
        self.alignment_heads = mask.to_sparse()

    def embed_audio(self, mel: Tensor) -> Tensor:
        return self.encoder(mel)

    def logits(self, tokens: Tensor, audio_features: Tensor) -> Tensor:
        return self.decoder(tokens, audio_features)

    def forward(self, mel: Tensor, tokens: Tensor) -> Tensor:
        return self.decoder(tokens, self.encoder(mel))

    __call__ = forward

    @property
    def device(self) -> torch.device:
        return self._device

    def to(self, *args, **kwargs) -> "Whisper":
        """nn.Module.to for what an inference user passes: a device (`model.to("cuda:1")`), a floating dtype
        (`model.to(torch.float16)` == `model.half()`, `torch.float32` == `model.float()`), or both (positional or as
        `device=` / `dtype=`).  Anything else (a tensor, memory formats) is not part of this surface and raises."""
        device, dtype = kwargs.pop("device", None), kwargs.pop("dtype", None)
        kwargs.pop("non_blocking", None)
        if kwargs:
            raise TypeError(f"Whisper.to(): unsupported arguments {sorted(kwargs)}")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, int, torch.device)):
                device = a
            else:
                raise TypeError(f"Whisper.to(): expected a device or a dtype, got {type(a).__name__}")
        if dtype is not None:
            if dtype == torch.float16:
                self.half()
            elif dtype == torch.float32:
                self.float()
            else:
                raise TypeError(f"unsupported parameter dtype {dtype} (float16 or float32)")
        if device is not None:
            device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            if device != self._device:
                self._device = device
                for eng in self._engines.values():
                    eng.drop_cached_tasks()
                self._engines.clear()
        return self

    def cuda(self, device=None) -> "Whisper":
        return self.to(torch.device("cuda", device) if isinstance(device, int) else (device or "cuda"))

    def cpu(self) -> "Whisper":
        """the retained checkpoint tensors stay usable (state_dict, save); computing needs a GPU device (no CPU path)"""
        return self.to("cpu")

    def eval(self) -> "Whisper":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "Whisper":
        """inference-only engine: the flag is kept for callers that toggle it, nothing else changes (the reference has no
        dropout or batch statistics either, model.py:142-171)"""
        self.training = bool(mode)
        return self

    def requires_grad_(self, requires_grad: bool = True) -> "Whisper":
        if requires_grad:
            raise NotImplementedError("whisper_amd is an inference path: there is no backward pass")
        return self

    def half(self) -> "Whisper":
        """reference: `model.half()` turns the parameters into fp16, after which only fp16 inputs compute (what
        `DecodingOptions(fp16=True)` relies on, decoding.py:645-646).  Here: every input is run on the fp16 engine
        (fp32 tensors are converted on entry instead of raising a dtype error)."""
        self._half = True
        return self

    def float(self) -> "Whisper":
        """reference: fp32 master weights, the activation dtype follows the input (Linear casts the weight to x.dtype,
        model.py:44-50).  Here: the engine follows the input dtype again (fp32 in -> fp32 strict engine, fp16 in -> fp16)."""
        self._half = False
        return self

    def state_dict(self) -> Dict[str, Tensor]:
        return dict(self._state_dict)

    def named_parameters(self) -> Iterator[Tuple[str, Tensor]]:
        """the retained checkpoint tensors under the reference's names (the encoder's positional table is a buffer there
        and is skipped, model.py:185)"""
        for k, v in self._state_dict.items():
            if k != "encoder.positional_embedding":
                yield k, v

    def parameters(self) -> Iterator[Tensor]:
        """The retained checkpoint tensors, on the device and in the dtype they were loaded with (normally CPU fp32) — NOT the
        packed engine blobs.  The reference idiom `next(model.parameters()).device` therefore does not say where this model
        computes; `model.device` (and `model.half()` / the input dtype) do."""
        for _, v in self.named_parameters():
            yield v

    def load_state_dict(self, state_dict: Dict[str, Tensor], strict: bool = True):
        """nn.Module.load_state_dict for the reference's checkpoint layout (whisper/__init__.py:150-156): names and shapes
        are checked against model.py:174-249, the packed engines are dropped and rebuilt lazily from the new tensors.
        Returns (missing_keys, unexpected_keys) like torch; with strict=True (default) either being non-empty raises."""
        want = expected_state_shapes(self.dims)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        bad = [f"{k}: {tuple(state_dict[k].shape)} != {want[k]}" for k in want
               if k in state_dict and isinstance(state_dict[k], Tensor) and tuple(state_dict[k].shape) != want[k]]
        if bad:
            raise RuntimeError("size mismatch in load_state_dict: " + "; ".join(bad[:8]))
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:8]}, unexpected keys {unexpected[:8]}")
        notensor = [k for k in want if k in state_dict and not isinstance(state_dict[k], Tensor)]
        if notensor:
            raise TypeError(f"load_state_dict: values of {notensor[:8]} are not tensors")
        merged = dict(self._state_dict)
        # torch COPIES into the module's parameters: a later in-place edit of the caller's tensors must not reach the model
        merged.update({k: v.detach().clone() for k, v in state_dict.items() if k in want})
        still = [k for k in want if k not in merged]
        if still and strict:
            raise RuntimeError(f"load_state_dict: missing keys {still[:8]}")
        self._state_dict = merged
        for eng in self._engines.values():
            eng.drop_cached_tasks()
        self._engines.clear()
        return _IncompatibleKeys(missing, unexpected)

    @property
    def is_multilingual(self) -> bool:
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self) -> int:
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def install_kv_cache_hooks(self, cache: Optional[dict] = None):
        """(cache, hooks) as reference model.py:310-341: pass `cache` to `model.decoder(tokens, xa, kv_cache=cache)` to
        decode incrementally; `hook.remove()` on every returned hook releases the caches.  There are no nn.Module
        hooks underneath — the dict only carries the handle of the wh_task that owns the device-side K/V."""
        cache = {**cache} if cache is not None else {}
        cache[_TASK_KEY] = None
        return cache, [_CacheHook(cache)]

    detect_language = detect_language_function
    transcribe = transcribe_function
    transcribe_batch = transcribe_batch_function     # extension: lock-step batching over files
    decode = decode_function
