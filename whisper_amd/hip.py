"""ctypes binding of libwhisper_hip.so (include/whisper_hip.h) — the only door to the HIP kernels.

Nothing here computes: tensors are allocated by torch (device memory, streams), their raw pointers are
handed to the C ABI.  If the shared library is missing or the device is not a ROCm GPU, every entry
point raises — there is deliberately no CPU fallback on the product path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

_LIB = None
# WHISPER_AMD_LIB: another build of the same C ABI — the developer build `make -C whisper_amd/csrc dev`
# (libwhisper_hip_dev.so, -DWH_DEV: the only build that reads the A/B environment switches of tools/README.md).  Still a
# HIP library: there is no non-HIP path to select.
_LIB_PATH = os.environ.get("WHISPER_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwhisper_hip.so")

WH_F32, WH_F16 = 0, 1
WH_TASK_CAPTURE_Q = 1
WH_TASK_TWO_LAUNCH_SELF = 2
WH_TASK_TWO_LAUNCH_CROSS = 4
WH_TASK_EXPIRE_HANDOFFS = 16        # fault injection (include/whisper_hip.h): every hand-off poll gives up at once
WH_TASK_FUSED_SELF = 32             # opt-in: LN + QKV + cache append + self attention as one launch (sattn8_kernel)
WH_WEIGHTS_DEC_LN_FOLDED = 1
WH_WEIGHTS_ENC_QK_SCALED = 2
# sqrt(0.125 * log2 e): with it in both the query and the key projection, K.Q^T is the exp2 argument of the softmax
ENC_QK_SCALE = (0.125 * 1.4426950408889634) ** 0.5
MEL_SCRATCH_BYTES = 2048


class HipError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


_LAYER_FIELDS = (
    "attn_ln_w", "attn_ln_b", "qkv_w", "qkv_b", "out_w", "out_b",
    "cross_ln_w", "cross_ln_b", "cq_w", "cq_b", "ckv_w", "ckv_b", "cout_w", "cout_b",
    "mlp_ln_w", "mlp_ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class ModelWeights(C.Structure):
    _fields_ = [
        ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p),
        ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
        ("enc_pos", C.c_void_p),
        ("enc_layers", C.POINTER(LayerWeights)),
        ("enc_ln_post_w", C.c_void_p), ("enc_ln_post_b", C.c_void_p),
        ("tok_emb", C.c_void_p), ("dec_pos", C.c_void_p),
        ("dec_layers", C.POINTER(LayerWeights)),
        ("dec_ln_w", C.c_void_p), ("dec_ln_b", C.c_void_p),
        ("flags", C.c_uint32),
    ]


class GreedyParams(C.Structure):
    _fields_ = [
        ("sample_begin", C.c_int32), ("max_steps", C.c_int32), ("n_ctx", C.c_int32), ("eot", C.c_int32),
        ("timestamp_begin", C.c_int32), ("no_timestamps", C.c_int32),
        ("max_initial_timestamp_index", C.c_int32), ("suppress_blank", C.c_int32),
        ("blank_token", C.c_int32), ("suppress_mask", C.c_void_p),
        ("temperature", C.c_float), ("reserved", C.c_uint32), ("seed", C.c_uint64),
    ]


class BeamParams(C.Structure):
    _fields_ = [("rules", GreedyParams), ("beam_size", C.c_int32), ("max_candidates", C.c_int32)]


# name -> (restype, argtypes); also the list the CPU-only symbol test walks
SIGNATURES = {
    "wh_abi_version": (C.c_int, []),
    "wh_status_string": (C.c_char_p, [C.c_int]),
    "wh_last_hip_error": (C.c_int, []),
    "wh_last_hip_error_string": (C.c_char_p, []),
    "wh_log_mel": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wh_model_create": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(ModelWeights), C.POINTER(C.c_void_p)]),
    "wh_model_destroy": (None, [C.c_void_p]),
    "wh_encoder_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "wh_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wh_task_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "wh_task_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "wh_task_destroy": (None, [C.c_void_p]),
    "wh_task_set_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "wh_task_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p]),
    "wh_task_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "wh_task_rearrange": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "wh_task_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wh_task_set_lag": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "wh_task_position": (C.c_int, [C.c_void_p]),
    "wh_task_info": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wh_task_greedy": (C.c_int, [C.c_void_p, C.POINTER(GreedyParams), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "wh_task_beam": (C.c_int, [C.c_void_p, C.POINTER(BeamParams), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int32), C.c_void_p]),
    "wh_task_greedy_begin": (C.c_int, [C.c_void_p, C.POINTER(GreedyParams), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "wh_task_beam_begin": (C.c_int, [C.c_void_p, C.POINTER(BeamParams), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wh_task_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "wh_task_cross_qk": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wh_task_bench_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_void_p]),
    "wh_align_batch_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "wh_task_align_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "wh_median_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "wh_dtw_trace": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wh_dtw_backtrace_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "wh_align_matrix": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def lib_path() -> str:
    return _LIB_PATH


def lib():
    """Load libwhisper_hip.so (built in-tree by `make -C whisper_amd/csrc` / __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(_LIB_PATH):
            raise HipError(
                f"{_LIB_PATH} is missing: build it with `make -C whisper_amd/csrc` "
                "(or __graft_entry__.build()). There is no non-HIP fallback.")
        handle = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.wh_abi_version() != 1:
            raise HipError("libwhisper_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


WH_ERR_HANDOFF = 6
WH_RUNNING = 7                      # wh_task_poll: the loop begun with wh_task_*_begin has not ended yet


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        L = lib()
        msg = L.wh_status_string(rc).decode()
        if rc == 3:
            msg += f" ({L.wh_last_hip_error_string().decode()})"
        raise HipError(f"{what or 'libwhisper_hip'}: {msg}")


def require_gpu(device: torch.device) -> None:
    if device.type != "cuda" or not torch.cuda.is_available():
        raise HipError("the HIP path needs a ROCm GPU device ('cuda'); no CPU fallback exists in whisper_amd")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(stream: torch.cuda.Stream) -> int:
    return stream.cuda_stream


# ---------------------------------------------------------------------------------------------------
# weight packing: reference checkpoint names (whisper/model.py module tree) -> one device blob
# ---------------------------------------------------------------------------------------------------
def _align(x: int, a: int = 256) -> int:
    return (x + a - 1) // a * a


def _fold_ln(w: torch.Tensor, b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor):
    """Linear(LayerNorm(x)) with the affine part of the LayerNorm moved into the Linear:
    W (g * xhat + beta) + b  ==  (W * g) xhat + (b + W beta).  Computed in fp32 on the weights' device."""
    wf = w.float()
    return wf * ln_w.float().to(wf.device)[None, :], b.float().to(wf.device) + wf @ ln_b.float().to(wf.device)


def _block_pieces(sd: Dict[str, torch.Tensor], prefix: str, cross: bool, D: int, fold: bool = False,
                  qk_scale: float = 1.0):
    """yield (field, tensor, is_matrix) for one ResidualAttentionBlock (whisper/model.py:142-171).
    fold (decoder blocks of the fp16 engine): the three LayerNorms' weight / bias are folded into the projection
    that consumes them and replaced by (1, 0) — WH_WEIGHTS_DEC_LN_FOLDED in include/whisper_hip.h.
    qk_scale (encoder blocks of the fp16 engine): the self-attention query and key projections (weight rows and the
    query bias; the key has none) are multiplied by it — WH_WEIGHTS_ENC_QK_SCALED.  whisper/model.py:118-121 scales
    q and k by n_state_head ** -0.25 each; here both carry sqrt(0.125 * log2 e) instead and the attention kernel
    exponentiates K.Q^T with v_exp_f32 (base 2) directly."""
    z = torch.zeros(D, dtype=torch.float32)
    g = lambda k: sd[prefix + k]
    one = lambda: torch.ones(D, dtype=torch.float32)
    qkv_w = torch.cat([g("attn.query.weight"), g("attn.key.weight"), g("attn.value.weight")], 0)
    qkv_b = torch.cat([g("attn.query.bias").float().cpu(), z, g("attn.value.bias").float().cpu()], 0)
    if fold:
        qkv_w, qkv_b = _fold_ln(qkv_w, qkv_b, g("attn_ln.weight"), g("attn_ln.bias"))
    if qk_scale != 1.0:
        qkv_w = qkv_w.float().clone()
        qkv_w[: 2 * D] *= qk_scale
        qkv_b = qkv_b.clone()
        qkv_b[: 2 * D] *= qk_scale
    yield "attn_ln_w", one() if fold else g("attn_ln.weight"), False
    yield "attn_ln_b", z if fold else g("attn_ln.bias"), False
    yield "qkv_w", qkv_w, True
    yield "qkv_b", qkv_b, False
    yield "out_w", g("attn.out.weight"), True
    yield "out_b", g("attn.out.bias"), False
    if cross:
        cq_w, cq_b = g("cross_attn.query.weight"), g("cross_attn.query.bias")
        if fold:
            cq_w, cq_b = _fold_ln(cq_w, cq_b, g("cross_attn_ln.weight"), g("cross_attn_ln.bias"))
        yield "cross_ln_w", one() if fold else g("cross_attn_ln.weight"), False
        yield "cross_ln_b", z if fold else g("cross_attn_ln.bias"), False
        yield "cq_w", cq_w, True
        yield "cq_b", cq_b, False
        yield "ckv_w", torch.cat([g("cross_attn.key.weight"), g("cross_attn.value.weight")], 0), True
        yield "ckv_b", torch.cat([z, g("cross_attn.value.bias").float().cpu()], 0), False
        yield "cout_w", g("cross_attn.out.weight"), True
        yield "cout_b", g("cross_attn.out.bias"), False
    fc1_w, fc1_b = g("mlp.0.weight"), g("mlp.0.bias")
    if fold:
        fc1_w, fc1_b = _fold_ln(fc1_w, fc1_b, g("mlp_ln.weight"), g("mlp_ln.bias"))
    yield "mlp_ln_w", one() if fold else g("mlp_ln.weight"), False
    yield "mlp_ln_b", z if fold else g("mlp_ln.bias"), False
    yield "fc1_w", fc1_w, True
    yield "fc1_b", fc1_b, False
    yield "fc2_w", g("mlp.2.weight"), True
    yield "fc2_b", g("mlp.2.bias"), False


def _conv_as_gemm(w: torch.Tensor, k_pad: int) -> torch.Tensor:
    """conv weight [D][C][3] -> GEMM weight [D][k_pad] with column kk*C + c (zero padded)"""
    D, Cin, _ = w.shape
    out = torch.zeros(D, k_pad, dtype=w.dtype, device=w.device)
    out[:, : 3 * Cin] = w.permute(0, 2, 1).reshape(D, 3 * Cin)
    return out


def scales_encoder_qk(dtype: int) -> bool:
    """fp16 blobs carry ENC_QK_SCALE in the encoder's query / key projections (the MFMA flash-attention kernel is the
    only reader of those activations); the fp32 strict-parity engine keeps the reference's operation order."""
    return dtype == WH_F16


def folds_decoder_ln(dtype: int) -> bool:
    """The fp16 engine's blobs carry the decoder LayerNorm affine parameters folded into the projections (the decode
    GEMV then normalises its x fragments in registers without touching gamma / beta); the fp32 strict-parity engine
    keeps the reference's operation order."""
    return dtype == WH_F16


def model_pieces(sd: Dict[str, torch.Tensor], dims, fold_dec_ln: bool = False,
                 scale_enc_qk: bool = False) -> List[Tuple[str, torch.Tensor, bool]]:
    D = dims.n_audio_state
    kc1 = _align(3 * dims.n_mels, 64)
    out = [
        ("conv1_w", _conv_as_gemm(sd["encoder.conv1.weight"], kc1), True),
        ("conv1_b", sd["encoder.conv1.bias"], False),
        ("conv2_w", _conv_as_gemm(sd["encoder.conv2.weight"], 3 * D), True),
        ("conv2_b", sd["encoder.conv2.bias"], False),
        ("enc_pos", sd["encoder.positional_embedding"], False),
        ("enc_ln_post_w", sd["encoder.ln_post.weight"], False),
        ("enc_ln_post_b", sd["encoder.ln_post.bias"], False),
        ("tok_emb", sd["decoder.token_embedding.weight"], True),
        ("dec_pos", sd["decoder.positional_embedding"], False),
        ("dec_ln_w", sd["decoder.ln.weight"], False),
        ("dec_ln_b", sd["decoder.ln.bias"], False),
    ]
    for i in range(dims.n_audio_layer):
        for f, t, mat in _block_pieces(sd, f"encoder.blocks.{i}.", False, D, qk_scale=ENC_QK_SCALE if scale_enc_qk else 1.0):
            out.append((f"enc.{i}.{f}", t, mat))
    for i in range(dims.n_text_layer):
        for f, t, mat in _block_pieces(sd, f"decoder.blocks.{i}.", True, dims.n_text_state, fold=fold_dec_ln):
            out.append((f"dec.{i}.{f}", t, mat))
    return out


def blob_layout(dims, dtype: int) -> Tuple[Dict[str, Tuple[int, Tuple[int, ...], bool]], int]:
    """offsets of every packed tensor, computed from dims alone (every rank derives the same layout)"""
    D, Dt, M = dims.n_audio_state, dims.n_text_state, dims.n_mels
    kc1 = _align(3 * M, 64)
    shapes: List[Tuple[str, Tuple[int, ...], bool]] = [
        ("conv1_w", (D, kc1), True), ("conv1_b", (D,), False),
        ("conv2_w", (D, 3 * D), True), ("conv2_b", (D,), False),
        ("enc_pos", (dims.n_audio_ctx, D), False),
        ("enc_ln_post_w", (D,), False), ("enc_ln_post_b", (D,), False),
        ("tok_emb", (dims.n_vocab, Dt), True), ("dec_pos", (dims.n_text_ctx, Dt), False),
        ("dec_ln_w", (Dt,), False), ("dec_ln_b", (Dt,), False),
    ]

    def block(prefix, d, cross):
        s = [("attn_ln_w", (d,), False), ("attn_ln_b", (d,), False), ("qkv_w", (3 * d, d), True),
             ("qkv_b", (3 * d,), False), ("out_w", (d, d), True), ("out_b", (d,), False)]
        if cross:
            s += [("cross_ln_w", (d,), False), ("cross_ln_b", (d,), False), ("cq_w", (d, d), True),
                  ("cq_b", (d,), False), ("ckv_w", (2 * d, d), True), ("ckv_b", (2 * d,), False),
                  ("cout_w", (d, d), True), ("cout_b", (d,), False)]
        s += [("mlp_ln_w", (d,), False), ("mlp_ln_b", (d,), False), ("fc1_w", (4 * d, d), True),
              ("fc1_b", (4 * d,), False), ("fc2_w", (d, 4 * d), True), ("fc2_b", (d,), False)]
        return [(prefix + n, sh, m) for n, sh, m in s]

    for i in range(dims.n_audio_layer):
        shapes += block(f"enc.{i}.", D, False)
    for i in range(dims.n_text_layer):
        shapes += block(f"dec.{i}.", Dt, True)
    esize = 2 if dtype == WH_F16 else 4
    layout, off = {}, 0
    for name, shape, mat in shapes:
        n = 1
        for s in shape:
            n *= s
        layout[name] = (off, shape, mat)
        off = _align(off + n * (esize if mat else 4))
    return layout, off + 4096   # tail slack: padded-K GEMM reads never leave the blob; its last 64 bytes = blob_header


BLOB_MAGIC = 0x31424857          # "WHB1"


def blob_header(blob: torch.Tensor, total: int) -> Optional[Tuple[int, int]]:
    """(dtype, flags) recorded by pack_weights in the last 64 bytes of the blob, or None for a blob packed without
    a header.  The flags say how the kernels must read the weights (LayerNorm folded / softmax scale in q, k): they
    travel WITH the bytes (also through broadcast_weights) instead of being re-derived from the dtype at load."""
    h = blob[total - 64: total - 48].view(torch.int32).cpu().tolist()
    return (h[1], h[2]) if h[0] == BLOB_MAGIC else None


def pack_weights(sd: Dict[str, torch.Tensor], dims, dtype: int, device: torch.device) -> torch.Tensor:
    """Cast + copy every tensor of a reference-format state_dict into one device blob."""
    layout, total = blob_layout(dims, dtype)
    blob = torch.zeros(total, dtype=torch.uint8, device=device)
    tdt = torch.float16 if dtype == WH_F16 else torch.float32
    for name, t, mat in model_pieces(sd, dims, fold_dec_ln=folds_decoder_ln(dtype), scale_enc_qk=scales_encoder_qk(dtype)):
        off, shape, mat2 = layout[name]
        assert mat == mat2 and tuple(t.shape) == tuple(shape), (name, t.shape, shape)
        dt = tdt if mat else torch.float32
        nbytes = t.numel() * (2 if dt == torch.float16 else 4)
        blob[off: off + nbytes].view(dt).copy_(t.detach().to(device=device, dtype=dt).reshape(-1))
    flags = (WH_WEIGHTS_DEC_LN_FOLDED if folds_decoder_ln(dtype) else 0) | (WH_WEIGHTS_ENC_QK_SCALED if scales_encoder_qk(dtype) else 0)
    blob[total - 64: total - 48].view(torch.int32).copy_(torch.tensor([BLOB_MAGIC, dtype, flags, 0], dtype=torch.int32))
    return blob


_FROZEN = False


def _freeze_startup_objects() -> None:
    """Once per process, when the first engine is created: move everything the cyclic garbage collector tracks so far — ~170 000
    objects, nearly all of them `import torch`'s functions, tuples and dicts, which live as long as the process — into the
    permanent generation (gc.freeze, the standard advice for long-running services).  A full (generation 2) collection otherwise
    walks all of them, 40 - 80 ms on the host, and it falls into whichever call happens to allocate the container that trips
    the threshold: round 5's word-timestamp leg showed it as one call in five taking 111 - 134 ms instead of 76
    (tools/wts_gc_probe.py: the slow call contains exactly one generation-2 pass of 40 ms; none after the freeze).  Objects
    created later are collected as always.  WHISPER_AMD_NO_GC_FREEZE=1 leaves the collector alone."""
    global _FROZEN
    if _FROZEN or os.environ.get("WHISPER_AMD_NO_GC_FREEZE") == "1":
        return
    _FROZEN = True
    import gc
    gc.collect()
    gc.freeze()


class HipModel:
    """wh_model handle + the weight blob it points into."""

    def __init__(self, dims, dtype: int, blob: torch.Tensor):
        require_gpu(blob.device)
        self.dims, self.dtype, self.blob = dims, dtype, blob
        self.device = blob.device
        self.torch_dtype = torch.float16 if dtype == WH_F16 else torch.float32
        layout, total = blob_layout(dims, dtype)
        assert blob.numel() >= total
        base = blob.data_ptr()
        addr = lambda n: base + layout[n][0]
        self._enc = (LayerWeights * dims.n_audio_layer)()
        self._dec = (LayerWeights * dims.n_text_layer)()
        for i in range(dims.n_audio_layer):
            for f in _LAYER_FIELDS:
                key = f"enc.{i}.{f}"
                setattr(self._enc[i], f, addr(key) if key in layout else None)
        for i in range(dims.n_text_layer):
            for f in _LAYER_FIELDS:
                setattr(self._dec[i], f, addr(f"dec.{i}.{f}"))
        w = ModelWeights()
        for f in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "enc_pos", "enc_ln_post_w", "enc_ln_post_b",
                  "tok_emb", "dec_pos", "dec_ln_w", "dec_ln_b"):
            setattr(w, f, addr(f))
        w.enc_layers = C.cast(self._enc, C.POINTER(LayerWeights))
        w.dec_layers = C.cast(self._dec, C.POINTER(LayerWeights))
        hdr = blob_header(blob, total)
        if hdr is not None:                            # the blob says how it was packed
            if hdr[0] != dtype:
                raise HipError(f"weight blob was packed for dtype {hdr[0]}, engine asked for {dtype}")
            w.flags = hdr[1]
        else:                                          # header-less blob (packed by an older pack_weights)
            w.flags = (WH_WEIGHTS_DEC_LN_FOLDED if folds_decoder_ln(dtype) else 0) | (
                WH_WEIGHTS_ENC_QK_SCALED if scales_encoder_qk(dtype) else 0)
        d = Dims(*[getattr(dims, n) for n, _ in Dims._fields_])
        h = C.c_void_p()
        check(lib().wh_model_create(C.byref(d), dtype, C.byref(w), C.byref(h)), "wh_model_create")
        self.handle = h
        self.stream = torch.cuda.Stream(device=self.device)
        self._enc_ws: Optional[torch.Tensor] = None
        self._task_cache: List["HipTask"] = []          # idle tasks, most recently used last
        # Several PASSES may be in flight on one engine (`lane`, whisper_amd.decode_many): a decode chain is ~190 dependent
        # launches per token and leaves the chip idle between them, so independent chains on their own streams fill each
        # other's gaps (large-v3, 8 clips per pass: 692 audio-s/s one pass at a time, 917 / 1025 with 2 / 3 in flight).
        # Host-side state shared by the lanes' threads is guarded here; the encoder has ONE workspace and ONE stream.
        self._lock = threading.RLock()                  # task cache, workspace (re)allocation
        self._enc_lock = threading.Lock()               # one encoder's launches are enqueued without another's in between
        self._tls = threading.local()                   # .stream: the lane stream of the calling thread (None: self.stream)
        self._lane_pool: List[torch.cuda.Stream] = []   # idle lane streams (`lane`)
        self.debug_task_flags = 0                       # OR-ed into the flags of every task created (tests: WH_TASK_EXPIRE_HANDOFFS)
        # workspaces kept alive between windows: WH_TASK_CACHE_GB, else 10 % of the device memory (28 GB of the 288 GB of
        # an MI355X; a smaller GPU gets a smaller cache).  Allocation failures anywhere on this engine's path drop the
        # cache and retry once (`_alloc`), because torch's own out-of-memory retry cannot reclaim tensors we hold.
        env = os.environ.get("WH_TASK_CACHE_GB")
        if env is not None:
            self.task_cache_bytes = int(float(env) * (1 << 30))
        else:
            with torch.cuda.device(self.device):
                self.task_cache_bytes = int(torch.cuda.mem_get_info()[1] * 0.10)
        _freeze_startup_objects()

    # -- decoding tasks are expensive to set up (GBs of workspace, a 250-node graph capture): keep them -------------
    @contextlib.contextmanager
    def lane(self, stream: Optional[torch.cuda.Stream] = None):
        """Run the calling THREAD's work on a stream of its own: inside the block `stream` (a new one by default) is
        torch's current stream and the stream of every task this thread acquires from the engine, so that the thread's
        log-mel, sampling and decode chains overlap those of other threads' lanes.  Without it all tasks of an engine share
        `self.stream` and run one after the other.  The encoder stays on the engine's stream (one workspace): encoders of
        different lanes are ordered among themselves and overlap the other lanes' decode chains."""
        # lane streams are pooled: the GPU has four hardware queues, and every stream ever created keeps its place on one of
        # them — a fresh stream per lane and call would soon have two live lanes sharing a queue
        own = stream is None
        if own:
            with self._lock:
                st = self._lane_pool.pop() if self._lane_pool else None
            if st is None:
                st = torch.cuda.Stream(device=self.device)
        else:
            st = stream
        prev = getattr(self._tls, "stream", None)
        self._tls.stream = st
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(st):
                yield st
        finally:
            self._tls.stream = prev
            if own:
                with self._lock:
                    self._lane_pool.append(st)

    def adopt_lane_streams(self, streams: Sequence[torch.cuda.Stream]) -> None:
        """hand idle streams to the lane pool (a caller that ran its own lanes, like bench.py, gives its streams back so that
        later `lane()` calls reuse them instead of creating more streams than the GPU has hardware queues)"""
        with self._lock:
            for st in streams:
                if all(st is not x for x in self._lane_pool):
                    self._lane_pool.append(st)

    def task_stream(self) -> torch.cuda.Stream:
        """the stream tasks acquired by the calling thread run on"""
        return getattr(self._tls, "stream", None) or self.stream

    def acquire_task(self, n_audio: int, n_group: int, max_prefill: int, capture_q: bool = False,
                     stream: Optional[torch.cuda.Stream] = None) -> "HipTask":
        """A reset task of this shape: a cached one (its workspace and captured step graph are reused) or a new one.
        `task.close()` hands it back.  max_prefill is rounded up so that windows with prompts of different lengths
        share a task.  The task runs on `stream`, else on the calling thread's lane stream (`lane`), else on the engine's."""
        if max_prefill <= 8:
            max_prefill = 8
        elif max_prefill <= 64:
            max_prefill = 64
        else:
            max_prefill = self.dims.n_text_ctx
        if stream is None:
            stream = self.task_stream()
        # A task of a LANE (several chains share the chip: `lane`, run_interleaved, run_in_lanes) runs NO kernel that spins: its cross
        # attention is two launches like its self attention.  The fused launch (csrc/xattn.hip: consumers poll for q while their K/V
        # streams in) wins where a chain has the chip to itself (13.1 vs 15.4 us per layer); beside other chains — and their encoders,
        # whose persistent GEMM workgroups hold whole CUs — a producer workgroup is now and then dispatched milliseconds late, the
        # 8192-poll hang guard trips and the task falls back to these same two-launch kernels after wasting a pass: 2 of 25 fresh
        # processes with three 8-row chains in flight (876 instead of 1054 audio-s/s; results exact either way), against a steady 1015
        # with the two-launch form (profiles/r06_lanes.txt).  Predictable beats 4 % faster most of the time.
        in_lane = getattr(self._tls, "stream", None) is not None and stream is self._tls.stream
        key = (n_audio, n_group, max_prefill, capture_q, stream, in_lane)
        task = None
        with self._lock:
            for i in range(len(self._task_cache) - 1, -1, -1):
                if self._task_cache[i].cache_key == key:
                    task = self._task_cache.pop(i)
                    break
        if task is not None:
            task.reset()
            return task
        task = HipTask(self, n_audio, n_group, max_prefill, capture_q=capture_q, stream=stream, two_launch_cross=in_lane)
        task._cached = True
        return task

    def _alloc(self, nbytes: int) -> torch.Tensor:
        """device bytes for a workspace; when the allocator is out of memory the idle cached tasks go first"""
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        except torch.OutOfMemoryError:
            if not self._task_cache:
                raise
            self.drop_cached_tasks()
            torch.cuda.empty_cache()
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _release_task(self, task: "HipTask") -> bool:
        if task.ws is None or task.held_bytes() > self.task_cache_bytes:
            return False
        evicted = []
        with self._lock:
            if any(t is task for t in self._task_cache):     # closed twice: it is already idle, do not list it again
                return True
            self._task_cache.append(task)
            total = sum(t.held_bytes() for t in self._task_cache)
            while total > self.task_cache_bytes and len(self._task_cache) > 1:
                old = self._task_cache.pop(0)
                total -= old.held_bytes()
                evicted.append(old)
        for old in evicted:
            old.destroy()
        return True

    def drop_cached_tasks(self) -> None:
        while True:
            with self._lock:
                if not self._task_cache:
                    return
                t = self._task_cache.pop()
            t.destroy()

    def __del__(self):
        try:
            self.drop_cached_tasks()
            if getattr(self, "handle", None):
                lib().wh_model_destroy(self.handle)
                self.handle = None
        except Exception:   # noqa: BLE001 — interpreter shutdown: module globals (even HipError) may already be None
            pass

    # -- AudioEncoder.forward ----------------------------------------------------------------------
    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        d = self.dims
        assert mel.is_cuda and mel.dim() == 3
        if mel.shape[1:] != (d.n_mels, 2 * d.n_audio_ctx):
            raise AssertionError("incorrect audio shape")   # whisper/model.py:197
        if mel.dtype not in (torch.float32, torch.float16):
            mel = mel.float()
        mel = mel.contiguous()
        B = mel.shape[0]
        need = lib().wh_encoder_workspace_bytes(self.handle, B)
        out = torch.empty(B, d.n_audio_ctx, d.n_audio_state, dtype=self.torch_dtype, device=self.device)
        # one encoder at a time ENQUEUES (a few hundred launches into the engine's stream, all on the one workspace): another
        # thread's encoder must not slip its launches in between.  The lock is held for the enqueue only, not for the run.
        with self._enc_lock, torch.cuda.device(self.device):   # the C ABI launches on the calling thread's current device
            if self._enc_ws is None or self._enc_ws.numel() < need:
                if self._enc_ws is not None:
                    self.stream.synchronize()          # a running encoder may still use the workspace being replaced
                self._enc_ws = None
                self._enc_ws = self._alloc(need)
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            check(lib().wh_encode(self.handle, mel.data_ptr(), int(mel.dtype == torch.float16), B, out.data_ptr(),
                                  self._enc_ws.data_ptr(), self._enc_ws.numel(), stream_ptr(self.stream)), "wh_encode")
            cur.wait_stream(self.stream)
        mel.record_stream(self.stream)
        out.record_stream(self.stream)
        return out


class HipTask:
    """wh_task: KV caches + workspace of one DecodingTask (whisper/decoding.py:144-176 PyTorchInference).

    No call here synchronises the device: creation touches no device memory, `reset` is a stream-ordered memset and
    the workspace is returned to torch's allocator with `record_stream`.  `HipModel.acquire_task` keeps finished
    tasks (workspace + captured step graph) for the next window of the same shape."""

    def __init__(self, model: HipModel, n_audio: int, n_group: int, max_prefill: int, capture_q: bool = False,
                 stream: Optional[torch.cuda.Stream] = None, two_launch_self: bool = False, two_launch_cross: bool = False,
                 expire_handoffs: bool = False, extra_flags: int = 0, fused_self: bool = False):
        self.model = model
        self.n_audio, self.n_group, self.n_rows = n_audio, n_group, n_audio * n_group
        self.max_prefill = max_prefill
        self.capture_q = capture_q
        self.two_launch_cross = bool(two_launch_cross)
        flags = ((WH_TASK_CAPTURE_Q if capture_q else 0) | (WH_TASK_TWO_LAUNCH_SELF if two_launch_self else 0)
                 | (WH_TASK_TWO_LAUNCH_CROSS if two_launch_cross else 0)
                 | (WH_TASK_EXPIRE_HANDOFFS if expire_handoffs else 0) | (WH_TASK_FUSED_SELF if fused_self else 0)
                 | int(extra_flags) | int(model.debug_task_flags))
        # the calling thread's lane stream (`HipModel.lane`) when it has one, else the engine's: a task a lane creates for
        # itself (e.g. HipInference's two-launch task after a hand-off time-out) must not land on the encoder's stream
        self.stream = stream if stream is not None else model.task_stream()
        with torch.cuda.device(model.device):
            need = lib().wh_task_workspace_bytes(model.handle, n_audio, n_group, max_prefill, flags)
            if need == 0:
                raise HipError("wh_task_workspace_bytes: invalid arguments")
            self.ws = model._alloc(need)
            h = C.c_void_p()
            check(lib().wh_task_create(model.handle, n_audio, n_group, max_prefill, flags, self.ws.data_ptr(),
                                       self.ws.numel(), C.byref(h)), "wh_task_create")
        self.handle = h
        self._cached = False
        self.reset()

    @property
    def cache_key(self):
        return (self.n_audio, self.n_group, self.max_prefill, self.capture_q, self.stream, self.two_launch_cross)

    def close(self):
        """release: into the engine's task cache when it came from `acquire_task`, otherwise destroy"""
        if getattr(self, "handle", None) is None:
            return
        if self._cached and self.model._release_task(self):
            return
        self.destroy()

    def destroy(self):
        if getattr(self, "handle", None):
            # the captured step graph and the workspace may still be in flight on the task's stream: wait for THAT
            # stream only, then hand the memory back
            self.stream.synchronize()
            lib().wh_task_destroy(self.handle)
            self.handle = None
            self.ws = None
            self._align_scratch = None

    def held_bytes(self) -> int:
        """device bytes this task keeps while it idles in its engine's task cache: the workspace + the alignment score slabs"""
        extra = getattr(self, "_align_scratch", None)
        return (0 if self.ws is None else self.ws.numel()) + (0 if extra is None else extra.numel())

    def __del__(self):
        try:
            self.destroy()
        except Exception:   # noqa: BLE001 — interpreter shutdown: module globals (even HipError) may already be None
            pass            # (the library or torch may already be gone)

    @contextlib.contextmanager
    def _call(self):
        """one C-ABI call on the task's stream: the library launches on the calling thread's CURRENT device, so make the
        model's device current for the duration of the call (and restore the caller's afterwards); the task's stream
        waits for the caller's stream before the call and the caller's stream for the task's after it"""
        with torch.cuda.device(self.model.device):
            cur = torch.cuda.current_stream(self.model.device)
            self.stream.wait_stream(cur)
            yield cur
            cur.wait_stream(self.stream)

    def set_audio(self, features: torch.Tensor):
        assert features.is_cuda and features.dtype == self.model.torch_dtype and features.is_contiguous()
        assert features.shape[0] == self.n_audio
        with self._call():
            check(lib().wh_task_set_audio(self.handle, features.data_ptr(), stream_ptr(self.stream)), "wh_task_set_audio")
        features.record_stream(self.stream)

    def prefill(self, tokens: torch.Tensor, sel: Optional[Sequence[int]] = None) -> torch.Tensor:
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.dim() == 2
        assert tokens.shape[0] == self.n_rows and tokens.stride(1) == 1
        T0 = tokens.shape[1]
        n_sel = T0 if sel is None else len(sel)
        logits = torch.empty(self.n_rows, n_sel, self.model.dims.n_vocab, dtype=torch.float32, device=tokens.device)
        sel_arr = None if sel is None else (C.c_int32 * n_sel)(*sel)
        with self._call():
            check(lib().wh_task_prefill(self.handle, tokens.data_ptr(), tokens.stride(0), T0, sel_arr, n_sel,
                                        logits.data_ptr(), stream_ptr(self.stream)), "wh_task_prefill")
        tokens.record_stream(self.stream)
        return logits

    def step(self, last_tokens: torch.Tensor) -> torch.Tensor:
        """last_tokens: int64 view [n_rows] (any stride)."""
        assert last_tokens.is_cuda and last_tokens.dtype == torch.int64 and last_tokens.dim() == 1
        logits = torch.empty(self.n_rows, self.model.dims.n_vocab, dtype=torch.float32, device=last_tokens.device)
        with self._call():
            check(lib().wh_task_step(self.handle, last_tokens.data_ptr(), last_tokens.stride(0), logits.data_ptr(),
                                     stream_ptr(self.stream)), "wh_task_step")
        last_tokens.record_stream(self.stream)
        return logits

    def rearrange(self, source_indices: Sequence[int]):
        arr = (C.c_int32 * len(source_indices))(*[int(i) for i in source_indices])
        assert len(source_indices) == self.n_rows
        with self._call():
            check(lib().wh_task_rearrange(self.handle, arr, stream_ptr(self.stream)), "wh_task_rearrange")

    def reset(self):
        with self._call():
            check(lib().wh_task_reset(self.handle, stream_ptr(self.stream)), "wh_task_reset")

    def set_lag(self, lag: Optional[Sequence[int]]):
        """ragged prompts: row r's sequence is the longest row's shifted left by lag[r] (include/whisper_hip.h)"""
        arr = None
        if lag is not None:
            assert len(lag) == self.n_rows
            arr = (C.c_int32 * len(lag))(*[int(v) for v in lag])
        with self._call():
            check(lib().wh_task_set_lag(self.handle, arr, stream_ptr(self.stream)), "wh_task_set_lag")

    @property
    def position(self) -> int:
        return lib().wh_task_position(self.handle)

    @property
    def fused_cross_attention(self) -> bool:
        """the decode step runs LN -> cross query -> cross attention as one launch (csrc/xattn.hip)"""
        return lib().wh_task_info(self.handle, 0, None) == 1

    @property
    def fused_self_attention(self) -> bool:
        """the decode step runs LN -> QKV -> cache append -> self attention as one launch (csrc/xattn.hip)"""
        return lib().wh_task_info(self.handle, 2, None) == 1

    @property
    def handoff_fallbacks(self) -> int:
        """times wh_task_greedy / wh_task_beam re-ran a loop on the two-launch kernels after a hand-off time-out (the task
        stays on them afterwards: fused_cross_attention / fused_self_attention turn False)"""
        return lib().wh_task_info(self.handle, 4, None)

    def handoff_timeouts(self) -> int:
        """bounded hand-off spins of the fused cross attention that ran out (0 on a healthy device); synchronises"""
        with self._call():
            n = lib().wh_task_info(self.handle, 1, stream_ptr(self.stream))
        return n

    def greedy(self, tokens: torch.Tensor, params: GreedyParams, sot_index: int, no_speech_token: int):
        """tokens: int64 [n_rows][>= sample_begin + max_steps], initial tokens in the first columns.
        Returns (n_tokens, sum_logprobs[n_rows], no_speech_probs[n_rows] | None)."""
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.stride(1) == 1
        dev = tokens.device
        sum_lp = torch.empty(self.n_rows, dtype=torch.float32, device=dev)
        nsp = torch.empty(self.n_rows, dtype=torch.float32, device=dev) if no_speech_token >= 0 else None
        n_out = C.c_int32(0)
        with self._call():
            check(lib().wh_task_greedy(self.handle, C.byref(params), tokens.data_ptr(), tokens.stride(0), sot_index,
                                       no_speech_token, sum_lp.data_ptr(), _ptr(nsp), C.byref(n_out),
                                       stream_ptr(self.stream)), "wh_task_greedy")
        return n_out.value, sum_lp, nsp

    def beam(self, tokens: torch.Tensor, params: BeamParams, sot_index: int, no_speech_token: int):
        """tokens: int64 [2][n_rows][>= sample_begin + max_steps + 1], initial tokens in the first columns of [0].
        Returns (n_tokens, sum_logprobs[n_rows], no_speech_probs[n_rows] | None, finished) where finished =
        (tokens [n_audio][max_candidates][stride], lengths, scores [n_audio][max_candidates], counts [n_audio])."""
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.dim() == 3 and tokens.is_contiguous()
        assert tokens.shape[0] == 2 and tokens.shape[1] == self.n_rows
        dev = tokens.device
        n_audio, mc, stride = self.n_rows // params.beam_size, params.max_candidates, tokens.shape[2]
        sum_lp = torch.empty(self.n_rows, dtype=torch.float32, device=dev)
        nsp = torch.empty(self.n_rows, dtype=torch.float32, device=dev) if no_speech_token >= 0 else None
        fin_tok = torch.zeros(n_audio, mc, stride, dtype=torch.int64, device=dev)
        fin_len = torch.zeros(n_audio, mc, dtype=torch.int32, device=dev)
        fin_score = torch.zeros(n_audio, mc, dtype=torch.float32, device=dev)
        fin_count = torch.zeros(n_audio, dtype=torch.int32, device=dev)
        n_out = C.c_int32(0)
        with self._call():
            check(lib().wh_task_beam(self.handle, C.byref(params), tokens.data_ptr(), stride, sot_index, no_speech_token,
                                     sum_lp.data_ptr(), _ptr(nsp), fin_tok.data_ptr(), fin_len.data_ptr(),
                                     fin_score.data_ptr(), fin_count.data_ptr(), C.byref(n_out),
                                     stream_ptr(self.stream)), "wh_task_beam")
        return n_out.value, sum_lp, nsp, (fin_tok, fin_len, fin_score, fin_count)

    # -- the same loops without a blocked caller (wh_task_greedy_begin / wh_task_beam_begin + wh_task_poll) ----------------
    def greedy_begin(self, tokens: torch.Tensor, params: GreedyParams, sot_index: int, no_speech_token: int) -> "PendingLoop":
        """`greedy` split at the points where the host would wait: queues the prompt pass and the first decision and returns a
        `PendingLoop`; call its `poll()` until it returns the result tuple of `greedy` (None while the loop runs).  Nothing
        here waits for the device, so one host thread can keep several tasks on different streams going."""
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.stride(1) == 1
        dev = tokens.device
        sum_lp = torch.empty(self.n_rows, dtype=torch.float32, device=dev)
        nsp = torch.empty(self.n_rows, dtype=torch.float32, device=dev) if no_speech_token >= 0 else None
        pend = PendingLoop(self, (tokens, params, sum_lp, nsp), lambda n: (n, sum_lp, nsp))
        with torch.cuda.device(self.model.device):
            pend.caller = torch.cuda.current_stream(self.model.device)
            self.stream.wait_stream(pend.caller)
            check(lib().wh_task_greedy_begin(self.handle, C.byref(params), tokens.data_ptr(), tokens.stride(0), sot_index,
                                             no_speech_token, sum_lp.data_ptr(), _ptr(nsp), stream_ptr(self.stream)),
                  "wh_task_greedy_begin")
        return pend

    def beam_begin(self, tokens: torch.Tensor, params: BeamParams, sot_index: int, no_speech_token: int) -> "PendingLoop":
        """`beam` without a blocked caller (see `greedy_begin`); `poll()` returns the result tuple of `beam` at the end"""
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.dim() == 3 and tokens.is_contiguous()
        assert tokens.shape[0] == 2 and tokens.shape[1] == self.n_rows
        dev = tokens.device
        n_audio, mc, stride = self.n_rows // params.beam_size, params.max_candidates, tokens.shape[2]
        sum_lp = torch.empty(self.n_rows, dtype=torch.float32, device=dev)
        nsp = torch.empty(self.n_rows, dtype=torch.float32, device=dev) if no_speech_token >= 0 else None
        fin_tok = torch.zeros(n_audio, mc, stride, dtype=torch.int64, device=dev)
        fin_len = torch.zeros(n_audio, mc, dtype=torch.int32, device=dev)
        fin_score = torch.zeros(n_audio, mc, dtype=torch.float32, device=dev)
        fin_count = torch.zeros(n_audio, dtype=torch.int32, device=dev)
        pend = PendingLoop(self, (tokens, params, sum_lp, nsp, fin_tok, fin_len, fin_score, fin_count),
                           lambda n: (n, sum_lp, nsp, (fin_tok, fin_len, fin_score, fin_count)))
        with torch.cuda.device(self.model.device):
            pend.caller = torch.cuda.current_stream(self.model.device)
            self.stream.wait_stream(pend.caller)
            check(lib().wh_task_beam_begin(self.handle, C.byref(params), tokens.data_ptr(), stride, sot_index, no_speech_token,
                                           sum_lp.data_ptr(), _ptr(nsp), fin_tok.data_ptr(), fin_len.data_ptr(),
                                           fin_score.data_ptr(), fin_count.data_ptr(), stream_ptr(self.stream)),
                  "wh_task_beam_begin")
        return pend

    def bench_kernel(self, kind: int, iters: int) -> Tuple[float, float]:
        """(average ms per launch, algorithmic bytes per launch): `iters` layer-rotated launches replayed from a hipGraph
        and timed with HIP events on the launch stream inside wh_task_bench_kernel (best of 3 replays)"""
        nbytes, ms = C.c_double(0.0), C.c_float(0.0)
        with self._call():
            check(lib().wh_task_bench_kernel(self.handle, kind, iters, C.byref(nbytes), C.byref(ms), stream_ptr(self.stream)), "bench")
        return float(ms.value), nbytes.value

    def cross_qk(self, row: int, layers: Sequence[int], heads: Sequence[int], tok_begin: int, n_tok: int) -> torch.Tensor:
        n = len(layers)
        out = torch.empty(n, n_tok, self.model.dims.n_audio_ctx, dtype=torch.float32, device=self.model.device)
        la = (C.c_int32 * n)(*[int(x) for x in layers])
        ha = (C.c_int32 * n)(*[int(x) for x in heads])
        with self._call():
            check(lib().wh_task_cross_qk(self.handle, row, la, ha, n, tok_begin, n_tok, out.data_ptr(),
                                         stream_ptr(self.stream)), "wh_task_cross_qk")
        return out


    def align_batch(self, layers: Sequence[int], heads: Sequence[int], n_tok: Sequence[int], n_frames: Sequence[int],
                    width: int, row_begin: int, qk_scale: float = 1.0):
        """find_alignment core for every row of the task (wh_task_align_batch + wh_dtw_backtrace_batch).  Returns
        (cost [R][Nmax][Fmax] fp32, jumps int32 [R][Nmax], path_len int32 [R], all on the device): jumps[r][i] = the
        frame at which the DTW path of clip r first reaches text row i (`time_indices[jumps]` of timing.py:226-228);
        path_len[r] = number of path entries, or -1 when the walk met a trace code outside {0, 1, 2} (the reference
        raises ValueError there, timing.py:77 — the caller must).  Nothing is copied to the host here and nothing
        synchronises beyond the C call's own upload of the size arrays."""
        R, P = self.n_rows, len(layers)
        assert len(n_tok) == R and len(n_frames) == R
        Tmax, Fmax = max(n_tok), max(n_frames)
        Nmax = Tmax - 1 - row_begin
        dev = self.model.device
        need = lib().wh_align_batch_scratch_bytes(R, P, Tmax, self.model.dims.n_audio_ctx, Fmax)
        # The score slabs (GBs: QK, softmax and z-norm of every alignment head x token x frame) belong to the TASK, like its
        # K/V workspace: a cached task (`acquire_task`) brings them along, so a second call of the same shape allocates nothing.
        # They used to be a fresh torch.empty per call, recorded on the task's stream: the caching allocator may not hand such a
        # block back before that stream's work is known to be done, so every other call or so paid a multi-GB hipMalloc
        # (round 4: the word-timestamp leg was bimodal, 74-80 ms or 130-150 ms).  Grown, never shrunk; released with the task.
        scratch = getattr(self, "_align_scratch", None)
        if scratch is None or scratch.numel() < need:
            self._align_scratch = None
            scratch = self._align_scratch = self.model._alloc(int(need))
            scratch.record_stream(self.stream)
        cost = torch.empty(R, Nmax, Fmax, dtype=torch.float32, device=dev)
        stride = (Nmax + 1) * (Fmax + 1)
        trace = torch.empty(R, stride, dtype=torch.int8, device=dev)
        jumps = torch.zeros(R, Nmax, dtype=torch.int32, device=dev)
        plen = torch.zeros(R, dtype=torch.int32, device=dev)
        sizes = torch.tensor([[n_tok[r] - 1 - row_begin for r in range(R)], [int(f) for f in n_frames]],
                             dtype=torch.int32).to(dev)                      # before _enter(): ordered on the caller's stream
        arr = lambda v: (C.c_int32 * len(v))(*[int(x) for x in v])
        with self._call():
            check(lib().wh_task_align_batch(self.handle, arr(layers), arr(heads), P, arr(n_tok), arr(n_frames), width, row_begin,
                                            float(qk_scale), cost.data_ptr(), trace.data_ptr(), stride, scratch.data_ptr(),
                                            scratch.numel(), stream_ptr(self.stream)), "wh_task_align_batch")
            check(lib().wh_dtw_backtrace_batch(trace.data_ptr(), stride, sizes[0].data_ptr(), sizes[1].data_ptr(), R, Nmax, Fmax,
                                               jumps.data_ptr(), Nmax, None, 0, plen.data_ptr(), stream_ptr(self.stream)),
                  "wh_dtw_backtrace_batch")
        for t_ in (cost, trace, jumps, plen, sizes):
            t_.record_stream(self.stream)
        return cost, jumps, plen


class PendingLoop:
    """A fused greedy / beam loop that has been begun (`HipTask.greedy_begin` / `beam_begin`).  `poll()` queues further
    decode steps and returns None while the loop runs — it never waits for the device — and the loop's results once it has
    ended (from then on the task takes other calls again).  `wait()` polls to the end, sleeping in between."""

    def __init__(self, task: "HipTask", keep, result):
        self.task, self._keep, self._result = task, keep, result     # _keep: every buffer the loop reads or writes
        self.caller: Optional[torch.cuda.Stream] = None
        self.done = None

    def poll(self):
        if self.done is not None:
            return self.done
        n = C.c_int32(0)
        with torch.cuda.device(self.task.model.device):            # the library launches on the thread's CURRENT device
            rc = lib().wh_task_poll(self.task.handle, C.byref(n))
            if rc == WH_RUNNING:
                return None
            check(rc, "wh_task_poll")
            self.caller.wait_stream(self.task.stream)               # as `_call` does after a blocking call
        self.done = self._result(n.value)
        return self.done

    def wait(self, sleep_s: float = 2e-4):
        import time
        while True:
            r = self.poll()
            if r is not None:
                return r
            time.sleep(sleep_s)


# ---------------------------------------------------------------------------------------------------
# stand-alone kernels
# ---------------------------------------------------------------------------------------------------
def log_mel(audio: torch.Tensor, filters: torch.Tensor) -> torch.Tensor:
    """audio fp32 [n] or [B][n] on the GPU (already padded); filters fp32 [n_mels][201]."""
    require_gpu(audio.device)
    single = audio.dim() == 1
    a = (audio[None] if single else audio).contiguous().float()
    B, n = a.shape
    n_mels = filters.shape[0]
    out = torch.empty(B, n_mels, n // 160, dtype=torch.float32, device=a.device)
    scratch = torch.empty(MEL_SCRATCH_BYTES, dtype=torch.uint8, device=a.device)
    s = torch.cuda.current_stream(a.device)
    check(lib().wh_log_mel(a.data_ptr(), n, B, n_mels, filters.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                           stream_ptr(s)), "wh_log_mel")
    return out[0] if single else out


def median_filter(x: torch.Tensor, width: int) -> torch.Tensor:
    require_gpu(x.device)
    xc = x.contiguous().float()
    n = xc.shape[-1]
    rows = xc.numel() // n if n > 0 else 0
    out = torch.empty_like(xc)
    s = torch.cuda.current_stream(x.device)
    check(lib().wh_median_filter(xc.data_ptr(), out.data_ptr(), rows, n, width, stream_ptr(s)), "wh_median_filter")
    return out


def dtw_trace(x: torch.Tensor) -> torch.Tensor:
    """x fp32 [N][M] cost matrix on the GPU -> int8 trace [N+1][M+1] (dtw_cpu codes)."""
    require_gpu(x.device)
    xc = x.contiguous().float()
    N, M = xc.shape
    trace = torch.empty(N + 1, M + 1, dtype=torch.int8, device=x.device)
    s = torch.cuda.current_stream(x.device)
    check(lib().wh_dtw_trace(xc.data_ptr(), N, M, trace.data_ptr(), stream_ptr(s)), "wh_dtw_trace")
    return trace


def dtw_backtrace(trace: torch.Tensor, want_path: bool = True):
    """trace int8 [N+1][M+1] on the GPU (dtw_trace) -> (jumps int32 [N], path int32 [2][len] | None) on the device:
    the back-trace walk of whisper/timing.py:57-79 done by wh_dtw_backtrace_batch (a batch of one)."""
    require_gpu(trace.device)
    assert trace.dtype == torch.int8 and trace.dim() == 2 and trace.is_contiguous()
    N, M = trace.shape[0] - 1, trace.shape[1] - 1
    dev = trace.device
    sizes = torch.tensor([N, M], dtype=torch.int32).to(dev)
    jumps = torch.zeros(N, dtype=torch.int32, device=dev)
    path = torch.zeros(2, N + M, dtype=torch.int32, device=dev) if want_path else None
    plen = torch.zeros(1, dtype=torch.int32, device=dev)      # always: -1 reports a corrupt trace (timing.py:77)
    s = torch.cuda.current_stream(dev)
    check(lib().wh_dtw_backtrace_batch(trace.data_ptr(), trace.numel(), sizes[0:1].data_ptr(), sizes[1:2].data_ptr(), 1, N, M,
                                       jumps.data_ptr(), N, _ptr(path), N + M, _ptr(plen), stream_ptr(s)),
          "wh_dtw_backtrace_batch")
    n = int(plen.item())
    if n < 0:
        raise ValueError("Unexpected trace[i, j]")          # reference timing.py:77
    if not want_path:
        return jumps, None
    return jumps, path[:, N + M - n:]


def align_matrix(qk: torch.Tensor, n_frames: int, width: int, row_begin: int, row_end: int,
                 qk_scale: float = 1.0) -> torch.Tensor:
    """qk fp32 [heads][tokens][n_audio_ctx] -> DTW cost matrix fp32 [row_end-row_begin][n_frames]
    (softmax over frames, z-norm over tokens, median filter, -mean over heads; whisper/timing.py:207-216)."""
    require_gpu(qk.device)
    q = qk.contiguous().float()
    H, T, Tk = q.shape
    out = torch.empty(row_end - row_begin, n_frames, dtype=torch.float32, device=q.device)
    scratch = torch.empty(2 * H * T * n_frames + 4, dtype=torch.float32, device=q.device)
    s = torch.cuda.current_stream(q.device)
    check(lib().wh_align_matrix(q.data_ptr(), H, T, Tk, n_frames, width, row_begin, row_end, float(qk_scale),
                                out.data_ptr(), scratch.data_ptr(), stream_ptr(s)), "wh_align_matrix")
    return out
