"""Synthetic seeded checkpoints and the dims table of the released Whisper models.

No Whisper checkpoint exists offline, so tests and bench.py run on seeded weights that have the exact
parameter names / shapes of the reference (whisper/model.py:174-249) and are saved in its checkpoint format
{"dims", "model_state_dict"} (whisper/__init__.py:154-156) — `whisper_amd.load_model(path)` and the
reference's `whisper.load_model(path)` both read them.
"""
import math
from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch

# dims of the released checkpoints (they live in each checkpoint's "dims")
_DIMS = {
    # name: (n_mels, D, heads, enc_layers, dec_layers, n_vocab)
    "tiny.en": (80, 384, 6, 4, 4, 51864), "tiny": (80, 384, 6, 4, 4, 51865),
    "base.en": (80, 512, 8, 6, 6, 51864), "base": (80, 512, 8, 6, 6, 51865),
    "small.en": (80, 768, 12, 12, 12, 51864), "small": (80, 768, 12, 12, 12, 51865),
    "medium.en": (80, 1024, 16, 24, 24, 51864), "medium": (80, 1024, 16, 24, 24, 51865),
    "large-v1": (80, 1280, 20, 32, 32, 51865), "large-v2": (80, 1280, 20, 32, 32, 51865),
    "large-v3": (128, 1280, 20, 32, 32, 51866), "large": (128, 1280, 20, 32, 32, 51866),
    "large-v3-turbo": (128, 1280, 20, 32, 4, 51866), "turbo": (128, 1280, 20, 32, 4, 51866),
    # reduced-depth shapes for fast tests (same widths / vocab as the real ones)
    "micro.en": (80, 384, 6, 2, 2, 51864), "micro": (80, 384, 6, 2, 2, 51865),
    "micro-v3": (128, 384, 6, 2, 2, 51866),
    # large-v3 widths at 2 + 2 layers: exercises the D = 1280 kernel shapes against the CPU oracle in seconds
    "wide-v3": (128, 1280, 20, 2, 2, 51866),
    # the other released widths (base / small / medium) at 2 + 2 layers, for kernel-shape coverage
    "w512": (80, 512, 8, 2, 2, 51865), "w768": (80, 768, 12, 2, 2, 51865), "w1024": (80, 1024, 16, 2, 2, 51865),
}


def dims_for(name: str) -> SimpleNamespace:
    m, d, h, le, ld, v = _DIMS[name]
    return SimpleNamespace(n_mels=m, n_audio_ctx=1500, n_audio_state=d, n_audio_head=h, n_audio_layer=le,
                           n_vocab=v, n_text_ctx=448, n_text_state=d, n_text_head=h, n_text_layer=ld)


def dims_dict(dims) -> dict:
    return {k: int(getattr(dims, k)) for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head",
                                               "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
                                               "n_text_head", "n_text_layer")}


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """whisper/model.py:62-68"""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


DEC_OUT_GAIN = 2.5          # decoder attn.out / cross_attn.out / mlp.2
DEC_CROSS_QK_GAIN = 3.0     # decoder cross_attn.query / cross_attn.key


def synthetic_state_dict(dims, seed: int = 0, device="cpu", fp16_exact: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded weights with the reference's parameter names/shapes (whisper/model.py:174-249).
    Scaled so the network is not degenerate: default nn.Embedding init makes the tied-logit model echo its last
    token, and so does any init in which the token embedding dominates the final
    residual.  The decoder's block outputs (`attn.out`, `cross_attn.out`, `mlp.2`) therefore carry gain
    DEC_OUT_GAIN and the cross-attention query / key projections gain DEC_CROSS_QK_GAIN (peaked, audio-dependent
    attention): a 24-step greedy decode then visits >= 10 distinct ids in 20 steps on every model shape and mode used in the tests,
    with text and timestamp tokens interleaved, instead of repeating one id (measured with the CPU oracle).  With fp16_exact every value is representable in fp16, so
    the fp32 reference and the fp16 kernels see identical weights.  CPU generation uses numpy's PCG64
    (bit-reproducible across machines); GPU generation uses torch's generator on that device."""
    device = torch.device(device)
    if device.type == "cpu":
        rng = np.random.default_rng(seed)
        def randn(*shape):
            return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32))
    else:
        gen = torch.Generator(device=device); gen.manual_seed(seed)
        def randn(*shape):
            return torch.randn(*shape, generator=gen, device=device, dtype=torch.float32)
    q = (lambda t: t.half().float()) if fp16_exact else (lambda t: t)
    sd: Dict[str, torch.Tensor] = {}
    D, Dt = dims.n_audio_state, dims.n_text_state

    def linear(prefix, n_out, n_in, bias=True, gain=0.7):
        sd[prefix + ".weight"] = q(randn(n_out, n_in) * (gain / math.sqrt(n_in)))
        if bias:
            sd[prefix + ".bias"] = q(randn(n_out) * 0.02)

    def lnorm(prefix, n):
        sd[prefix + ".weight"] = q(1.0 + 0.05 * randn(n))
        sd[prefix + ".bias"] = q(0.05 * randn(n))

    def block(prefix, n, cross):
        g_out = DEC_OUT_GAIN if cross else 0.7            # `cross` == decoder block
        for a in (["attn", "cross_attn"] if cross else ["attn"]):
            g_qk = DEC_CROSS_QK_GAIN if a == "cross_attn" else 0.7
            linear(f"{prefix}.{a}.query", n, n, gain=g_qk)
            linear(f"{prefix}.{a}.key", n, n, bias=False, gain=g_qk)
            linear(f"{prefix}.{a}.value", n, n)
            linear(f"{prefix}.{a}.out", n, n, gain=g_out)
            lnorm(f"{prefix}.{a}_ln", n)
        linear(f"{prefix}.mlp.0", 4 * n, n)
        linear(f"{prefix}.mlp.2", n, 4 * n, gain=g_out)
        lnorm(f"{prefix}.mlp_ln", n)

    sd["encoder.conv1.weight"] = q(randn(D, dims.n_mels, 3) * (1.0 / math.sqrt(3 * dims.n_mels)))
    sd["encoder.conv1.bias"] = q(randn(D) * 0.02)
    sd["encoder.conv2.weight"] = q(randn(D, D, 3) * (1.0 / math.sqrt(3 * D)))
    sd["encoder.conv2.bias"] = q(randn(D) * 0.02)
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, D).to(device)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", D, False)
    lnorm("encoder.ln_post", D)
    sd["decoder.token_embedding.weight"] = q(randn(dims.n_vocab, Dt) * 0.05)
    sd["decoder.positional_embedding"] = q(randn(dims.n_text_ctx, Dt) * 0.05)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", Dt, True)
    lnorm("decoder.ln", Dt)
    return sd


def save_checkpoint(path: str, dims, sd: Dict[str, torch.Tensor]) -> None:
    """reference checkpoint format, whisper/__init__.py:150-156"""
    torch.save({"dims": dims_dict(dims), "model_state_dict": {k: v.cpu() for k, v in sd.items()}}, path)
