"""Model forward on CPU — restates whisper/model.py:39-249 functionally (no nn.Module, explicit KV cache).

Also holds the synthetic-checkpoint generator shared by tests and bench: no Whisper checkpoint exists
offline, so both the reference and whisper_amd load seeded weights saved in the reference's checkpoint
format {"dims", "model_state_dict"} (whisper/__init__.py:154-156).
"""
import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

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


def synthetic_state_dict(dims, seed: int = 0, device="cpu", fp16_exact: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded weights with the reference's parameter names/shapes (whisper/model.py:174-249).
    Scaled so the network is not degenerate (default nn.Embedding init makes the tied-logit model echo
    its last token).  With fp16_exact every value is representable in fp16, so
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
        for a in (["attn", "cross_attn"] if cross else ["attn"]):
            linear(f"{prefix}.{a}.query", n, n)
            linear(f"{prefix}.{a}.key", n, n, bias=False)
            linear(f"{prefix}.{a}.value", n, n)
            linear(f"{prefix}.{a}.out", n, n)
            lnorm(f"{prefix}.{a}_ln", n)
        linear(f"{prefix}.mlp.0", 4 * n, n)
        linear(f"{prefix}.mlp.2", n, 4 * n)
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


class OracleModel:
    """Functional Whisper on CPU tensors.  `dtype` is the activation dtype (float32 = the reference's CPU
    path, whisper/transcribe.py:128-136)."""

    def __init__(self, dims, sd: Dict[str, torch.Tensor], dtype=torch.float32):
        self.dims, self.dtype = dims, dtype
        self.sd = {k: v.detach().cpu().float() for k, v in sd.items()}

    # -- primitives -----------------------------------------------------------------------------
    def _ln(self, x, p):      # model.py:39-41
        return F.layer_norm(x.float(), (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5).to(x.dtype)

    def _lin(self, x, p):     # model.py:44-50
        b = self.sd.get(p + ".bias")
        return F.linear(x, self.sd[p + ".weight"].to(x.dtype), None if b is None else b.to(x.dtype))

    def _attend(self, q, k, v, n_head, causal_offset: Optional[int]):
        """model.py:114-139 manual path: scale d_head**-0.25 on q and k, fp32 softmax.  causal_offset = number of
        cached positions preceding the queries, or None for no mask.  Returns (out, qk)."""
        B, Tq, Dm = q.shape
        scale = (Dm // n_head) ** -0.25
        qh = q.view(B, Tq, n_head, -1).permute(0, 2, 1, 3)
        kh = k.view(k.shape[0], k.shape[1], n_head, -1).permute(0, 2, 1, 3)
        vh = v.view(v.shape[0], v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        qk = (qh * scale) @ (kh * scale).transpose(-1, -2)
        if causal_offset is not None and Tq > 1:      # model.py:125: the T=1 step attends to everything cached
            Tk = k.shape[1]
            mask = torch.full((Tq, Tk), -np.inf).triu_(1 + causal_offset)
            qk = qk + mask
        qk = qk.float()
        w = F.softmax(qk, dim=-1).to(q.dtype)
        return (w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2), qk

    # -- encoder (model.py:188-204) ---------------------------------------------------------------
    def encoder(self, mel: torch.Tensor) -> torch.Tensor:
        d, sd = self.dims, self.sd
        x = mel.to(self.dtype)
        x = F.gelu(F.conv1d(x, sd["encoder.conv1.weight"].to(x.dtype), sd["encoder.conv1.bias"].to(x.dtype), padding=1))
        x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"].to(x.dtype), sd["encoder.conv2.bias"].to(x.dtype), stride=2, padding=1))
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == sd["encoder.positional_embedding"].shape, "incorrect audio shape"
        x = (x + sd["encoder.positional_embedding"]).to(x.dtype)
        for i in range(d.n_audio_layer):
            p = f"encoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            a, _ = self._attend(self._lin(h, p + ".attn.query"), self._lin(h, p + ".attn.key"),
                                self._lin(h, p + ".attn.value"), d.n_audio_head, None)
            x = x + self._lin(a, p + ".attn.out")
            h = self._ln(x, p + ".mlp_ln")
            x = x + self._lin(F.gelu(self._lin(h, p + ".mlp.0")), p + ".mlp.2")
        return self._ln(x, "encoder.ln_post")

    # -- decoder (model.py:227-249 with the KV cache of model.py:310-341 made explicit) -------------
    def new_cache(self) -> dict:
        return {"self_k": [None] * self.dims.n_text_layer, "self_v": [None] * self.dims.n_text_layer,
                "cross_k": [None] * self.dims.n_text_layer, "cross_v": [None] * self.dims.n_text_layer, "qk": None}

    def decoder(self, tokens: torch.Tensor, xa: torch.Tensor, cache: Optional[dict] = None,
                keep_qk: bool = False) -> torch.Tensor:
        """tokens (R, T) int64; xa (B, 1500, D) with R % B == 0 (rows r use audio r // (R//B)).
        With a cache, `tokens` are the new positions only.  Returns fp32 logits (R, T, V)."""
        d, sd = self.dims, self.sd
        R, T = tokens.shape
        offset = 0 if cache is None or cache["self_k"][0] is None else cache["self_k"][0].shape[1]
        x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][offset: offset + T]
        x = x.to(xa.dtype)
        group = R // xa.shape[0]
        qks = []
        for i in range(d.n_text_layer):
            p = f"decoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            k, v = self._lin(h, p + ".attn.key"), self._lin(h, p + ".attn.value")
            if cache is not None:
                if cache["self_k"][i] is not None:
                    k = torch.cat([cache["self_k"][i], k], dim=1)      # model.py:332
                    v = torch.cat([cache["self_v"][i], v], dim=1)
                cache["self_k"][i], cache["self_v"][i] = k, v
            a, _ = self._attend(self._lin(h, p + ".attn.query"), k, v, d.n_text_head, offset)
            x = x + self._lin(a, p + ".attn.out")
            h = self._ln(x, p + ".cross_attn_ln")
            if cache is not None and cache["cross_k"][i] is not None:   # model.py:106-109
                ck, cv = cache["cross_k"][i], cache["cross_v"][i]
            else:
                ck, cv = self._lin(xa, p + ".cross_attn.key"), self._lin(xa, p + ".cross_attn.value")
                if cache is not None:
                    cache["cross_k"][i], cache["cross_v"][i] = ck, cv
            if group > 1:
                ck, cv = ck.repeat_interleave(group, 0), cv.repeat_interleave(group, 0)
            a, qk = self._attend(self._lin(h, p + ".cross_attn.query"), ck, cv, d.n_text_head, None)
            if keep_qk:
                qks.append(qk)
            x = x + self._lin(a, p + ".cross_attn.out")
            h = self._ln(x, p + ".mlp_ln")
            x = x + self._lin(F.gelu(self._lin(h, p + ".mlp.0")), p + ".mlp.2")
        x = self._ln(x, "decoder.ln")
        if keep_qk and cache is not None:
            cache["qk"] = qks
        elif keep_qk:
            self.last_qk = qks
        return (x @ sd["decoder.token_embedding.weight"].to(x.dtype).T).float()      # model.py:245-247

    def rearrange(self, cache: dict, source_indices: List[int]) -> None:
        """decoding.py:172-176: only the self-attention caches are gathered"""
        if source_indices != list(range(len(source_indices))):
            for i in range(self.dims.n_text_layer):
                cache["self_k"][i] = cache["self_k"][i][source_indices]
                cache["self_v"][i] = cache["self_v"][i][source_indices]
