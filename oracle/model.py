"""Model forward on CPU — restates whisper/model.py:39-249 functionally (no nn.Module, explicit KV cache).

The synthetic-checkpoint generator (whisper_amd/synthetic.py) is re-exported here for the tests.
"""
import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from whisper_amd.synthetic import (dims_dict, dims_for, save_checkpoint, sinusoids,  # noqa: F401,E402
                                   synthetic_state_dict)


class OracleModel:
    """Functional Whisper on CPU tensors.  `dtype` is the activation dtype (float32 = the reference's CPU
    path, whisper/transcribe.py:128-136)."""

    def __init__(self, dims, sd: Dict[str, torch.Tensor], dtype=torch.float32, sdpa: bool = False):
        """sdpa: compute attention with torch's scaled_dot_product_attention wherever the scores themselves are not
        asked for — what the reference does by default (model.py:124-128, `SDPA_AVAILABLE and use_sdpa`).  The tests
        keep the explicit form (model.py:130-139, the definition); bench.py's CPU baseline times the fused form,
        because that is what a user of the reference's CPU path runs."""
        self.dims, self.dtype, self.sdpa = dims, dtype, sdpa
        self.sd = {k: v.detach().cpu().float() for k, v in sd.items()}

    # -- primitives -----------------------------------------------------------------------------
    def _ln(self, x, p):      # model.py:39-41
        return F.layer_norm(x.float(), (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5).to(x.dtype)

    def _lin(self, x, p):     # model.py:44-50
        b = self.sd.get(p + ".bias")
        return F.linear(x, self.sd[p + ".weight"].to(x.dtype), None if b is None else b.to(x.dtype))

    def _attend(self, q, k, v, n_head, causal_offset: Optional[int], need_qk: bool = True):
        """model.py:114-139 manual path: scale d_head**-0.25 on q and k, fp32 softmax.  causal_offset = number of
        cached positions preceding the queries, or None for no mask.  Returns (out, qk); with `sdpa` and no need for
        the scores, the fused form of model.py:124-128 (qk = None)."""
        B, Tq, Dm = q.shape
        scale = (Dm // n_head) ** -0.25
        qh = q.view(B, Tq, n_head, -1).permute(0, 2, 1, 3)
        kh = k.view(k.shape[0], k.shape[1], n_head, -1).permute(0, 2, 1, 3)
        vh = v.view(v.shape[0], v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        if self.sdpa and not need_qk and (causal_offset is None or Tq == 1 or causal_offset == 0):
            a = F.scaled_dot_product_attention(qh, kh, vh, is_causal=causal_offset is not None and Tq > 1)
            return a.permute(0, 2, 1, 3).flatten(start_dim=2), None
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
                                self._lin(h, p + ".attn.value"), d.n_audio_head, None, need_qk=False)
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
        x = self.decoder_hidden(tokens, xa, cache, keep_qk)
        return (x @ self.sd["decoder.token_embedding.weight"].to(x.dtype).T).float()      # model.py:245-247

    def decoder_hidden(self, tokens: torch.Tensor, xa: torch.Tensor, cache: Optional[dict] = None,
                       keep_qk: bool = False) -> torch.Tensor:
        """model.py:227-244: everything up to and including `self.ln(x)`; (R, T, D).  Split from `decoder` for
        oracle/condition.py, which edits rows of the tied embedding between the two halves."""
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
            a, _ = self._attend(self._lin(h, p + ".attn.query"), k, v, d.n_text_head, offset, need_qk=False)
            x = x + self._lin(a, p + ".attn.out")
            h = self._ln(x, p + ".cross_attn_ln")
            if cache is not None and cache["cross_k"][i] is not None:   # model.py:106-109
                ck, cv = cache["cross_k"][i], cache["cross_v"][i]
            else:
                ck, cv = self._lin(xa, p + ".cross_attn.key"), self._lin(xa, p + ".cross_attn.value")
                if cache is not None:
                    cache["cross_k"][i], cache["cross_v"][i] = ck, cv
            cq = self._lin(h, p + ".cross_attn.query")
            if group > 1:
                # decoding.py:734 repeat_interleave()s the audio features so that row r attends to audio r // group.
                # The same sums without materialising `group` copies of K / V (2 x 300 MB per layer and step at 40 rows
                # of large-v3): the rows of one audio become extra query positions of that audio — cross attention has
                # no mask, every (query, key) product and every softmax row is unchanged.
                Bq = xa.shape[0]
                a, qk = self._attend(cq.reshape(Bq, group * T, -1), ck, cv, d.n_text_head, None, need_qk=keep_qk)
                a = a.reshape(R, T, -1)
                if qk is not None:
                    qk = qk.view(Bq, d.n_text_head, group, T, -1).permute(0, 2, 1, 3, 4).reshape(R, d.n_text_head, T, -1)
            else:
                a, qk = self._attend(cq, ck, cv, d.n_text_head, None, need_qk=keep_qk)
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
        return x

    def rearrange(self, cache: dict, source_indices: List[int]) -> None:
        """decoding.py:172-176: only the self-attention caches are gathered"""
        if source_indices != list(range(len(source_indices))):
            for i in range(self.dims.n_text_layer):
                cache["self_k"][i] = cache["self_k"][i][source_indices]
                cache["self_v"][i] = cache["self_v"][i][source_indices]
