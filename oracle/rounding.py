"""fp16 rounding model of the HIP engine on the oracle.  TEST INFRASTRUCTURE (like the rest of oracle/).

`RoundingOracle` is OracleModel's decoder (whisper/model.py:227-249 restated in oracle/model.py) with an fp16 round trip
inserted at the named sites where the HIP fp16 engine stores or consumes fp16 (fp32 residual stream and
accumulators; fp16 weights incl. the LayerNorm affine folded into the next projection, projection inputs, q / k / v, the
K/V caches, attention outputs, MLP activations, the final hidden state).  It answers two questions the engine itself cannot
(one packed blob per dtype, no per-stage switch):
  * which site carries the engine's logit error on a given input (one site at a time against the plain fp32 oracle);
  * what error the engine SHOULD show if rounding at those sites is all there is (all sites together) — the GPU tests
    hold the measured error against this figure, so a defective kernel cannot hide inside a loose bound.
Used by tests/test_wide_gpu.py::test_turbo_dims_vs_oracle and tools/attribute_fp16_error.py.
"""
from typing import Dict, Iterable

import torch
import torch.nn.functional as F

from .model import OracleModel

SITES = ["ln_fold", "proj_in", "qkv_self", "self_out", "cross_q", "cross_kv", "cross_out", "mlp_h", "final_h"]


def r16(t):
    return t.half().float()


class RoundingOracle(OracleModel):
    """decoder of oracle/model.py with fp16 round trips at `sites` (a set of SITES)"""

    def __init__(self, dims, sd, sites):
        super().__init__(dims, sd)
        self.sites = set(sites)
        if "ln_fold" in self.sites:
            # the engine folds every decoder LayerNorm's affine into the projection that follows it and rounds the product
            # to fp16 (WH_WEIGHTS_DEC_LN_FOLDED): W' = fp16(W diag(g)), b' = b + W beta (fp32)
            sd2 = dict(self.sd)
            for i in range(dims.n_text_layer):
                p = f"decoder.blocks.{i}"
                for ln, projs in ((".attn_ln", [".attn.query", ".attn.key", ".attn.value"]), (".cross_attn_ln", [".cross_attn.query"]),
                                  (".mlp_ln", [".mlp.0"])):
                    g, beta = self.sd[p + ln + ".weight"], self.sd[p + ln + ".bias"]
                    for pr in projs:
                        W = self.sd[p + pr + ".weight"]
                        b = self.sd.get(p + pr + ".bias")
                        sd2[p + pr + ".weight"] = r16(W * g[None, :])
                        sd2[p + pr + ".bias"] = (b if b is not None else 0) + W @ beta
                    sd2[p + ln + ".weight"] = torch.ones_like(g)
                    sd2[p + ln + ".bias"] = torch.zeros_like(beta)
            self.sd = sd2

    def _s(self, name, t):
        return r16(t) if name in self.sites else t

    def decoder_hidden(self, tokens, xa, cache=None, keep_qk=False):
        d, sd = self.dims, self.sd
        R, T = tokens.shape
        offset = 0 if cache is None or cache["self_k"][0] is None else cache["self_k"][0].shape[1]
        x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][offset: offset + T]
        for i in range(d.n_text_layer):
            p = f"decoder.blocks.{i}"
            h = self._s("proj_in", self._ln(x, p + ".attn_ln"))
            q = self._s("qkv_self", self._lin(h, p + ".attn.query"))
            k = self._s("qkv_self", self._lin(h, p + ".attn.key"))
            v = self._s("qkv_self", self._lin(h, p + ".attn.value"))
            if cache is not None:
                if cache["self_k"][i] is not None:
                    k = torch.cat([cache["self_k"][i], k], dim=1)
                    v = torch.cat([cache["self_v"][i], v], dim=1)
                cache["self_k"][i], cache["self_v"][i] = k, v
            a, _ = self._attend(q, k, v, d.n_text_head, offset, need_qk=False)
            x = x + self._lin(self._s("self_out", a), p + ".attn.out")
            h = self._s("proj_in", self._ln(x, p + ".cross_attn_ln"))
            if cache is not None and cache["cross_k"][i] is not None:
                ck, cv = cache["cross_k"][i], cache["cross_v"][i]
            else:
                ck = self._s("cross_kv", self._lin(xa, p + ".cross_attn.key"))
                cv = self._s("cross_kv", self._lin(xa, p + ".cross_attn.value"))
                if cache is not None:
                    cache["cross_k"][i], cache["cross_v"][i] = ck, cv
            cq = self._s("cross_q", self._lin(h, p + ".cross_attn.query"))
            a, _ = self._attend(cq, ck, cv, d.n_text_head, None, need_qk=False)
            x = x + self._lin(self._s("cross_out", a), p + ".cross_attn.out")
            h = self._s("proj_in", self._ln(x, p + ".mlp_ln"))
            x = x + self._lin(self._s("mlp_h", F.gelu(self._lin(h, p + ".mlp.0"))), p + ".mlp.2")
        return self._s("final_h", self._ln(x, "decoder.ln"))


def site_table(dims, sd, toks: torch.Tensor, feats: torch.Tensor, sites: Iterable = None) -> Dict[str, Dict[str, float]]:
    """teacher-forced logits of `toks` (R, T) on `feats`: per site (and "ALL") max / rms |dlogit| of the rounding model against
    the plain fp32 oracle"""
    base = OracleModel(dims, sd)
    out = {}
    with torch.no_grad():
        want = base.decoder(toks, feats)
        for group in ([[s] for s in (sites if sites is not None else SITES)] + [list(SITES)]):
            got = RoundingOracle(dims, sd, group).decoder(toks, feats)
            d = (got - want).abs()
            out["ALL" if len(group) > 1 else group[0]] = {"max": float(d.max()), "rms": float((d.double() ** 2).mean().sqrt())}
    return out
