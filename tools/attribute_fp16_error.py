"""tools/attribute_fp16_error.py — developer tool (CPU only; not part of the product).

Which rounding site of the fp16 engine carries its logit error on a given input?  (turbo dims,
seed-4 weights, the same weights give max |dlogit| 0.136 on plain noise features (`_feats(seed=33)`) and 0.017 on
features with a per-clip offset — attribute it.)

The engine cannot be switched to fp32 one stage at a time (it is one packed blob per dtype), so the attribution runs on the
oracle: `RoundingOracle` is oracle.OracleModel with an fp16 round trip inserted at the named sites where the HIP fp16
engine stores or consumes fp16 (fp32 residual stream and accumulators; fp16 weights, projection inputs,
q / k / v, K/V caches, attention outputs, MLP activations, final hidden state).  One site at a time against the plain fp32
oracle on the same teacher-forced tokens, then all sites together (what the engine should measure, up to summation order).

  python tools/attribute_fp16_error.py [turbo|large-v3] [rows] [positions]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle.model import OracleModel  # noqa: E402

from oracle.rounding import SITES, RoundingOracle  # noqa: E402


def feats_plain(dims, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, dims.n_audio_ctx, dims.n_audio_state, generator=g).half().float()


def feats_offset(dims, n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, dims.n_audio_ctx, dims.n_audio_state, generator=g)
            + 3.0 * torch.randn(n, 1, dims.n_audio_state, generator=g)).half().float()


def attribute(name="turbo", rows=4, positions=12, seed=4, out=print):
    dims = oracle.dims_for(name)
    sd = oracle.synthetic_state_dict(dims, seed=seed)
    base = OracleModel(dims, sd)
    g = torch.Generator().manual_seed(4)
    toks = torch.randint(0, dims.n_vocab, (rows, positions), generator=g)
    res = {}
    for label, feats in (("plain noise features (_feats seed 33)", feats_plain(dims, rows, 33)),
                         ("per-clip offset features (_offset_feats seed 12)", feats_offset(dims, rows, 12))):
        with torch.no_grad():
            want = base.decoder(toks, feats)
            # how peaked the cross attention is on this input: mean over (layer, row, head, query) of the largest weight
            base.decoder(toks, feats, keep_qk=True)
            peak = float(torch.stack([F.softmax(qk, -1).amax(-1).mean() for qk in base.last_qk]).mean())
            out(f"{name} seed {seed}, {label}: |logit| max {float(want.abs().max()):.2f}, mean largest cross-attention weight {peak:.3f}")
            table = {}
            for sites in [[s] for s in SITES] + [SITES]:
                got = RoundingOracle(dims, sd, sites).decoder(toks, feats)
                d = (got - want).abs()
                key = "+".join(sites) if len(sites) == 1 else "ALL"
                table[key] = (float(d.max()), float((d.double() ** 2).mean().sqrt()))
                out(f"    {key:10s} max |dlogit| {table[key][0]:.4f}   rms {table[key][1]:.5f}")
        res[label] = {"mean_largest_cross_weight": peak, "sites": table}
    return res


if __name__ == "__main__":
    attribute(sys.argv[1] if len(sys.argv) > 1 else "turbo", int(sys.argv[2]) if len(sys.argv) > 2 else 4,
              int(sys.argv[3]) if len(sys.argv) > 3 else 12)
