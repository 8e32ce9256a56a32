#!/usr/bin/env python
"""large-v3 WIDTHS (D = 1280, 20 heads, 51866 tokens; 2 + 2 layers = `wide-v3` of whisper_amd.synthetic) through the
LIVE reference: whisper.decode on 8 rows of seeded random audio features (the reference accepts already-encoded
features, decoding.py:654-657), greedy, fp32, CPU — exactly the inputs of tests/test_wide_gpu.py::
test_wide_fused_greedy_vs_oracle, so the D = 1280 kernel shapes are pinned to the reference itself and not only to
the oracle.  Also one beam-search decode (beam 5) of the first 2 rows.

    python tests/golden/make_golden_wide.py      (build container only; ~1 minute)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "shims"), "/root/reference"]

import whisper  # noqa: E402  (the reference)

from whisper_amd.synthetic import dims_for, save_checkpoint, synthetic_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    dims = dims_for("wide-v3")
    sd = synthetic_state_dict(dims, seed=3)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "wide-v3.pt")
        save_checkpoint(path, dims, sd)
        model = whisper.load_model(path, device="cpu")
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(8, dims.n_audio_ctx, dims.n_audio_state, generator=g)
    out = {}
    res = whisper.decode(model, feats, whisper.DecodingOptions(language="en", fp16=False, sample_len=20))
    width = max(len(r.tokens) for r in res)
    out["greedy_tokens"] = np.array([r.tokens + [-1] * (width - len(r.tokens)) for r in res], dtype=np.int64)
    out["greedy_stats"] = np.array([[r.avg_logprob, r.no_speech_prob] for r in res])
    beams = [whisper.decode(model, feats[i], whisper.DecodingOptions(language="en", fp16=False, sample_len=10, beam_size=5))
             for i in range(2)]
    width = max(len(r.tokens) for r in beams)
    out["beam5_tokens"] = np.array([r.tokens + [-1] * (width - len(r.tokens)) for r in beams], dtype=np.int64)
    out["beam5_stats"] = np.array([r.avg_logprob for r in beams])
    np.savez_compressed(os.path.join(HERE, "wide_v3.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print(out["greedy_tokens"][:2], out["beam5_tokens"])


if __name__ == "__main__":
    main()
