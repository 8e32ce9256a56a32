#!/usr/bin/env python
"""Per-segment previous-text prompts through the LIVE reference, one segment at a time (its DecodingTask shares one
initial_tokens tuple between all rows, decoding.py:719): the inputs of tests/test_api_gpu.py::
test_ragged_prompts_equal_single_row_decodes — 6 clips, prompts of 0 / 1 / 5 / 17 / 60 / 150 tokens, greedy, fp32 —
so that ONE batched call with prompts of different lengths on the HIP path is pinned to the reference row by row.

    python tests/golden/make_golden_prompts.py      (build container only)
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


def audio(seed, n=480000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n).astype(np.float32) * 0.05
    x += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1870 * t)).astype(np.float32)
    return x


def prompts():
    rng = np.random.default_rng(7)
    return [None, [1234], rng.integers(300, 40000, 5).tolist(), rng.integers(300, 40000, 17).tolist(),
            rng.integers(300, 40000, 60).tolist(), rng.integers(300, 40000, 150).tolist()]


def main():
    torch.set_num_threads(8)
    out = {}
    for name in ("micro.en", "micro-v3"):
        key = name.replace(".", "_").replace("-", "_")
        dims = dims_for(name)
        sd = synthetic_state_dict(dims, seed=1)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, f"{name}.pt")
            save_checkpoint(path, dims, sd)
            model = whisper.load_model(path, device="cpu")
        rows, stats = [], []
        for i, p in enumerate(prompts()):
            mel = whisper.pad_or_trim(whisper.log_mel_spectrogram(audio(50 + i), dims.n_mels), 3000)
            r = whisper.decode(model, mel, whisper.DecodingOptions(language="en", fp16=False, sample_len=14, prompt=p))
            rows.append(r.tokens)
            stats.append([r.avg_logprob, r.no_speech_prob])
        width = max(len(r) for r in rows)
        out[f"{key}_tokens"] = np.array([r + [-1] * (width - len(r)) for r in rows], dtype=np.int64)
        out[f"{key}_stats"] = np.array(stats)
    np.savez_compressed(os.path.join(HERE, "prompts_micro.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v[:2].tolist() if "tokens" in k else "")


if __name__ == "__main__":
    main()
