#!/usr/bin/env python
"""Beam search (beam 3, 8 steps) of three clips through the LIVE reference, one clip at a time (its BeamSearchDecoder
cannot take more than one audio per call): the inputs of tests/test_api_gpu.py::
test_batched_beam_vs_oracle, where the HIP path decodes the three clips as ONE task.

    python tests/golden/make_golden_beam.py      (build container only)
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
        for i in range(3):
            mel = whisper.pad_or_trim(whisper.log_mel_spectrogram(audio(40 + i), dims.n_mels), 3000)
            r = whisper.decode(model, mel, whisper.DecodingOptions(language="en", fp16=False, sample_len=8, beam_size=3))
            rows.append(r.tokens)
            stats.append(r.avg_logprob)
        width = max(len(r) for r in rows)
        out[f"{key}_tokens"] = np.array([r + [-1] * (width - len(r)) for r in rows], dtype=np.int64)
        out[f"{key}_avg_logprob"] = np.array(stats)
    np.savez_compressed(os.path.join(HERE, "beam_micro.npz"), **out)
    for k, v in out.items():
        print(k, v.tolist())


if __name__ == "__main__":
    main()
