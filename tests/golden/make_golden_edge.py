#!/usr/bin/env python
"""Edge cases of model.transcribe() (whisper/transcribe.py:38-514) from the LIVE reference, CPU fp32, micro-size
synthetic checkpoints (build container only; /root/reference is not on the GPU box):

    python tests/golden/make_golden_edge.py   ->  tests/golden/edge_cases.npz

  empty    : zero samples — the window loop never runs (content_frames = 0)
  short    : 0.31 s of signal — one window, almost all padding
  tail     : 31 s — a full window, then a 1 s tail window (seek arithmetic at the end of the file)
  silence  : 12 s of zeros — the no-speech / logprob-threshold branch (transcribe.py:331-341) on a flat spectrum
Each with and without condition_on_previous_text where it matters; stored: segment count, token ids, seeks, bounds,
no_speech_prob and avg_logprob per segment, language, text length."""
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


def audio(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n).astype(np.float32) * 0.05
    x += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1870 * t)).astype(np.float32)
    return x


CASES = {
    "empty": lambda: np.zeros(0, dtype=np.float32),
    "short": lambda: audio(31, 4960),
    "tail": lambda: audio(32, 16000 * 31),
    "silence": lambda: np.zeros(16000 * 12, dtype=np.float32),
}


def main():
    torch.set_num_threads(8)
    out = {}
    for name in ("micro.en", "micro-v3"):
        key = name.replace(".", "_").replace("-", "_")
        dims = dims_for(name)
        sd = synthetic_state_dict(dims, seed=1)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "m.pt")
            save_checkpoint(path, dims, sd)
            model = whisper.load_model(path, device="cpu")
        for case, make in CASES.items():
            a = make()
            r = model.transcribe(a, temperature=0.0, fp16=False, language="en", sample_len=12,
                                 condition_on_previous_text=True)
            p = f"{key}_{case}"
            out[p + "_n_segments"] = np.array([len(r["segments"])])
            out[p + "_tokens"] = np.array([t for s in r["segments"] for t in s["tokens"]], dtype=np.int64)
            out[p + "_bounds"] = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]], dtype=np.float64).reshape(-1, 3)
            out[p + "_stats"] = np.array([[s["no_speech_prob"], s["avg_logprob"]] for s in r["segments"]], dtype=np.float64).reshape(-1, 2)
            out[p + "_text_len"] = np.array([len(r["text"])])
            print(p, len(r["segments"]), "segments", out[p + "_bounds"].tolist())
        if model.is_multilingual:
            # language detection on an empty clip (transcribe.py:139-155: the first window is all padding)
            r = model.transcribe(np.zeros(0, dtype=np.float32), temperature=0.0, fp16=False, sample_len=4)
            out[f"{key}_empty_detected_language"] = np.array([r["language"]])
            print(key, "empty clip language:", r["language"])
    path = os.path.join(HERE, "edge_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
