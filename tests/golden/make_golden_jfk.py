#!/usr/bin/env python
"""BASELINE.json configs[0] ("tiny.en, tests/jfk.flac, greedy decode on the reference CPU path") as a golden fixture.

No tiny.en checkpoint exists offline, so the weights are the seeded synthetic tiny.en of whisper_amd.synthetic; the
audio is the reference's own tests/jfk.flac, decoded by whisper_amd.load_audio (native FLAC path: no ffmpeg in this
image; the decoder verifies the stream's MD5 signature).  The LIVE reference then runs its CPU fp32 path on it:
log-mel, encoder features, and model.transcribe(temperature=0).  Stored: the 16 kHz mono audio as int16 (so the GPU
box, which has no /root/reference, can replay it), and the reference's outputs.

    python tests/golden/make_golden_jfk.py      (build container only)
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

from whisper_amd.audio import load_audio  # noqa: E402
from whisper_amd.synthetic import dims_for, save_checkpoint, synthetic_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    audio = load_audio("/root/reference/tests/jfk.flac")
    pcm = np.round(audio * 32768.0).astype(np.int16)
    assert np.array_equal(pcm.astype(np.float32) / 32768.0, audio)
    out = {"jfk_pcm16": pcm}
    dims = dims_for("tiny.en")
    sd = synthetic_state_dict(dims, seed=2)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tiny.en.pt")
        save_checkpoint(path, dims, sd)
        model = whisper.load_model(path, device="cpu")
    mel = whisper.log_mel_spectrogram(audio, dims.n_mels)
    out["mel_slice"] = mel[:, ::25].numpy()
    out["mel_stats"] = np.array([mel.mean().item(), mel.std().item(), mel.min().item(), mel.max().item()])
    with torch.no_grad():
        feats = model.encoder(whisper.pad_or_trim(mel, 3000)[None])
    out["enc_slice"] = feats[0, ::50, :32].numpy()
    r = model.transcribe(audio, temperature=0.0, fp16=False, language="en", word_timestamps=True,
                         condition_on_previous_text=True, no_speech_threshold=None, logprob_threshold=None,
                         compression_ratio_threshold=None)
    out["n_segments"] = np.array([len(r["segments"])])
    out["tokens"] = np.array([t for s in r["segments"] for t in s["tokens"]], dtype=np.int64)
    out["seg_bounds"] = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
    out["seg_logprob"] = np.array([s["avg_logprob"] for s in r["segments"]])
    out["word_times"] = np.array([[w["start"], w["end"]] for s in r["segments"] for w in s["words"]])
    res = whisper.decode(model, whisper.pad_or_trim(mel, 3000),
                         whisper.DecodingOptions(language="en", fp16=False, without_timestamps=True, sample_len=40))
    out["greedy_nots_tokens"] = np.array(res.tokens, dtype=np.int64)
    out["greedy_nots_stats"] = np.array([res.avg_logprob, res.no_speech_prob])
    path = os.path.join(HERE, "jfk_tiny_en.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
