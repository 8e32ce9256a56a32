"""developer helper: wall-clock of the public API on BASELINE.json configs 4 and 5 (per-GPU share), synthetic weights.
  config 4 share: large-v3, 8 clips x beam 5 (40 rows), decode() through the generic Inference seam + host beam search
  config 5 share: turbo, 4 clips, transcribe(word_timestamps=True)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_amd
from whisper_amd.model import ModelDimensions, Whisper
from whisper_amd.synthetic import dims_dict, dims_for, synthetic_state_dict

dev = torch.device("cuda:0")


def make(name):
    dims = dims_for(name)
    sd = synthetic_state_dict(dims, seed=0, device=dev)
    return Whisper(ModelDimensions(**dims_dict(dims)), sd, device=dev), dims


def audio(b, n=480000):
    rng = np.random.default_rng(b)
    t = np.arange(n) / 16000.0
    return (rng.standard_normal(n) * 0.05 + 0.2 * np.sin(2 * np.pi * (220 + 20 * b) * t)).astype(np.float32)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "beam"):
    model, dims = make("large-v3")
    mel = whisper_amd.log_mel_spectrogram(torch.from_numpy(np.stack([audio(b) for b in range(8)])).to(dev), dims.n_mels)
    for N, kw in ((32, dict(beam_size=5)), (32, dict())):
        opts = whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=N, suppress_tokens="-1,50257", **kw)
        whisper_amd.decode(model, mel, opts)                      # warm-up (engine pack, graph capture)
        t0 = sync()
        res = whisper_amd.decode(model, mel, opts)
        t1 = sync()
        print(f"large-v3 B=8 {kw or 'greedy'} sample_len={N}: {1e3 * (t1 - t0):.1f} ms  "
              f"({1e3 * (t1 - t0) / N:.2f} ms/step incl. encoder)  tokens[0][:8]={res[0].tokens[:8]}", flush=True)
    del model
    torch.cuda.empty_cache()
if which in ("all", "words"):
    model, dims = make("turbo")
    for wt in (False, True):
        a = audio(3)
        kw = dict(language="en", temperature=0.0, fp16=True, word_timestamps=wt, sample_len=48, condition_on_previous_text=False)
        whisper_amd.transcribe(model, a, **kw)
        t0 = sync()
        out = [whisper_amd.transcribe(model, audio(b), **kw) for b in range(4)]
        t1 = sync()
        nseg = sum(len(o["segments"]) for o in out)
        nw = sum(len(s.get("words", [])) for o in out for s in o["segments"])
        print(f"turbo 4 x 30 s transcribe(word_timestamps={wt}): {1e3 * (t1 - t0):.1f} ms total, {nseg} segments, {nw} words", flush=True)

if which in ("all", "batch"):
    model, dims = make("turbo")
    files = [audio(b, 480000 * 2 + 16000 * b) for b in range(8)]          # 8 files of 60-67 s: 3 windows each
    kw = dict(language="en", temperature=0.0, fp16=True, sample_len=48, condition_on_previous_text=False,
              no_speech_threshold=None, logprob_threshold=None, compression_ratio_threshold=None)
    whisper_amd.transcribe(model, files[0], **kw)
    t0 = sync()
    seq = [whisper_amd.transcribe(model, a, **kw) for a in files]
    t1 = sync()
    whisper_amd.transcribe_batch(model, files[:2], **kw)
    t2 = sync()
    bat = whisper_amd.transcribe_batch(model, files, **kw)
    t3 = sync()
    same = all([s["tokens"] for s in a["segments"]] == [s["tokens"] for s in b["segments"]] for a, b in zip(seq, bat))
    secs = sum(len(a) for a in files) / 16000.0
    print(f"turbo, 8 files ({secs:.0f} audio-s): transcribe() one by one {1e3 * (t1 - t0):.0f} ms = {secs / (t1 - t0):.0f} audio-s/s; "
          f"transcribe_batch {1e3 * (t3 - t2):.0f} ms = {secs / (t3 - t2):.0f} audio-s/s; identical tokens: {same}", flush=True)
