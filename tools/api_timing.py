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
