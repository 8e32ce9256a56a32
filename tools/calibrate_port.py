"""Calibrate bench.py's `cpu_baseline` ("kind": "port" = the oracle) against the LIVE reference on the same host cores
(build container only: /root/reference exists here, not on the GPU box).  Same seed-0 large-v3 weights, same clip, same
protocol as bench.cpu_baseline (1 warm-up + 3 repeats, medians): log-mel, AudioEncoder on one 30 s window, 12 greedy
decode steps through the reference's own DecodingTask (fp32, EOT suppressed) vs oracle.greedy_decode.
    python tools/calibrate_port.py [model] [threads]  ->  a table for BASELINE.md"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "shims"), "/root/reference"]
import numpy as np
import torch

import oracle
import whisper  # the reference
from whisper_amd.synthetic import dims_dict, dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
torch.set_num_threads(threads)
dims = dims_for(name)
sd = synthetic_state_dict(dims, seed=0)
rng = np.random.default_rng(0)
t = np.arange(480000) / 16000.0
audio = (rng.standard_normal(480000) * 0.05 + 0.2 * np.sin(2 * np.pi * 220 * t)).astype(np.float32)
K, REPS = 12, 3


def timed(fn):
    fn()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return r, statistics.median(ts)


# ---- live reference
ref = whisper.model.Whisper(whisper.model.ModelDimensions(**dims_dict(dims)))
ref.load_state_dict(sd)
ref.eval()
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
mel_r, t_mel_r = timed(lambda: whisper.log_mel_spectrogram(audio, dims.n_mels))
mel_r = whisper.pad_or_trim(mel_r, 3000)
with torch.no_grad():
    feats_r, t_enc_r = timed(lambda: ref.encoder(mel_r[None]))
opts = whisper.DecodingOptions(language="en", fp16=False, sample_len=K, suppress_tokens=[-1, tok.eot])
with torch.no_grad():
    res_r, t_dec_r = timed(lambda: whisper.decode(ref, feats_r, opts))
del ref

# ---- the port
om = oracle.OracleModel(dims, sd)
filt = oracle.mel_filterbank(dims.n_mels)
mel_o, t_mel_o = timed(lambda: oracle.log_mel_spectrogram(audio, filt))
with torch.no_grad():
    feats_o, t_enc_o = timed(lambda: om.encoder(mel_o[None]))
init = list(tok.sot_sequence)
suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.eot]))
rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx, timestamp_begin=tok.timestamp_begin,
                             no_timestamps=tok.no_timestamps, suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)
with torch.no_grad():
    res_o, t_dec_o = timed(lambda: oracle.greedy_decode(om, feats_o, init, K, rules))
same = res_r[0].tokens == res_o["tokens"][0, len(init):].tolist()
print(f"model {name}, {threads} threads, {REPS} repeats (medians)")
print(f"| stage | live reference | port (oracle) | port / reference |")
print(f"|---|---|---|---|")
print(f"| log-mel (30 s) | {t_mel_r * 1e3:.1f} ms | {t_mel_o * 1e3:.1f} ms | {t_mel_o / t_mel_r:.2f} |")
print(f"| encoder (1 window) | {t_enc_r:.2f} s | {t_enc_o:.2f} s | {t_enc_o / t_enc_r:.2f} |")
print(f"| {K} greedy steps (prefill + {K - 1} single-token steps + filters) | {t_dec_r:.2f} s = {t_dec_r / K * 1e3:.0f} ms/step | {t_dec_o:.2f} s = {t_dec_o / K * 1e3:.0f} ms/step | {t_dec_o / t_dec_r:.2f} |")
tot_r = t_mel_r + t_enc_r + t_dec_r / K * 224
tot_o = t_mel_o + t_enc_o + t_dec_o / K * 224
print(f"| => audio-s/s at 224 steps | {30 / tot_r:.3f} | {30 / tot_o:.3f} | {tot_o / tot_r:.2f} (time) |")
print("token ids equal:", same)
