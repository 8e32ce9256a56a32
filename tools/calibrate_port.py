"""Calibrate bench.py's `cpu_baseline` ("kind": "port" = the oracle with SDPA attention) against the LIVE reference on the
same host cores (build container only: /root/reference exists here, not on the GPU box).  Same seed-0 weights, same
clips, same protocol as bench.cpu_baseline: log-mel and AudioEncoder on one 30 s window (1 warm-up + repeats, medians),
and greedy decode runs of two lengths through the reference's own DecodingTask (fp32, EOT suppressed) vs
oracle.greedy_decode, from which the fixed cost of a run (prompt pass incl. cross K/V) and the cost per step follow —
at batch 1 and with B clips decoded as one batch (`whisper.decode(model, mel[B])`, decoding.py:713-789).

    python tools/calibrate_port.py [model] [threads] [batch]  ->  tables for BASELINE.md §2b"""
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
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.set_num_threads(threads)
dims = dims_for(name)
sd = synthetic_state_dict(dims, seed=0)
t = np.arange(480000) / 16000.0
clips = []
for b in range(BATCH):
    rng = np.random.default_rng(b)
    clips.append((rng.standard_normal(480000) * 0.05 + 0.2 * np.sin(2 * np.pi * (220 + 20 * b) * t)).astype(np.float32))
K, REPS = 12, 3


def timed(fn, reps=REPS):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return r, statistics.median(ts)


def two_lengths(run, n_short, n_long, reps):
    """(fixed seconds per run, seconds per step, result of the long run)"""
    ts, tl, res = [], [], None
    for _ in range(reps):
        t0 = time.perf_counter(); run(n_short); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); res = run(n_long); tl.append(time.perf_counter() - t0)
    step = (statistics.median(tl) - statistics.median(ts)) / (n_long - n_short)
    return max(statistics.median(ts) - n_short * step, 0.0), step, res


tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
init = list(tok.sot_sequence)
suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.eot]))
rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx, timestamp_begin=tok.timestamp_begin,
                             no_timestamps=tok.no_timestamps, suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)

# ---- live reference
ref = whisper.model.Whisper(whisper.model.ModelDimensions(**dims_dict(dims)))
ref.load_state_dict(sd)
ref.eval()
mel_r, t_mel_r = timed(lambda: whisper.log_mel_spectrogram(clips[0], dims.n_mels))
mels_r = torch.stack([whisper.pad_or_trim(whisper.log_mel_spectrogram(c, dims.n_mels), 3000) for c in clips])
with torch.no_grad():
    feats_r, t_enc_r = timed(lambda: ref.encoder(mels_r[:1]))
    t0 = time.perf_counter(); featsB_r = ref.encoder(mels_r); t_encB_r = time.perf_counter() - t0
    opts = lambda n: whisper.DecodingOptions(language="en", fp16=False, sample_len=n, suppress_tokens=[-1, tok.eot])
    whisper.decode(ref, feats_r, opts(2))
    fx1_r, st1_r, res1_r = two_lengths(lambda n: whisper.decode(ref, feats_r, opts(n)), K, 2 * K, REPS)
    whisper.decode(ref, featsB_r, opts(2))
    fxB_r, stB_r, resB_r = two_lengths(lambda n: whisper.decode(ref, featsB_r, opts(n)), K // 2, 3 * K // 2, 1)
del ref

# ---- the port
om = oracle.OracleModel(dims, sd, sdpa=True)
filt = oracle.mel_filterbank(dims.n_mels)
mel_o, t_mel_o = timed(lambda: oracle.log_mel_spectrogram(clips[0], filt))
mels_o = torch.stack([oracle.log_mel_spectrogram(c, filt) for c in clips])
with torch.no_grad():
    feats_o, t_enc_o = timed(lambda: om.encoder(mels_o[:1]))
    t0 = time.perf_counter(); featsB_o = om.encoder(mels_o); t_encB_o = time.perf_counter() - t0
    oracle.greedy_decode(om, feats_o, init, 2, rules)
    fx1_o, st1_o, res1_o = two_lengths(lambda n: oracle.greedy_decode(om, feats_o, init, n, rules), K, 2 * K, REPS)
    oracle.greedy_decode(om, featsB_o, init, 2, rules)
    fxB_o, stB_o, resB_o = two_lengths(lambda n: oracle.greedy_decode(om, featsB_o, init, n, rules), K // 2, 3 * K // 2, 1)

same1 = res1_r[0].tokens == res1_o["tokens"][0, len(init):].tolist()
sameB = all(resB_r[b].tokens == resB_o["tokens"][b, len(init):].tolist() for b in range(BATCH))
N = 224
print(f"model {name}, {threads} threads, {REPS} repeats (medians), decode runs of {K}/{2 * K} steps (batch 1) and {K // 2}/{3 * K // 2} steps (batch {BATCH})")
print("| stage | live reference | port (oracle, SDPA) | port / reference |")
print("|---|---|---|---|")
print(f"| log-mel (30 s) | {t_mel_r * 1e3:.1f} ms | {t_mel_o * 1e3:.1f} ms | {t_mel_o / t_mel_r:.2f} |")
print(f"| encoder, 1 window | {t_enc_r:.2f} s | {t_enc_o:.2f} s | {t_enc_o / t_enc_r:.2f} |")
print(f"| batch 1: prompt pass (incl. cross K/V) | {fx1_r:.2f} s | {fx1_o:.2f} s | {fx1_o / max(fx1_r, 1e-9):.2f} |")
print(f"| batch 1: decode step | {st1_r * 1e3:.0f} ms | {st1_o * 1e3:.0f} ms | {st1_o / st1_r:.2f} |")
tot_r = t_mel_r + t_enc_r + fx1_r + st1_r * N
tot_o = t_mel_o + t_enc_o + fx1_o + st1_o * N
print(f"| => batch 1 audio-s/s at {N} steps | {30 / tot_r:.3f} | {30 / tot_o:.3f} | {tot_o / tot_r:.2f} (time) |")
print(f"| encoder, {BATCH} windows as one batch (once) | {t_encB_r:.1f} s | {t_encB_o:.1f} s | {t_encB_o / t_encB_r:.2f} |")
print(f"| batch {BATCH}: prompt pass (incl. cross K/V) | {fxB_r:.2f} s | {fxB_o:.2f} s | {fxB_o / max(fxB_r, 1e-9):.2f} |")
print(f"| batch {BATCH}: decode step | {stB_r * 1e3:.0f} ms | {stB_o * 1e3:.0f} ms | {stB_o / stB_r:.2f} |")
totB_r = BATCH * t_mel_r + t_encB_r + fxB_r + stB_r * N
totB_o = BATCH * t_mel_o + t_encB_o + fxB_o + stB_o * N
print(f"| => batch {BATCH} audio-s/s at {N} steps | {30 * BATCH / totB_r:.3f} | {30 * BATCH / totB_o:.3f} | {totB_o / totB_r:.2f} (time) |")
print("token ids equal: batch 1", same1, f"| batch {BATCH}", sameB)
