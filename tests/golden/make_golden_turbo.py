#!/usr/bin/env python
"""BASELINE.json configs[4] dims through the LIVE reference: turbo = large-v3 widths with 32 encoder / 4 decoder
layers (unequal depths: the dims drive both stacks, whisper/model.py:252-277) and `word_timestamps=True`
(whisper/timing.py:163-242 with the default alignment heads of a fresh model, model.py:252-260: every head of the
last half of the DECODER layers = layers 2, 3 x 20 heads).

One 30 s synthetic clip on CPU fp32: encoder slice, teacher-forced logits slice, greedy ids (on noise features),
`find_alignment` on the greedy text tokens and on a fixed sentence, and `transcribe(word_timestamps=True)`.  The seeded weights are
regenerated on the GPU box from `synthetic_state_dict(dims_for("turbo"), seed=4)` (PCG64, bit-reproducible).

    python tests/golden/make_golden_turbo.py      (build container only; a few minutes: the 32-layer encoder on CPU)
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
from whisper.timing import find_alignment  # noqa: E402

from whisper_amd.synthetic import dims_for, save_checkpoint, synthetic_state_dict  # noqa: E402

SEED = 4


def audio(seed, n=480000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n).astype(np.float32) * 0.05
    x += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1870 * t)).astype(np.float32)
    return x


def main():
    torch.set_num_threads(8)
    dims = dims_for("turbo")
    assert (dims.n_audio_layer, dims.n_text_layer) == (32, 4)
    sd = synthetic_state_dict(dims, seed=SEED)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "turbo.pt")
        save_checkpoint(path, dims, sd)
        model = whisper.load_model(path, device="cpu")
    del sd
    assert len(model.encoder.blocks) == 32 and len(model.decoder.blocks) == 4
    out = {}
    a = audio(31)
    mel = whisper.pad_or_trim(whisper.log_mel_spectrogram(a, dims.n_mels), 3000)
    with torch.no_grad():
        feats = model.encoder(mel[None])
    out["enc_slice"] = feats[0, ::50, :24].numpy()
    out["enc_stats"] = np.array([feats.abs().mean().item(), feats.std().item()])
    g = torch.Generator().manual_seed(5)
    toks = torch.randint(0, dims.n_vocab, (2, 9), generator=g)
    with torch.no_grad():
        logits = model.decoder(toks, feats.repeat(2, 1, 1))
    out["tf_tokens"] = toks.numpy()
    out["tf_logits_slice"] = logits[:, :, ::997].numpy()
    out["tf_logits_argmax"] = logits.argmax(-1).numpy()
    # greedy ids on 2 rows of seeded noise features (the reference accepts encoded features, decoding.py:654-657): the
    # 32-layer random-init encoder maps every clip to nearly the same features and the 4-layer decoder then echoes one
    # token; on noise features the decode visits ~20 distinct ids per row (tests/test_wide_gpu.py uses the same rows)
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(2, dims.n_audio_ctx, dims.n_audio_state, generator=g)
    rs = whisper.decode(model, noise, whisper.DecodingOptions(language="en", fp16=False, sample_len=24))
    width = max(len(r.tokens) for r in rs)
    out["greedy_tokens"] = np.array([r.tokens + [-1] * (width - len(r.tokens)) for r in rs], dtype=np.int64)
    out["greedy_stats"] = np.array([[r.avg_logprob, r.no_speech_prob] for r in rs])
    res = rs[0]
    tokenizer = whisper.tokenizer.get_tokenizer(model.is_multilingual, num_languages=model.num_languages,
                                                language="en", task="transcribe")
    out["alignment_heads"] = model.alignment_heads.to_dense().numpy()
    for tag, text in (("fixed", tokenizer.encode(" hello world this is a test of word level timing")),
                      ("greedy", [t for t in res.tokens if t < tokenizer.eot])):
        al = find_alignment(model, tokenizer, text, mel, 3000)
        out[f"align_{tag}_tokens"] = np.array(text, dtype=np.int64)
        out[f"align_{tag}_start"] = np.array([w.start for w in al])
        out[f"align_{tag}_end"] = np.array([w.end for w in al])
        out[f"align_{tag}_prob"] = np.array([w.probability for w in al])
    r = model.transcribe(a, temperature=0.0, fp16=False, language="en", sample_len=16, word_timestamps=True,
                         condition_on_previous_text=True)
    out["tr_n_segments"] = np.array([len(r["segments"])])
    out["tr_tokens"] = np.array([t for s in r["segments"] for t in s["tokens"]], dtype=np.int64)
    out["tr_seg_bounds"] = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
    out["tr_word_times"] = np.array([[w["start"], w["end"]] for s in r["segments"] for w in s["words"]]).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "turbo_dims.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print(out["greedy_tokens"], out["align_fixed_start"], out["tr_word_times"][:6])


if __name__ == "__main__":
    main()
