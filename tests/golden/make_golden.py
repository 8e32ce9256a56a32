#!/usr/bin/env python
"""Generate tests/golden/*.npz from the LIVE reference (openai/whisper imported from /root/reference under the
two import shims of tests/shims/).  Run in the build container only:

    python tests/golden/make_golden.py

Inputs are seeded (numpy PCG64 audio, whisper_amd.synthetic checkpoints), so only outputs are stored.
Everything the reference computes here is CPU fp32 (its own CPU path, whisper/transcribe.py:128-136)."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "shims"), "/root/reference"]

import whisper  # noqa: E402  (the reference)
from whisper.timing import dtw_cpu, find_alignment, median_filter  # noqa: E402

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
    # ---- log-mel (audio.py:110-157) ---------------------------------------------------------------
    for n_mels in (80, 128):
        a = audio(100 + n_mels, 16000 * 4 + 37)
        out[f"mel{n_mels}_4s"] = whisper.log_mel_spectrogram(a, n_mels).numpy()
        out[f"mel{n_mels}_4s_pad"] = whisper.log_mel_spectrogram(a, n_mels, padding=1600).numpy()[:, -40:]
    ab = np.stack([audio(7, 32000), audio(8, 32000) * 3.0])
    out["mel80_batch"] = whisper.log_mel_spectrogram(torch.from_numpy(ab), 80).numpy()
    out["mel_filters_80"] = whisper.audio.mel_filters("cpu", 80).numpy()
    out["mel_filters_128"] = whisper.audio.mel_filters("cpu", 128).numpy()

    for name in ("micro.en", "micro-v3"):
        key = name.replace(".", "_").replace("-", "_")
        dims = dims_for(name)
        sd = synthetic_state_dict(dims, seed=1)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, f"{name}.pt")
            save_checkpoint(path, dims, sd)
            model = whisper.load_model(path, device="cpu")
        a = audio(3)
        mel = whisper.log_mel_spectrogram(a, dims.n_mels)
        mel = whisper.pad_or_trim(mel, 3000)
        with torch.no_grad():
            feats = model.encoder(mel[None])
        out[f"{key}_enc_slice"] = feats[0, ::50, :24].numpy()
        out[f"{key}_enc_absmean"] = np.array([feats.abs().mean().item(), feats.std().item()])
        # teacher-forced logits (model.py:227-249)
        g = torch.Generator().manual_seed(5)
        toks = torch.randint(0, dims.n_vocab, (2, 9), generator=g)
        with torch.no_grad():
            logits = model.decoder(toks, feats.repeat(2, 1, 1))
        out[f"{key}_tf_tokens"] = toks.numpy()
        out[f"{key}_tf_logits_slice"] = logits[:, :, ::997].numpy()
        out[f"{key}_tf_logits_argmax"] = logits.argmax(-1).numpy()
        # greedy decode through the reference's DecodingTask (decoding.py:508-789)
        for tag, kw in (("ts", {}), ("nots", {"without_timestamps": True}),
                        ("prompt", {"prompt": [1000, 2000, 3000], "prefix": [400, 500]})):
            res = whisper.decode(model, mel, whisper.DecodingOptions(language="en", fp16=False, sample_len=20, **kw))
            out[f"{key}_greedy_{tag}_tokens"] = np.array(res.tokens, dtype=np.int64)
            out[f"{key}_greedy_{tag}_stats"] = np.array([res.avg_logprob, res.no_speech_prob, res.compression_ratio])
        res = whisper.decode(model, mel, whisper.DecodingOptions(language="en", fp16=False, sample_len=12, beam_size=3))
        out[f"{key}_beam3_tokens"] = np.array(res.tokens, dtype=np.int64)
        out[f"{key}_beam3_stats"] = np.array([res.avg_logprob, res.no_speech_prob])
        res = whisper.decode(model, mel, whisper.DecodingOptions(language="en", fp16=False, sample_len=10, beam_size=2, patience=2.0))
        out[f"{key}_beam2p_tokens"] = np.array(res.tokens, dtype=np.int64)
        if model.is_multilingual:
            lang_tok, probs = whisper.detect_language(model, mel)
            out[f"{key}_lang_token"] = np.array([int(lang_tok)])
            top = sorted(probs.items(), key=lambda kv: -kv[1])[:5]
            out[f"{key}_lang_top5_p"] = np.array([p for _, p in top])
            out[f"{key}_lang_top5"] = np.array([c for c, _ in top])
        # word alignment (timing.py:163-242) on a fixed token list
        tokenizer = whisper.tokenizer.get_tokenizer(model.is_multilingual, num_languages=model.num_languages,
                                                    language="en", task="transcribe")
        text_tokens = tokenizer.encode(" hello world this is a test of word level timing")
        al = find_alignment(model, tokenizer, text_tokens, mel, 3000)
        out[f"{key}_align_tokens"] = np.array(text_tokens, dtype=np.int64)
        out[f"{key}_align_start"] = np.array([w.start for w in al])
        out[f"{key}_align_end"] = np.array([w.end for w in al])
        out[f"{key}_align_prob"] = np.array([w.probability for w in al])
        # long-form transcribe (transcribe.py:38-514): 50 s, two windows, word timestamps
        a50 = np.concatenate([audio(21), audio(22, 320000)])
        r = model.transcribe(a50, temperature=0.0, fp16=False, language="en", sample_len=16, word_timestamps=True,
                             condition_on_previous_text=True)
        out[f"{key}_tr_n_segments"] = np.array([len(r["segments"])])
        out[f"{key}_tr_tokens"] = np.array([t for s in r["segments"] for t in s["tokens"]], dtype=np.int64)
        out[f"{key}_tr_seg_bounds"] = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
        out[f"{key}_tr_word_times"] = np.array([[w["start"], w["end"]] for s in r["segments"] for w in s["words"]])

    # ---- timing known answers (tests/test_timing.py generators) ----------------------------------------
    rng = np.random.default_rng(0)
    x = rng.standard_normal((57, 211)).astype(np.float32)
    out["dtw_in"] = x
    out["dtw_path"] = dtw_cpu(x.astype(np.float64))
    xm = torch.from_numpy(rng.standard_normal((3, 5, 97)).astype(np.float32))
    out["median_in"] = xm.numpy()
    for w in (3, 7):
        out[f"median_w{w}"] = median_filter(xm, w).numpy()

    path = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
