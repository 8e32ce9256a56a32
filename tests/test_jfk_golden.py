"""BASELINE.json configs[0] — "tiny.en, tests/jfk.flac, greedy" — against the LIVE reference's outputs stored in
tests/golden/jfk_tiny_en.npz (made by tests/golden/make_golden_jfk.py: the reference's own speech sample, decoded by
the native FLAC path, tiny.en dims with the seeded synthetic weights because no checkpoint exists offline).

CPU part: the oracle restatement on real speech (mel, encoder, greedy ids).  GPU part: the HIP path end to end through
the public API — log-mel, model.transcribe() with word timestamps, decode() — token ids exact."""
import os

import numpy as np
import pytest
import torch

import oracle
from whisper_amd.tokenizer import get_tokenizer

J = np.load(os.path.join(os.path.dirname(__file__), "golden", "jfk_tiny_en.npz"))
AUDIO = J["jfk_pcm16"].astype(np.float32) / 32768.0


def test_oracle_on_jfk():
    dims = oracle.dims_for("tiny.en")
    filt = oracle.mel_filterbank(dims.n_mels)
    mel = oracle.log_mel_spectrogram(AUDIO, filt)
    assert mel.shape == (80, 1100)
    assert np.abs(mel[:, ::25].numpy() - J["mel_slice"]).max() < 1e-4
    assert abs(mel.mean().item() - J["mel_stats"][0]) < 1e-5 and abs(mel.max().item() - J["mel_stats"][3]) < 1e-4
    om = oracle.OracleModel(dims, oracle.synthetic_state_dict(dims, seed=2))
    padded = torch.nn.functional.pad(mel, (0, 3000 - mel.shape[1]))
    with torch.no_grad():
        feats = om.encoder(padded[None])
    assert np.abs(feats[0, ::50, :32].numpy() - J["enc_slice"]).max() < 2e-4
    tok = get_tokenizer(False)
    init = list(tok.sot_sequence_including_notimestamps)
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev,
                                                         tok.sot_lm, tok.no_speech]))
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx,
                                 timestamp_begin=None, no_timestamps=tok.no_timestamps, suppress_blank=True,
                                 blank_token=tok.encode(" ")[0], suppress_tokens=suppress, no_speech=tok.no_speech)
    with torch.no_grad():
        out = oracle.greedy_decode(om, feats, init, 40, rules)
    toks = out["tokens"][0, len(init):].tolist()
    toks = toks[: toks.index(tok.eot)] if tok.eot in toks else toks
    assert toks == J["greedy_nots_tokens"].tolist()
    assert abs(out["no_speech_probs"][0] - J["greedy_nots_stats"][1]) < 1e-6


@pytest.mark.gpu
def test_hip_path_on_jfk(gpu_device, tmp_path):
    import whisper_amd
    from whisper_amd.synthetic import dims_for, save_checkpoint, synthetic_state_dict
    dims = dims_for("tiny.en")
    path = str(tmp_path / "tiny.en.pt")
    save_checkpoint(path, dims, synthetic_state_dict(dims, seed=2))
    model = whisper_amd.load_model(path, device=gpu_device)
    mel = whisper_amd.log_mel_spectrogram(AUDIO, dims.n_mels, device=gpu_device)
    assert np.abs(mel[:, ::25].cpu().numpy() - J["mel_slice"]).max() < 1e-4            # log-mel atol
    r = model.transcribe(AUDIO, temperature=0.0, fp16=False, language="en", word_timestamps=True,
                         condition_on_previous_text=True, no_speech_threshold=None, logprob_threshold=None,
                         compression_ratio_threshold=None)
    assert len(r["segments"]) == int(J["n_segments"][0])
    assert [t for s in r["segments"] for t in s["tokens"]] == J["tokens"].tolist()       # greedy token ids: exact
    bounds = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
    assert np.array_equal(bounds[:, 0], J["seg_bounds"][:, 0])
    assert np.abs(bounds[:, 1:] - J["seg_bounds"][:, 1:]).max() < 1e-6
    assert np.abs(np.array([s["avg_logprob"] for s in r["segments"]]) - J["seg_logprob"]).max() < 1e-3
    words = np.array([[w["start"], w["end"]] for s in r["segments"] for w in s["words"]]).reshape(-1, 2)
    assert words.shape == J["word_times"].shape
    assert np.abs(words - J["word_times"]).max() < 1e-6                                  # fp32 engine: frame indices exact
    res = whisper_amd.decode(model, whisper_amd.pad_or_trim(mel, 3000),
                             whisper_amd.DecodingOptions(language="en", fp16=False, without_timestamps=True, sample_len=40))
    assert res.tokens == J["greedy_nots_tokens"].tolist()
    assert abs(res.avg_logprob - J["greedy_nots_stats"][0]) < 1e-4
    assert abs(res.no_speech_prob - J["greedy_nots_stats"][1]) < 1e-5 + 1e-3 * J["greedy_nots_stats"][1]
    # the fp16 engine on real speech: same first tokens
    res16 = whisper_amd.decode(model, whisper_amd.pad_or_trim(mel, 3000),
                               whisper_amd.DecodingOptions(language="en", fp16=True, without_timestamps=True, sample_len=40))
    assert res16.tokens[:4] == res.tokens[:4]


def _cached_checkpoint(name: str):
    """where the reference's load_model keeps a released checkpoint (whisper/__init__.py:126-135), or None"""
    from whisper_amd.registry import MODEL_URLS
    default = os.path.join(os.path.expanduser("~"), ".cache")
    path = os.path.join(os.getenv("XDG_CACHE_HOME", default), "whisper", os.path.basename(MODEL_URLS[name]))
    return path if os.path.isfile(path) else None


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny.en", "tiny", "base", "large-v3-turbo", "large-v3"])
def test_real_checkpoint_if_present(gpu_device, name):
    """BASELINE.json configs[0] unmodified — the reference's own tests/test_transcribe.py:24-39 on its own speech sample —
    whenever a RELEASED checkpoint sits in ~/.cache/whisper/ (there is no network here, so normally it does not and the
    test skips; the seeded-weights version above always runs).  Both engines: the fp32 strict one and the fp16 one."""
    import whisper_amd
    from whisper_amd.tokenizer import get_tokenizer
    path = _cached_checkpoint(name)
    if path is None:
        pytest.skip(f"no released {name} checkpoint under ~/.cache/whisper (no network in this environment)")
    model = whisper_amd.load_model(name, device=gpu_device)
    language = "en" if name.endswith(".en") else None
    for fp16 in (False, True):
        result = model.transcribe(AUDIO, language=language, temperature=0.0, word_timestamps=True, fp16=fp16)
        assert result["language"] == "en"
        assert result["text"] == "".join([s["text"] for s in result["segments"]])
        transcription = result["text"].lower()
        assert "my fellow americans" in transcription
        assert "your country" in transcription
        assert "do for you" in transcription
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages)
        all_tokens = [t for s in result["segments"] for t in s["tokens"]]
        assert tokenizer.decode(all_tokens) == result["text"]
        assert tokenizer.decode_with_timestamps(all_tokens).startswith("<|0.00|>")
        timing_checked = False
        for segment in result["segments"]:
            for timing in segment["words"]:
                assert timing["start"] < timing["end"]
                if timing["word"].strip(" ,") == "Americans":
                    assert timing["start"] <= 1.8
                    assert timing["end"] >= 1.8
                    timing_checked = True
        assert timing_checked
