"""End-to-end parity of the public API on the GPU: whisper_amd.load_model / decode / detect_language /
find_alignment / transcribe against (a) outputs of the LIVE reference stored in tests/golden (token ids exact in
the fp32 strict-parity engine) and (b) the CPU oracle for cases the reference cannot run (batched beam)."""
import os

import numpy as np
import pytest
import torch

import oracle
import whisper_amd
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, save_checkpoint, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.npz"))


def audio(seed, n=480000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n).astype(np.float32) * 0.05
    x += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1870 * t)).astype(np.float32)
    return x


@pytest.fixture(scope="module", params=["micro.en", "micro-v3"])
def setup(request, gpu_device, tmp_path_factory):
    name = request.param
    key = name.replace(".", "_").replace("-", "_")
    dims = dims_for(name)
    sd = synthetic_state_dict(dims, seed=1)
    path = str(tmp_path_factory.mktemp("ckpt") / f"{name}.pt")
    save_checkpoint(path, dims, sd)
    model = whisper_amd.load_model(path, device=gpu_device)
    mel = whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(3), dims.n_mels, device=gpu_device), 3000)
    return key, dims, sd, model, mel


def test_mel_golden(gpu_device):
    """HIP log-mel vs the reference's torch.stft path (stored outputs): atol 1e-4"""
    for n_mels in (80, 128):
        a = audio(100 + n_mels, 16000 * 4 + 37)
        got = whisper_amd.log_mel_spectrogram(a, n_mels, device=gpu_device).cpu().numpy()
        assert np.abs(got - G[f"mel{n_mels}_4s"]).max() < 1e-4
        got = whisper_amd.log_mel_spectrogram(a, n_mels, padding=1600, device=gpu_device).cpu().numpy()[:, -40:]
        assert np.abs(got - G[f"mel{n_mels}_4s_pad"]).max() < 1e-4
    ab = np.stack([audio(7, 32000), audio(8, 32000) * 3.0])
    got = whisper_amd.log_mel_spectrogram(torch.from_numpy(ab), 80, device=gpu_device).cpu().numpy()
    assert np.abs(got - G["mel80_batch"]).max() < 1e-4        # global max over the batch (audio.py:155)


def test_encoder_and_logits_golden(setup):
    key, dims, sd, model, mel = setup
    feats = model.encoder(mel[None].float())
    assert feats.dtype == torch.float32
    assert np.abs(feats[0, ::50, :24].cpu().numpy() - G[f"{key}_enc_slice"]).max() < 3e-4
    toks = torch.from_numpy(G[f"{key}_tf_tokens"]).to(mel.device)
    logits = model.decoder(toks, feats.repeat(2, 1, 1))
    assert logits.dtype == torch.float32 and logits.shape == (2, 9, dims.n_vocab)
    assert np.abs(logits[:, :, ::997].cpu().numpy() - G[f"{key}_tf_logits_slice"]).max() < 1e-3   # north_star: 1e-3
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), G[f"{key}_tf_logits_argmax"])
    # fp16 engine: same call with fp16 activations
    feats16 = model.encoder(mel[None].half())
    assert feats16.dtype == torch.float16
    lg16 = model.decoder(toks, feats16.repeat(2, 1, 1))
    assert np.abs(lg16[:, :, ::997].cpu().numpy() - G[f"{key}_tf_logits_slice"]).max() < 6e-2


@pytest.mark.parametrize("tag,kw", [("ts", {}), ("nots", {"without_timestamps": True}),
                                    ("prompt", {"prompt": [1000, 2000, 3000], "prefix": [400, 500]})])
def test_greedy_decode_golden(setup, tag, kw):
    """fused device-side greedy loop, fp32 engine: token ids exactly those of the reference's DecodingTask"""
    key, dims, sd, model, mel = setup
    res = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=20, **kw))
    assert res.tokens == G[f"{key}_greedy_{tag}_tokens"].tolist()
    avg, nsp, cr = G[f"{key}_greedy_{tag}_stats"]
    assert abs(res.avg_logprob - avg) < 1e-3
    assert abs(res.no_speech_prob - nsp) < max(1e-7, 2e-3 * nsp)
    assert abs(res.compression_ratio - cr) < 1e-9
    assert res.audio_features.shape == (dims.n_audio_ctx, dims.n_audio_state)


def test_generic_loop_equals_fused(setup):
    """a user-supplied (no-op) LogitFilter forces the host-driven loop (per-step wh_task_step + vectorised
    filters): same tokens and logprobs as the fused loop"""
    from whisper_amd.decoding import DecodingTask, LogitFilter

    class Noop(LogitFilter):
        def apply(self, logits, tokens):
            return None

    key, dims, sd, model, mel = setup
    opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=20)
    fused = DecodingTask(model, opts).run(mel[None])[0]
    task = DecodingTask(model, opts)
    task.logit_filters.append(Noop())
    generic = task.run(mel[None])[0]
    assert generic.tokens == fused.tokens == G[f"{key}_greedy_ts_tokens"].tolist()
    assert abs(generic.avg_logprob - fused.avg_logprob) < 1e-4
    assert abs(generic.no_speech_prob - fused.no_speech_prob) < 1e-6


def test_beam_decode_golden(setup):
    key, dims, sd, model, mel = setup
    res = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=12, beam_size=3))
    assert res.tokens == G[f"{key}_beam3_tokens"].tolist()
    assert abs(res.avg_logprob - G[f"{key}_beam3_stats"][0]) < 1e-3
    res = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=10,
                                                                     beam_size=2, patience=2.0))
    assert res.tokens == G[f"{key}_beam2p_tokens"].tolist()


def test_batched_beam_vs_oracle(setup, gpu_device):
    """n_audio = 3 x beam 3 in ONE task (the reference raises here): must equal each audio
    decoded alone by the CPU oracle"""
    key, dims, sd, model, mel = setup
    mels = torch.stack([whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(40 + i), dims.n_mels,
                                                                                device=gpu_device), 3000) for i in range(3)])
    results = whisper_amd.decode(model, mels, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=8, beam_size=3))
    om = oracle.OracleModel(dims, sd)
    multilingual = dims.n_vocab >= 51865
    tok = get_tokenizer(multilingual, num_languages=dims.n_vocab - 51765 - int(multilingual), language="en", task="transcribe")
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]))
    init = list(tok.sot_sequence)
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, timestamp_begin=tok.timestamp_begin,
                                 no_timestamps=tok.no_timestamps, suppress_tokens=suppress, blank_token=tok.encode(" ")[0],
                                 no_speech=tok.no_speech)
    filt = oracle.mel_filterbank(dims.n_mels)
    for i, res in enumerate(results):
        omel = oracle.log_mel_spectrogram(audio(40 + i), filt)
        with torch.no_grad():
            out = oracle.beam_decode(om, om.encoder(omel[None]), init, 8, rules, 3)
        body, lp = oracle.decoding.rank_candidates(out["candidates"][0], len(init), tok.eot)
        assert res.tokens == body, i
    # ... and the live reference decoding each clip alone (tests/golden/make_golden_beam.py)
    Bm = np.load(os.path.join(os.path.dirname(__file__), "golden", "beam_micro.npz"))
    for i, res in enumerate(results):
        assert res.tokens == [t for t in Bm[f"{key}_tokens"][i].tolist() if t >= 0], i


FP16_LOGIT_BOUND = 6e-2      # |logit(fp16 engine) - logit(fp32 oracle)|, asserted in test_kernels_gpu / test_wide_gpu


def _oracle_rules(dims):
    multilingual = dims.n_vocab >= 51865
    tok = get_tokenizer(multilingual, num_languages=dims.n_vocab - 51765 - int(multilingual), language="en", task="transcribe")
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]))
    init = list(tok.sot_sequence)
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, timestamp_begin=tok.timestamp_begin,
                                 no_timestamps=tok.no_timestamps, suppress_tokens=suppress, blank_token=tok.encode(" ")[0],
                                 no_speech=tok.no_speech)
    return tok, init, rules


def assert_same_or_near_tie(got, want, om, feats, init, rules, what):
    """Token lists of the fp16 engine: equal, or at the first difference the two candidates are within twice the fp16
    logit bound of each other in the oracle's filtered logits (a rounding-level tie; everything after it legitimately
    differs)."""
    t = oracle.first_divergence(got, want)
    if t is None:
        return
    assert t < len(got) and t < len(want), (what, t, got, want)
    lg = oracle.filtered_logits(om, feats, init, want[:t], rules)
    margin = abs(float(lg[got[t]]) - float(lg[want[t]]))
    assert margin < 2 * FP16_LOGIT_BOUND, (what, "diverged at", t, "margin", margin, got[t], want[t])


def test_batch_invariance_fp16(setup, gpu_device):
    """size-independent property at the bench batch size: 8 clips decoded together == each decoded alone.
    fp32 engine: token ids exact.  fp16 engine: the number of cross-attention key splits depends on the row count
    (fp32 partial sums meet in a different order, activations are then rounded to fp16), so a row may leave its
    single-clip decode at a rounding-level near-tie — and only there (checked against the oracle's margin)."""
    key, dims, sd, model, mel = setup
    mels = torch.stack([whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(60 + i), dims.n_mels,
                                                                                device=gpu_device), 3000) for i in range(8)])
    om = oracle.OracleModel(dims, sd)
    tok, init, rules = _oracle_rules(dims)
    filt = oracle.mel_filterbank(dims.n_mels)
    for fp16 in (False, True):
        opts = whisper_amd.DecodingOptions(language="en", fp16=fp16, sample_len=24)
        together = whisper_amd.decode(model, mels, opts)
        for i in range(8):
            alone = whisper_amd.decode(model, mels[i], opts)
            if not fp16:
                assert alone.tokens == together[i].tokens, i
                continue
            if alone.tokens != together[i].tokens:
                with torch.no_grad():
                    feats = om.encoder(oracle.log_mel_spectrogram(audio(60 + i), filt)[None])[0]
                # the oracle's own continuation decides which of the two is "want"; both must be near-ties of it
                t = oracle.first_divergence(alone.tokens, together[i].tokens)
                prefix = alone.tokens[:t]
                lg = oracle.filtered_logits(om, feats, init, prefix, rules)
                margin = abs(float(lg[alone.tokens[t]]) - float(lg[together[i].tokens[t]]))
                assert margin < 2 * FP16_LOGIT_BOUND, (i, t, margin)


def test_fp16_tracks_fp32(setup):
    """fp16 engine vs the fp32 reference tokens: not guaranteed identical (activation rounding), but the first
    tokens and the no-speech probability must agree"""
    key, dims, sd, model, mel = setup
    res = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=20))
    want = G[f"{key}_greedy_ts_tokens"].tolist()
    assert res.tokens[:4] == want[:4]
    assert abs(res.no_speech_prob - G[f"{key}_greedy_ts_stats"][1]) < 0.05 * G[f"{key}_greedy_ts_stats"][1] + 1e-7


def test_detect_language_golden(setup):
    key, dims, sd, model, mel = setup
    if not model.is_multilingual:
        with pytest.raises(ValueError):
            whisper_amd.detect_language(model, mel)
        return
    tok, probs = whisper_amd.detect_language(model, mel.float())
    assert int(tok) == int(G[f"{key}_lang_token"][0])
    top = sorted(probs.items(), key=lambda kv: -kv[1])[:5]
    assert [c for c, _ in top] == G[f"{key}_lang_top5"].tolist()
    assert np.allclose([p for _, p in top], G[f"{key}_lang_top5_p"], rtol=2e-3)


def test_find_alignment_golden(setup):
    """cross-attention QK capture + softmax/z-norm/median + DTW kernels vs the reference's find_alignment"""
    from whisper_amd.timing import find_alignment
    key, dims, sd, model, mel = setup
    multilingual = model.is_multilingual
    tok = get_tokenizer(multilingual, num_languages=model.num_languages, language="en", task="transcribe")
    al = find_alignment(model, tok, G[f"{key}_align_tokens"].tolist(), mel.float(), 3000)
    starts, ends = np.array([w.start for w in al]), np.array([w.end for w in al])
    assert len(al) == len(G[f"{key}_align_start"])
    assert np.abs(starts - G[f"{key}_align_start"]).max() < 1e-6         # fp32 engine: frame indices exact
    assert np.abs(ends - G[f"{key}_align_end"]).max() < 1e-6
    assert np.allclose([w.probability for w in al], G[f"{key}_align_prob"], rtol=5e-3, atol=1e-6)


def test_find_alignment_batch_equals_single(setup, gpu_device):
    """BASELINE configs[4] shape: word alignment of a batch of clips in one pass (one teacher-forced prefill, one launch
    per alignment stage, one DTW workgroup per clip) == find_alignment clip by clip: same words, same frame times
    (exact), same probabilities (1e-4); clips of different token counts and frame counts, one of them empty."""
    from whisper_amd.timing import find_alignment, find_alignment_batch
    key, dims, sd, model, mel = setup
    tok = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language="en", task="transcribe")
    mels = torch.stack([whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(70 + i), dims.n_mels, device=gpu_device), 3000)
                        for i in range(4)])
    texts = [tok.encode(" hello world this is a test of word level timing"), tok.encode(" one two three"), [],
             tok.encode(" the quick brown fox jumps over the lazy dog and keeps running for a while longer")]
    frames = [3000, 2000, 3000, 2600]
    got = find_alignment_batch(model, tok, texts, mels.float(), frames)
    assert got[2] == []
    for i in (0, 1, 3):
        want = find_alignment(model, tok, texts[i], mels[i].float(), frames[i])
        assert [w.word for w in got[i]] == [w.word for w in want] and [w.tokens for w in got[i]] == [w.tokens for w in want]
        assert np.abs(np.array([w.start for w in got[i]]) - np.array([w.start for w in want])).max() < 1e-6, i
        assert np.abs(np.array([w.end for w in got[i]]) - np.array([w.end for w in want])).max() < 1e-6, i
        assert np.allclose([w.probability for w in got[i]], [w.probability for w in want], rtol=1e-4, atol=1e-7)
    # and against the live reference's fixture for the golden clip / token list
    al = find_alignment_batch(model, tok, [G[f"{key}_align_tokens"].tolist()], mel[None].float(), [3000])[0]
    assert np.abs(np.array([w.start for w in al]) - G[f"{key}_align_start"]).max() < 1e-6
    assert np.abs(np.array([w.end for w in al]) - G[f"{key}_align_end"]).max() < 1e-6
    # with the encoder output handed in (what transcribe() does with DecodingResult.audio_features): nothing changes
    feats = model.encoder(mels.float())
    again = find_alignment_batch(model, tok, texts, mels.float(), frames, audio_features=feats)
    assert again == got
    assert find_alignment(model, tok, texts[3], mels[3].float(), frames[3], audio_features=feats[3]) == \
        find_alignment(model, tok, texts[3], mels[3].float(), frames[3])


def test_transcribe_golden(setup):
    """50 s, two windows + fallback-free greedy + word timestamps through model.transcribe(): segment token ids,
    seeks and boundaries equal the reference's; word times exact (fp32 engine)"""
    key, dims, sd, model, mel = setup
    a50 = np.concatenate([audio(21), audio(22, 320000)])
    r = model.transcribe(a50, temperature=0.0, fp16=False, language="en", sample_len=16, word_timestamps=True,
                         condition_on_previous_text=True)
    assert set(r) == {"text", "segments", "language"} and r["language"] == "en"
    assert len(r["segments"]) == int(G[f"{key}_tr_n_segments"][0])
    assert [t for s in r["segments"] for t in s["tokens"]] == G[f"{key}_tr_tokens"].tolist()
    bounds = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
    assert np.array_equal(bounds[:, 0], G[f"{key}_tr_seg_bounds"][:, 0])
    assert np.abs(bounds[:, 1:] - G[f"{key}_tr_seg_bounds"][:, 1:]).max() < 1e-6
    words = np.array([[w["start"], w["end"]] for s in r["segments"] for w in s["words"]])
    assert words.shape == G[f"{key}_tr_word_times"].shape
    assert np.abs(words - G[f"{key}_tr_word_times"]).max() < 1e-6
    for s in r["segments"]:
        assert {"id", "seek", "start", "end", "text", "tokens", "temperature", "avg_logprob", "compression_ratio",
                "no_speech_prob", "words"} <= set(s)


E = np.load(os.path.join(os.path.dirname(__file__), "golden", "edge_cases.npz"))
EDGE = {
    "empty": lambda: np.zeros(0, dtype=np.float32),
    "short": lambda: audio(31, 4960),
    "tail": lambda: audio(32, 16000 * 31),
    "silence": lambda: np.zeros(16000 * 12, dtype=np.float32),
}


@pytest.mark.parametrize("case", list(EDGE))
def test_transcribe_edge_cases(setup, case):
    """model.transcribe() on degenerate inputs against the LIVE reference (tests/golden/make_golden_edge.py): no samples at
    all (no window, no segment), 0.31 s of signal, a 31 s clip whose second window is a 1 s tail, 12 s of zeros.  fp32
    engine: segment count, token ids, seeks and bounds exact; no_speech_prob / avg_logprob to 1e-3."""
    key, dims, sd, model, mel = setup
    r = model.transcribe(EDGE[case](), temperature=0.0, fp16=False, language="en", sample_len=12,
                         condition_on_previous_text=True)
    p = f"{key}_{case}"
    assert set(r) == {"text", "segments", "language"}
    assert len(r["segments"]) == int(E[p + "_n_segments"][0])
    assert [t for s in r["segments"] for t in s["tokens"]] == E[p + "_tokens"].tolist()
    assert len(r["text"]) == int(E[p + "_text_len"][0])
    if r["segments"]:
        bounds = np.array([[s["seek"], s["start"], s["end"]] for s in r["segments"]])
        assert np.array_equal(bounds[:, 0], E[p + "_bounds"][:, 0])
        assert np.abs(bounds[:, 1:] - E[p + "_bounds"][:, 1:]).max() < 1e-6
        stats = np.array([[s["no_speech_prob"], s["avg_logprob"]] for s in r["segments"]])
        assert np.abs(stats - E[p + "_stats"]).max() < 1e-3
    if case == "empty" and model.is_multilingual:       # language detection on a window that is all padding
        r2 = model.transcribe(np.zeros(0, dtype=np.float32), temperature=0.0, fp16=False, sample_len=4)
        assert r2["language"] == str(E[f"{key}_empty_detected_language"][0]) and r2["segments"] == []


@pytest.mark.parametrize("cond,beam", [(False, None), (True, None), (True, 3)])
def test_transcribe_batch_equals_sequential(setup, cond, beam):
    """transcribe_batch (lock-step batching over files) must return exactly what transcribe() returns
    file by file — same tokens, seeks, boundaries, word times — for files of different lengths (1, 2 and 3 windows),
    with and without conditioning on the previous window (with it, prompts diverge: rows of different prompt lengths
    share one device-side call, every row at its own positions), greedy and beam search (beam 3: the ragged rows go
    through wh_task_beam with a lag per segment; the shared-history cache permutation is off there).  fp32 strict
    engine: the batched and single decodes are compared for exact equality."""
    key, dims, sd, model, mel = setup
    files = [audio(31, 200000), np.concatenate([audio(32), audio(33, 240000)]),
             np.concatenate([audio(34), audio(35), audio(36, 100000)]), audio(37, 480000)]
    kw = dict(temperature=0.0, fp16=False, language="en", sample_len=12, word_timestamps=True,
              condition_on_previous_text=cond, no_speech_threshold=None, logprob_threshold=None,
              compression_ratio_threshold=None)
    if beam:
        kw["beam_size"] = beam
    want = [model.transcribe(a, **kw) for a in files]
    got = model.transcribe_batch(files, **kw)
    assert len(got) == len(want)
    n_windows = 0
    for g, w in zip(got, want):
        assert g["language"] == w["language"] and g["text"] == w["text"]
        assert [s["tokens"] for s in g["segments"]] == [s["tokens"] for s in w["segments"]]
        assert [s["seek"] for s in g["segments"]] == [s["seek"] for s in w["segments"]]
        assert np.allclose([[s["start"], s["end"]] for s in g["segments"]], [[s["start"], s["end"]] for s in w["segments"]])
        gw = [[x["start"], x["end"]] for s in g["segments"] for x in s["words"]]
        ww = [[x["start"], x["end"]] for s in w["segments"] for x in s["words"]]
        assert np.allclose(gw, ww, atol=0.0201)
        assert np.allclose([s["avg_logprob"] for s in g["segments"]], [s["avg_logprob"] for s in w["segments"]], atol=1e-4)
        n_windows += len({s["seek"] for s in w["segments"]})
    assert n_windows >= 6


def test_transcribe_batch_with_degenerate_files(setup):
    """transcribe_batch over a list that mixes an EMPTY clip, a 0.31 s clip, silence and a two-window file (default
    thresholds on: the no-speech / log-prob branches and the temperature ladder are live): every file's result equals its
    own transcribe() — the empty file contributes no window and leaves the lock-step schedule of the others untouched."""
    key, dims, sd, model, mel = setup
    files = [EDGE["empty"](), EDGE["short"](), EDGE["silence"](), EDGE["tail"](), EDGE["empty"]()]
    kw = dict(temperature=0.0, fp16=False, language="en", sample_len=12, condition_on_previous_text=True)
    want = [model.transcribe(a, **kw) for a in files]
    got = model.transcribe_batch(files, **kw)
    assert len(got) == len(want) == 5
    for g, w in zip(got, want):
        assert g["language"] == w["language"] and g["text"] == w["text"]
        assert [s["tokens"] for s in g["segments"]] == [s["tokens"] for s in w["segments"]]
        assert [(s["seek"], s["start"], s["end"]) for s in g["segments"]] == [(s["seek"], s["start"], s["end"]) for s in w["segments"]]
    assert got[0]["segments"] == [] and got[4]["segments"] == [] and got[0]["text"] == ""
    assert model.transcribe_batch([], **kw) == []


def test_temperature_fallback_runs_sampling_path(setup):
    """transcribe.py:184-224: a window whose greedy result trips the log-prob threshold is re-decoded at the next
    temperature (sampling, drawn on the device); best_of > 1 exercises the grouped rows."""
    key, dims, sd, model, mel = setup
    torch.manual_seed(0)
    r = model.transcribe(audio(41, 200000), temperature=(0.0, 0.5), logprob_threshold=0.0, fp16=False, language="en",
                         sample_len=10, best_of=3, condition_on_previous_text=False)
    assert len(r["segments"]) >= 1
    assert all(s["temperature"] == 0.5 for s in r["segments"])          # the fallback result was kept
    assert all(np.isfinite(s["avg_logprob"]) for s in r["segments"])
    # the same ladder climbed by several files together: retries are decoded as batches (best_of rows per window,
    # per-row prompts of different lengths once the files have produced text)
    files = [audio(42, 200000), np.concatenate([audio(43), audio(44, 150000)]), audio(45, 480000)]
    got = model.transcribe_batch(files, temperature=(0.0, 0.5), logprob_threshold=0.0, fp16=False, language="en",
                                 sample_len=10, best_of=3, condition_on_previous_text=True, no_speech_threshold=None)
    assert len(got) == 3 and all(len(g["segments"]) >= 1 for g in got)
    assert all(s["temperature"] == 0.5 and np.isfinite(s["avg_logprob"]) for g in got for s in g["segments"])
    opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=8, temperature=0.7, best_of=2)
    res = whisper_amd.decode(model, mel, opts)
    assert len(res.tokens) <= 8 and res.temperature == 0.7
    with pytest.raises(ValueError):
        whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", beam_size=2, best_of=2))   # decoding.py:572-585


def _prompted_mels(dims, gpu_device, n):
    return torch.stack([whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(50 + i), dims.n_mels,
                                                                                device=gpu_device), 3000) for i in range(n)])


@pytest.mark.parametrize("fp16", [False, True])
def test_ragged_prompts_equal_single_row_decodes(setup, gpu_device, fp16):
    """DecodingTask(prompts=...): rows of ONE fused greedy call conditioned on previous-text
    prompts of different lengths (none, 1, 5, 17, 60 and 150 tokens; every row at its own cache positions on the
    device) must return what each segment returns decoded alone with options.prompt — the reference's only way to
    run it.  fp32: token ids exact and statistics to 1e-4; fp16: token ids equal, or the first difference is a
    rounding-level near-tie in the oracle's filtered logits (the batched call prefills through the GEMM path, the
    single rows through the few-row projections), statistics to 2e-2.  One row is also checked against the CPU oracle."""
    key, dims, sd, model, mel = setup
    rng = np.random.default_rng(7)
    prompts = [None, [1234], rng.integers(300, 40000, 5).tolist(), rng.integers(300, 40000, 17).tolist(),
               rng.integers(300, 40000, 60).tolist(), rng.integers(300, 40000, 150).tolist()]
    mels = _prompted_mels(dims, gpu_device, len(prompts))
    opts = whisper_amd.DecodingOptions(language="en", fp16=fp16, sample_len=14)
    got = whisper_amd.decode(model, mels, opts, prompts=prompts)
    tol = 2e-2 if fp16 else 1e-4
    for i, p in enumerate(prompts):
        want = whisper_amd.decode(model, mels[i], opts, prompt=p)
        if fp16 and got[i].tokens != want.tokens:
            om_ = oracle.OracleModel(dims, sd)
            tok_, init_, rules_ = _oracle_rules(dims)
            init_i = ([tok_.sot_prev] + list(p) if p else []) + init_
            rules_i = oracle.SamplingRules(**{**rules_.__dict__, "sample_begin": len(init_i), "sot_index": init_i.index(tok_.sot)})
            with torch.no_grad():
                feats_i = om_.encoder(oracle.log_mel_spectrogram(audio(50 + i), oracle.mel_filterbank(dims.n_mels))[None])[0]
            assert_same_or_near_tie(got[i].tokens, want.tokens, om_, feats_i, init_i, rules_i, i)
            continue
        assert got[i].tokens == want.tokens, i
        assert abs(got[i].avg_logprob - want.avg_logprob) < tol
        assert abs(got[i].no_speech_prob - want.no_speech_prob) < max(1e-6, tol * want.no_speech_prob)
        assert got[i].text == want.text and got[i].compression_ratio == want.compression_ratio
    if not fp16:
        # every row against the LIVE reference decoding that segment alone with options.prompt
        # (tests/golden/make_golden_prompts.py): ids exact, statistics to 1e-3
        P = np.load(os.path.join(os.path.dirname(__file__), "golden", "prompts_micro.npz"))
        for i in range(len(prompts)):
            assert got[i].tokens == [t for t in P[f"{key}_tokens"][i].tolist() if t >= 0], i
            assert abs(got[i].avg_logprob - P[f"{key}_stats"][i, 0]) < 1e-3
            assert abs(got[i].no_speech_prob - P[f"{key}_stats"][i, 1]) < max(1e-7, 2e-3 * P[f"{key}_stats"][i, 1])
        om = oracle.OracleModel(dims, sd)
        multilingual = dims.n_vocab >= 51865
        tok = get_tokenizer(multilingual, num_languages=dims.n_vocab - 51765 - int(multilingual), language="en", task="transcribe")
        suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]))
        i = 3
        init = [tok.sot_prev] + prompts[i] + list(tok.sot_sequence)
        rules = oracle.SamplingRules(sample_begin=len(init), sot_index=init.index(tok.sot), eot=tok.eot,
                                     timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                                     suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)
        omel = oracle.log_mel_spectrogram(audio(50 + i), oracle.mel_filterbank(dims.n_mels))
        with torch.no_grad():
            ref = oracle.greedy_decode(om, om.encoder(omel[None]), init, 14, rules)
        body = ref["tokens"][0, len(init):].tolist()
        body = body[: body.index(tok.eot)] if tok.eot in body else body
        assert got[i].tokens == body


def test_row_prompts_other_modes(setup, gpu_device):
    """per-row prompts outside the fused greedy loop: equal-length prompts under beam search and under the generic
    host loop equal the single-row decodes; language detection writes every row's own language slot; prompts of
    different lengths also run under the device-side beam search (equal to the single-row decodes) and are refused
    where rows cannot sit at different positions (host loop, long prompts)."""
    key, dims, sd, model, mel = setup
    rng = np.random.default_rng(8)
    mels = _prompted_mels(dims, gpu_device, 3)
    same_len = [rng.integers(300, 40000, 9).tolist() for _ in range(3)]
    opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=8, beam_size=2)
    got = whisper_amd.decode(model, mels, opts, prompts=same_len)
    for i in range(3):
        assert got[i].tokens == whisper_amd.decode(model, mels[i], opts, prompt=same_len[i]).tokens, i
    ragged = [same_len[0], same_len[1][:4], None]
    # prompts of different lengths under beam search: the device-side loop carries a lag per segment
    for bopts in (opts, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=10, beam_size=5, patience=1.5)):
        got = whisper_amd.decode(model, mels, bopts, prompts=ragged)
        for i in range(3):
            want = whisper_amd.decode(model, mels[i], bopts, prompt=ragged[i])
            assert got[i].tokens == want.tokens, i
            assert abs(got[i].avg_logprob - want.avg_logprob) < 1e-4 and got[i].text == want.text
    with pytest.raises(ValueError):      # a user filter forces the host loop, which cannot place rows at different positions
        task = whisper_amd.decoding.DecodingTask(model, opts, prompts=ragged)
        task.logit_filters.append(whisper_amd.decoding.LogitFilter())
        task.run(mels)
    with pytest.raises(ValueError):      # 230 + 3 initial tokens + 224 steps do not fit n_text_ctx for the short rows' shared counter
        whisper_amd.decode(model, mels, whisper_amd.DecodingOptions(language="en", fp16=False),
                           prompts=[list(range(1000, 1230)), [5], None])
    with pytest.raises(ValueError):
        whisper_amd.decode(model, mels, whisper_amd.DecodingOptions(language="en", fp16=False, prompt=[7]), prompts=ragged)
    if dims.n_vocab >= 51865:            # multilingual: detected language goes to sot_index + 1 of every row
        opts = whisper_amd.DecodingOptions(fp16=False, sample_len=6)
        got = whisper_amd.decode(model, mels, opts, prompts=ragged)
        for i in range(3):
            want = whisper_amd.decode(model, mels[i], opts, prompt=ragged[i])
            assert got[i].language == want.language and got[i].tokens == want.tokens, i


@pytest.mark.parametrize("kw", [dict(beam_size=4), dict(beam_size=2, patience=2.0), dict(beam_size=4, patience=0.5),
                                dict(beam_size=5, without_timestamps=True), dict(beam_size=8)])
def test_device_beam_search_equals_host_loop(setup, gpu_device, kw):
    """wh_task_beam (filters + log_softmax + top-(beam+1) + BeamSearchDecoder.update + cache permutation on the device,
    SURVEY.md §8f rank 2) against the host-driven loop (per-step wh_task_step, torch filters, the Python
    BeamSearchDecoder checked against the reference in tests/test_host_logic.py), forced by a no-op user filter:
    same tokens, same finished lists in the same order, same scores — for 3 segments decoded together, with patience
    above and below 1 (max_candidates != beam_size) and runs that end by completion as well as by the step budget."""
    from whisper_amd.decoding import DecodingTask, LogitFilter

    class Noop(LogitFilter):
        def apply(self, logits, tokens):
            return None

    key, dims, sd, model, mel = setup
    mels = _prompted_mels(dims, gpu_device, 3)
    for sample_len in (5, 24):
        opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=sample_len, **kw)
        fused_task = DecodingTask(model, opts)
        assert fused_task._fused_beam_ok()
        fused = fused_task.run(mels)
        host_task = DecodingTask(model, opts)
        host_task.logit_filters.append(Noop())
        assert not host_task._fused_beam_ok()
        host = host_task.run(mels)
        for a, (f, h) in enumerate(zip(fused, host)):
            assert f.tokens == h.tokens, (sample_len, a)
            assert abs(f.avg_logprob - h.avg_logprob) < 1e-5
            assert abs(f.no_speech_prob - h.no_speech_prob) < 1e-6
        ff, hf = fused_task.decoder.finished_sequences, host_task.decoder.finished_sequences
        assert [list(d.keys()) for d in ff] == [list(d.keys()) for d in hf]
        assert np.allclose([v for d in ff for v in d.values()], [v for d in hf for v in d.values()], atol=1e-4)


def test_beam_search_survives_a_handoff_timeout(setup, gpu_device):
    """2 clips x beam 4 = 8 rows in the fp16 engine, the task created with WH_TASK_FUSED_SELF (opt-in since round 6; a beam task
    cannot take the fused cross attention, which wants one row per audio): the step runs the fused self-attention launch.  With every hand-off
    forced to time out (WH_TASK_EXPIRE_HANDOFFS, fresh task) wh_task_beam re-runs the search on the two-launch kernels
    inside the call; decode() returns what it returns without the time-outs (the fused self attention is bit-identical
    to the two-launch form, so ids and scores agree exactly)."""
    key, dims, sd, model, mel = setup
    mels = _prompted_mels(dims, gpu_device, 2)
    opts = whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=12, beam_size=4)
    eng = model.engine(torch.float16)
    eng.drop_cached_tasks()
    want = whisper_amd.decode(model, mels, opts)
    eng.drop_cached_tasks()                                  # the next task is created with the fault-injection flag
    eng.debug_task_flags = hip.WH_TASK_EXPIRE_HANDOFFS | hip.WH_TASK_FUSED_SELF
    try:
        got = whisper_amd.decode(model, mels, opts)
    finally:
        eng.debug_task_flags = 0
    cached = [t for t in eng._task_cache if t.n_rows == 8]
    assert cached and cached[-1].handoff_fallbacks == 1 and not cached[-1].fused_self_attention
    eng.drop_cached_tasks()
    for g, w in zip(got, want):
        assert g.tokens == w.tokens and g.avg_logprob == w.avg_logprob and g.no_speech_prob == w.no_speech_prob


def test_host_driven_loop_survives_a_handoff_timeout(setup, gpu_device):
    """The loop that steps the decoder from the host (a user logit filter keeps it off the device-side loops) on the fp16
    engine's fused step kernels: with every hand-off forced to time out, DecodingTask notices after the window (the counter
    of wh_task_info(t, 1) moved) and decodes it again on a task that uses the two-launch kernels — the result equals a run
    that was on those kernels from the start."""
    from whisper_amd.decoding import DecodingTask, LogitFilter

    class Noop(LogitFilter):
        def apply(self, logits, tokens):
            return None

    key, dims, sd, model, mel = setup
    mels = _prompted_mels(dims, gpu_device, 2)
    opts = whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=10)
    eng = model.engine(torch.float16)

    def run(two_launch):
        eng.drop_cached_tasks()
        task = DecodingTask(model, opts)
        task.logit_filters.append(Noop())
        assert not task._fused_greedy_ok(torch.zeros(2, 3, dtype=torch.int64))
        task.inference.two_launch = two_launch
        return task, task.run(mels)

    _, want = run(True)
    eng.debug_task_flags = hip.WH_TASK_EXPIRE_HANDOFFS
    try:
        task, got = run(False)
    finally:
        eng.debug_task_flags = 0
    eng.drop_cached_tasks()
    assert task.inference.two_launch is True                 # the window was decoded a second time
    for g, w in zip(got, want):
        assert g.tokens == w.tokens and g.avg_logprob == w.avg_logprob and g.no_speech_prob == w.no_speech_prob


def test_device_sampling(setup, gpu_device):
    """Temperature sampling inside the fused loop (GreedyDecoder.update at T > 0, decoding.py:281-293; SURVEY.md §8f
    rank 2).  The reference draws from torch's generator, so parity is distributional:
    (a) T -> 0 reproduces the arg-max decode exactly, with the same log-probabilities;
    (b) same torch seed -> same tokens, other seed -> other tokens;
    (c) the accumulated log-probability of a drawn token is log_softmax(filtered logits)[token], UNSCALED by T;
    (d) over 384 draws of the first token, the mean of log p_T(token) sits within 5 standard errors of its
        expectation under p_T = softmax(filtered logits / T) computed on the host with the torch filters."""
    import torch.nn.functional as F
    from whisper_amd.decoding import DecodingTask
    key, dims, sd, model, mel = setup
    greedy = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=16))
    cold = whisper_amd.decode(model, mel, whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=16, temperature=1e-6))
    assert cold.tokens == greedy.tokens and abs(cold.avg_logprob - greedy.avg_logprob) < 1e-4

    opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=12, temperature=0.9, best_of=4)
    torch.manual_seed(11); a = whisper_amd.decode(model, mel, opts)
    torch.manual_seed(11); b = whisper_amd.decode(model, mel, opts)
    torch.manual_seed(12); c = whisper_amd.decode(model, mel, opts)
    assert a.tokens == b.tokens and a.avg_logprob == b.avg_logprob
    assert a.tokens != c.tokens
    assert np.isfinite(a.avg_logprob) and a.temperature == 0.9

    for T in (1.0, 0.5):
        opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=1, temperature=T, best_of=8)
        probe = DecodingTask(model, opts)
        assert probe._fused_greedy_ok(None)
        feats = probe._get_audio_features(mel[None])
        rows = torch.tensor([probe.initial_tokens]).repeat(8, 1).to(gpu_device)
        logits = probe.inference.logits(rows, feats)[:, -1].clone()
        probe.inference.cleanup_caching()
        for f in probe.logit_filters:
            f.apply(logits, rows)
        lp1 = F.log_softmax(logits[0].float(), -1).cpu()            # unscaled: what sum_logprobs accumulates
        lpT = F.log_softmax(logits[0].float() / T, -1).cpu()        # the distribution that is sampled
        pT = lpT.exp()
        finite = torch.isfinite(lpT)
        mean = float((pT[finite] * lpT[finite]).sum())
        var = float((pT[finite] * (lpT[finite] - mean) ** 2).sum())
        draws = []
        torch.manual_seed(100 + int(T * 10))
        for _ in range(48):
            task = DecodingTask(model, opts)
            toks, sums, _ = task._main_loop(feats, rows.clone())
            assert toks.shape == (8, rows.shape[1] + 1)
            for tkn, s in zip(toks[:, -1].tolist(), sums.tolist()):
                assert torch.isfinite(lpT[tkn]), tkn                  # never a filtered token
                assert abs(s - float(lp1[tkn])) < 2e-4                # (c)
                draws.append(float(lpT[tkn]))
        n = len(draws)
        assert n == 384 and len(set(draws)) > 8
        assert abs(np.mean(draws) - mean) < 5.0 * np.sqrt(var / n) + 1e-3, (T, np.mean(draws), mean, var)


def test_incremental_decoder_with_kv_cache_hooks(setup, gpu_device):
    """model.install_kv_cache_hooks() + model.decoder(tokens, xa, kv_cache=cache) (reference model.py:227-249, 310-341):
    feeding the tokens incrementally — 3 at first, then one at a time — returns, for every token fed, the logits of
    one teacher-forced pass over the whole sequence (fp32 engine; GEMM prefill vs GEMV step kernels: 5e-3 on logits
    of order 1), and remove() releases the task"""
    from whisper_amd import model as mm
    key, dims, sd, model, mel = setup
    feats = model.encoder(mel[None].float())
    tok = get_tokenizer(dims.n_vocab >= 51865, num_languages=dims.n_vocab - 51765 - int(dims.n_vocab >= 51865),
                        language="en", task="transcribe")
    rng = np.random.default_rng(12)
    rows = [list(tok.sot_sequence) + rng.integers(300, 40000, 9).tolist() for _ in range(2)]
    toks = torch.tensor(rows, device=gpu_device)
    T = toks.shape[1]
    full = model.decoder(toks, feats)                                  # (2, T, V), one pass
    cache, hooks = model.install_kv_cache_hooks()
    parts = [model.decoder(toks[:, :3], feats, kv_cache=cache)]
    for at in range(3, T):
        parts.append(model.decoder(toks[:, at: at + 1], feats, kv_cache=cache))
    inc = torch.cat(parts, dim=1)
    assert inc.shape == full.shape
    assert (inc - full).abs().max().item() < 5e-3
    task = cache[mm._TASK_KEY]
    assert task.position == T
    for h in hooks:
        h.remove()
    # the caches were released: destroyed, or parked in the engine's task cache for the next decode of this shape
    assert mm._TASK_KEY not in cache and (task.handle is None or task in model.engine(feats.dtype)._task_cache)


@pytest.mark.parametrize("fp16,beam", [(False, None), (True, None), (False, 3)])
def test_decode_many_in_lanes_equals_sequential(setup, fp16, beam):
    """whisper_amd.decode_many: several batches decoded with up to 3 chains in flight — each on a HIP stream of its own
    (HipModel.lane), all driven from the calling thread (run_interleaved: wh_task_*_begin + wh_task_poll in turn), the encoder on
    the engine's one stream.  With chain_rows=None every batch is its own chain and must give what decode() gives batch by
    batch: token ids, avg_logprob, no_speech_prob (exactly in the fp32 engine); greedy and beam search, both engines; raw audio
    batches take their log-mel per batch.  Batches of different sizes (different task shapes), more batches than lanes (a lane runs several), twice in a
    row (the lanes' tasks come back from the engine's cache).  With the default chain_rows=24 consecutive batches are coalesced
    into wider chains: exact in the fp32 engine, equal ids and log-probabilities to 5e-3 in the fp16 engine (the order of fp32
    partial sums follows the row count)."""
    key, dims, sd, model, mel = setup
    dev = mel.device
    clips = [audio(50 + i) for i in range(9)]
    mels = [whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(c, dims.n_mels, device=dev), 3000) for c in clips]
    batches = [torch.stack(mels[0:3]), torch.stack(mels[3:4]), torch.stack(mels[4:6]), torch.stack(mels[6:9]), torch.stack(mels[0:2])]
    opts = whisper_amd.DecodingOptions(language="en", fp16=fp16, sample_len=16, beam_size=beam)
    want = [whisper_amd.decode(model, (b.half() if fp16 else b), opts) for b in batches]
    # a lane's task runs its cross attention as two launches (no spinning kernel beside other chains; decode() alone uses the fused
    # launch, whose fp32 partial sums meet in another order): exact in the fp32 engine, ids equal and values to 5e-3 in the fp16 engine
    tol = 5e-3 if fp16 else 1e-6
    for _ in range(2):
        got = whisper_amd.decode_many(model, [(b.half() if fp16 else b) for b in batches], opts, in_flight=3, chain_rows=None)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert [r.tokens for r in g] == [r.tokens for r in w]
            assert np.allclose([r.avg_logprob for r in g], [r.avg_logprob for r in w], atol=tol)
            assert np.allclose([r.no_speech_prob for r in g], [r.no_speech_prob for r in w], atol=tol)
    # coalesced (default chain_rows=24): greedy 3 + 1 + 2 + 3 + 2 = 11 rows in one chain; beam 3: 9 + 3 + 6 | 9 + 6 rows
    got = whisper_amd.decode_many(model, [(b.half() if fp16 else b) for b in batches], opts, in_flight=2)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert [r.tokens for r in g] == [r.tokens for r in w]
        tol = 5e-3 if fp16 else 1e-6
        assert np.allclose([r.avg_logprob for r in g], [r.avg_logprob for r in w], atol=tol)
        assert np.allclose([r.no_speech_prob for r in g], [r.no_speech_prob for r in w], atol=tol)
    # raw audio in, log-mel per batch
    raw = [torch.from_numpy(np.stack(clips[0:3])).to(dev), torch.from_numpy(np.stack(clips[4:6])).to(dev)]
    got = whisper_amd.decode_many(model, raw, opts, in_flight=2, chain_rows=None)
    want_raw = [whisper_amd.decode(model, whisper_amd.log_mel_spectrogram(r, dims.n_mels).to(torch.float16 if fp16 else torch.float32), opts)
                for r in raw]
    assert [[r.tokens for r in g] for g in got] == [[r.tokens for r in w] for w in want_raw]
    # an exception inside a chain reaches the caller, and the engine keeps working afterwards
    with pytest.raises(Exception):
        whisper_amd.decode_many(model, [batches[0], torch.zeros(2, dims.n_mels, 17, device=dev)], opts, in_flight=2, chain_rows=None)
    assert [r.tokens for r in whisper_amd.decode(model, (batches[1].half() if fp16 else batches[1]), opts)] == [r.tokens for r in want[1]]


def test_run_in_lanes_threads_and_seeds(setup):
    """whisper_amd.run_in_lanes (one host thread + HIP stream per lane, for arbitrary callables): results in job order; an
    exception raised by a job reaches the caller; and sampling seeds are a function of torch's generator state and the JOB's
    index, not of which lane runs a job when — two runs under the same torch.manual_seed draw the same tokens at temperature 0.7
    (lanes used to race for draws from the process-wide generator)."""
    key, dims, sd, model, mel = setup
    dev = mel.device
    mels = [whisper_amd.pad_or_trim(whisper_amd.log_mel_spectrogram(audio(70 + i), dims.n_mels, device=dev), 3000)[None] for i in range(5)]
    opts = whisper_amd.DecodingOptions(language="en", fp16=False, sample_len=12, temperature=0.7, best_of=2)

    def run():
        torch.manual_seed(1234)
        return whisper_amd.run_in_lanes(model, [lambda m=m: whisper_amd.decode(model, m, opts) for m in mels], 3, torch.float32)
    a, b = run(), run()
    assert [[r.tokens for r in x] for x in a] == [[r.tokens for r in x] for x in b]
    torch.manual_seed(99)
    c = whisper_amd.run_in_lanes(model, [lambda m=m: whisper_amd.decode(model, m, opts) for m in mels], 3, torch.float32)
    assert [[r.tokens for r in x] for x in c] != [[r.tokens for r in x] for x in a]        # another seed, other draws

    def boom():
        raise ValueError("job failed")
    with pytest.raises(ValueError):
        whisper_amd.run_in_lanes(model, [lambda: 1, boom, lambda: 3], 2, torch.float32)
    assert whisper_amd.run_in_lanes(model, [lambda i=i: i * i for i in range(7)], 3, torch.float32) == [i * i for i in range(7)]


def test_transcribe_batch_in_flight_equals_one_lane(setup):
    """transcribe_batch(in_flight=2): two groups of files driven concurrently on their own threads and streams — every file's
    result (tokens, seeks, boundaries, word times) exactly what in_flight=1 returns, in input order; fp32 strict engine."""
    key, dims, sd, model, mel = setup
    files = [audio(31, 200000), np.concatenate([audio(32), audio(33, 240000)]), audio(37, 480000),
             np.concatenate([audio(34), audio(35), audio(36, 100000)]), audio(38, 90000)]
    kw = dict(temperature=0.0, fp16=False, language="en", sample_len=12, word_timestamps=True, batch_size=2,
              condition_on_previous_text=True, no_speech_threshold=None, logprob_threshold=None, compression_ratio_threshold=None)
    want = model.transcribe_batch(files, **kw)
    got = model.transcribe_batch(files, in_flight=2, **kw)
    assert len(got) == len(want) == 5
    for g, w in zip(got, want):
        assert g["language"] == w["language"] and g["text"] == w["text"]
        assert [s["tokens"] for s in g["segments"]] == [s["tokens"] for s in w["segments"]]
        assert [s["seek"] for s in g["segments"]] == [s["seek"] for s in w["segments"]]
        gw = [[x["start"], x["end"]] for s in g["segments"] for x in s["words"]]
        ww = [[x["start"], x["end"]] for s in w["segments"] for x in s["words"]]
        assert gw == ww
