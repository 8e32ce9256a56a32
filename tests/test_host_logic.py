"""Host-side mirror (whisper_amd.decoding / tokenizer / audio / transcribe) against the LIVE reference, on CPU.
The device kernels are not involved: the logit filters and token decoders are plain torch code that also runs
on CPU tensors, `transcribe` is driven by a scripted fake model.  Needs /root/reference (marker `reference`)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import REFERENCE, SHIMS

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    for p in (SHIMS, REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import whisper
    import whisper.decoding
    import whisper.timing
    import whisper.tokenizer
    return whisper


def test_tokenizer_equivalence(ref):
    from whisper_amd.tokenizer import get_tokenizer
    for ml, nl in ((False, 99), (True, 99), (True, 100)):
        a = get_tokenizer(ml, num_languages=nl, language="en" if ml else None, task="transcribe" if ml else None)
        b = ref.tokenizer.get_tokenizer(ml, num_languages=nl, language="en" if ml else None,
                                        task="transcribe" if ml else None)
        assert a.special_tokens == b.special_tokens
        assert a.sot_sequence == b.sot_sequence and a.non_speech_tokens == b.non_speech_tokens
        assert sorted(a.all_language_tokens) == sorted(b.all_language_tokens)
        for text in ["Hello world, this is Whisper!", " ♪♪ [MUSIC] (laughs)", "다람쥐 헌 쳇바퀴에 타고파", "naïve café — 12,345.67"]:
            assert a.encode(text) == b.encode(text)
            assert a.decode(a.encode(text)) == text
        toks = a.encode(" hello world this is a test") + [a.eot]
        assert a.split_to_word_tokens(toks) == b.split_to_word_tokens(toks)


def test_known_bpe_ids():
    """known GPT-2 ids and the reference's golden split (tests/test_tokenizer.py:27-34): pins the BPE itself"""
    from whisper_amd.tokenizer import get_tokenizer
    g = get_tokenizer(False)
    assert g.encode("hello world") == [31373, 995] and g.encode("Hello world") == [15496, 995]
    m = get_tokenizer(True)
    words, toks = m.split_tokens_on_unicode([8404, 871, 287, 6, 246, 526, 3210, 20378])
    assert words == [" elle", " est", " l", "'", "�", "é", "rit", "oire"]
    assert toks == [[8404], [871], [287], [6], [246], [526], [3210], [20378]]


def test_pad_or_trim_and_filters(ref):
    from whisper_amd import audio
    x = np.random.default_rng(0).standard_normal((3, 1000)).astype(np.float32)
    for n in (500, 1000, 1500):
        assert np.array_equal(audio.pad_or_trim(x, n), ref.pad_or_trim(x, n))
        assert torch.equal(audio.pad_or_trim(torch.from_numpy(x), n), ref.pad_or_trim(torch.from_numpy(x), n))
        assert torch.equal(audio.pad_or_trim(torch.from_numpy(x), n, axis=0), ref.pad_or_trim(torch.from_numpy(x), n, axis=0))
    for n in (80, 128):
        assert (audio.mel_filters("cpu", n) - ref.audio.mel_filters("cpu", n)).abs().max() < 1e-8


def _tok(ref_mod, multilingual=True):
    from whisper_amd.tokenizer import get_tokenizer
    return (get_tokenizer(multilingual, num_languages=100 if multilingual else 99, language="en" if multilingual else None,
                          task="transcribe" if multilingual else None),
            ref_mod.tokenizer.get_tokenizer(multilingual, num_languages=100 if multilingual else 99,
                                            language="en" if multilingual else None,
                                            task="transcribe" if multilingual else None))


def test_logit_filters_match_reference(ref):
    """vectorised filters == the reference's row loops (decoding.py:423-505), bit for bit, over crafted histories"""
    from whisper_amd import decoding as mine
    tk, rtk = _tok(ref)
    TB, V = tk.timestamp_begin, 51866
    g = torch.Generator().manual_seed(0)
    sb = 3
    histories = [
        [], [TB + 5], [TB + 5, 400], [TB + 5, 400, TB + 30], [TB + 5, 400, TB + 30, TB + 30],
        [TB + 5, 400, TB + 30, TB + 30, 900], [400, 401], [400, TB + 7], [TB + 1, TB + 1],
    ]
    for hist in histories:
        R = 4
        tokens = torch.tensor([[50258, 50259, 50360] + hist] * R)
        # make rows differ in their last token where possible
        if len(hist) >= 1:
            tokens[1, -1] = 777
            tokens[2, -1] = TB + 100
        logits = torch.randn(R, V, generator=g) * 3
        logits[3, TB:] += 8.0          # one row where timestamp mass dominates
        a, b = logits.clone(), logits.clone()
        for f in (mine.SuppressBlank(tk, sb), mine.SuppressTokens([1, 5, 9, tk.no_speech]),
                  mine.ApplyTimestampRules(tk, sb, 50)):
            f.apply(a, tokens)
        for f in (ref.decoding.SuppressBlank(rtk, sb), ref.decoding.SuppressTokens([1, 5, 9, rtk.no_speech]),
                  ref.decoding.ApplyTimestampRules(rtk, sb, 50)):
            f.apply(b, tokens)
        assert torch.equal(a, b), hist


class _NullInference:
    def __init__(self):
        self.calls = []

    def rearrange_kv_cache(self, idx):
        self.calls.append(list(idx))


def test_token_decoders_match_reference(ref):
    from whisper_amd import decoding as mine
    g = torch.Generator().manual_seed(1)
    eot, V = 50257, 51866
    # greedy
    tokens = torch.randint(0, 50000, (5, 6), generator=g)
    tokens[2, -1] = eot
    logits = torch.randn(5, V, generator=g)
    sa, sb_ = torch.zeros(5), torch.zeros(5)
    ta, ca = mine.GreedyDecoder(0.0, eot).update(tokens.clone(), logits.clone(), sa)
    tb, cb = ref.decoding.GreedyDecoder(0.0, eot).update(tokens.clone(), logits.clone(), sb_)
    assert torch.equal(ta, tb) and bool(ca) == bool(cb) and torch.equal(sa, sb_)
    # beam search: several steps with EOT made attractive at times, n_audio = 2
    G_, n_audio = 3, 2
    ia, ib = _NullInference(), _NullInference()
    da = mine.BeamSearchDecoder(G_, eot, ia, patience=2.0)
    db = ref.decoding.BeamSearchDecoder(G_, eot, ib, patience=2.0)
    da.reset(); db.reset()
    ta = tb = torch.randint(0, 50000, (n_audio * G_, 4), generator=g)
    sa, sb_ = torch.zeros(n_audio * G_), torch.zeros(n_audio * G_)
    for step in range(6):
        logits = torch.randn(n_audio * G_, V, generator=g) * 2
        if step in (2, 4):
            logits[:, eot] += 9.0
        if step == 3:
            logits[1] = logits[0]          # duplicate candidates across beams
            ta[1] = ta[0]; tb[1] = tb[0]
        ta, ca = da.update(ta, logits.clone(), sa)
        tb, cb = db.update(tb, logits.clone(), sb_)
        assert torch.equal(ta, tb) and ca == cb and torch.equal(sa, sb_) and ia.calls == ib.calls
    fa = da.finalize(ta.reshape(n_audio, G_, -1), sa.reshape(n_audio, G_))
    fb = db.finalize(tb.reshape(n_audio, G_, -1), sb_.reshape(n_audio, G_))
    assert [[t.tolist() for t in s] for s in fa[0]] == [[t.tolist() for t in s] for s in fb[0]]
    assert fa[1] == fb[1]
    ra = mine.MaximumLikelihoodRanker(None).rank(fa[0], fa[1])
    rb = ref.decoding.MaximumLikelihoodRanker(None).rank(fb[0], fb[1])
    assert [int(x) for x in ra] == [int(x) for x in rb]
    ra = mine.MaximumLikelihoodRanker(0.6).rank(fa[0], fa[1])
    rb = ref.decoding.MaximumLikelihoodRanker(0.6).rank(fb[0], fb[1])
    assert [int(x) for x in ra] == [int(x) for x in rb]


def _fake_model(multilingual=True):
    dims = SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=2,
                           n_vocab=51865 if multilingual else 51864, n_text_ctx=448, n_text_state=384,
                           n_text_head=6, n_text_layer=2)
    return SimpleNamespace(dims=dims, is_multilingual=multilingual, device=torch.device("cpu"),
                           num_languages=dims.n_vocab - 51765 - int(multilingual),
                           decoder=SimpleNamespace(blocks=[]))      # the reference's PyTorchInference walks .blocks


def test_task_setup_matches_reference(ref):
    """initial tokens, sot_index, suppress list, option validation (decoding.py:514-642)"""
    from whisper_amd import decoding as mine
    for ml in (True, False):
        model = _fake_model(ml)
        for kw in (dict(), dict(without_timestamps=True), dict(prompt="some previous text", prefix=" and a prefix"),
                   dict(prompt=list(range(1000, 1300)), sample_len=30), dict(suppress_tokens="-1,7,11"),
                   dict(suppress_tokens=[3, 4]), dict(suppress_tokens=""), dict(max_initial_timestamp=None),
                   dict(suppress_blank=False, beam_size=4, patience=1.5), dict(temperature=0.7, best_of=3)):
            a = mine.DecodingTask(model, mine.DecodingOptions(language="en", **kw))
            b = ref.decoding.DecodingTask(model, ref.DecodingOptions(language="en", **kw))
            assert a.initial_tokens == b.initial_tokens and a.sot_index == b.sot_index
            assert a.sample_begin == b.sample_begin and a.sample_len == b.sample_len and a.n_group == b.n_group
            assert [type(f).__name__ for f in a.logit_filters] == [type(f).__name__ for f in b.logit_filters]
            if kw.get("suppress_tokens", "-1"):
                assert a._get_suppress_tokens() == b._get_suppress_tokens()
        for bad in (dict(beam_size=2, best_of=2), dict(best_of=2), dict(patience=1.0), dict(length_penalty=1.5)):
            with pytest.raises(ValueError):
                mine.DecodingTask(model, mine.DecodingOptions(**bad))
            with pytest.raises(ValueError):
                ref.decoding.DecodingTask(model, ref.DecodingOptions(**bad))


def test_row_prompts_setup_matches_reference_per_row(ref):
    """DecodingTask(prompts=[...]) (our addition): every row's initial tokens are exactly what
    the reference's DecodingTask builds for that prompt alone (decoding.py:610-632); lags, the longest-row layout
    and the batching classes of transcribe_batch follow from the lengths"""
    from whisper_amd import decoding as mine
    from whisper_amd.transcribe import _prompt_batches
    model = _fake_model(True)
    prompts = [None, [5], list(range(2000, 2006)), list(range(1000, 1300)), []]
    for kw in (dict(), dict(without_timestamps=True), dict(prefix=[400, 500], sample_len=40)):
        task = mine.DecodingTask(model, mine.DecodingOptions(language="en", **kw), prompts=prompts)
        for p, row in zip(prompts, task.row_tokens):
            b = ref.decoding.DecodingTask(model, ref.DecodingOptions(language="en", prompt=p, **kw))
            assert row == b.initial_tokens
        longest = max(task.row_tokens, key=len)
        assert task.initial_tokens == longest and task.sample_begin == len(longest)
        assert task.row_lag == [len(longest) - len(r) for r in task.row_tokens]
        for r, lag in zip(task.row_tokens, task.row_lag):
            assert r.index(task.tokenizer.sot) == task.sot_index - lag        # rows differ only in the leading prompt
        assert task.ragged_limit() == 448 - task.sample_len
    with pytest.raises(ValueError):
        mine.DecodingTask(model, mine.DecodingOptions(language="en", prompt=[3]), prompts=prompts)
    # beam search runs on the device too (wh_task_beam carries a lag per segment): same limit; 9 beams fall back to the
    # host loop, which cannot place rows at different positions
    assert mine.DecodingTask(model, mine.DecodingOptions(language="en", beam_size=2), prompts=prompts).ragged_limit() == 448 - 224
    assert mine.DecodingTask(model, mine.DecodingOptions(language="en", beam_size=9), prompts=prompts).ragged_limit() is None
    assert mine.DecodingTask(model, mine.DecodingOptions(language="en", temperature=0.4, best_of=3),
                             prompts=prompts).ragged_limit() == 448 - 224       # sampling runs on the device too
    by_index = dict(enumerate(prompts))
    members = list(by_index)
    # greedy: everything below the limit shares one ragged class; the saturated prompt (227 initial tokens) is alone
    assert _prompt_batches(model, mine.DecodingOptions(language="en"), by_index, members, 16) == [[0, 1, 2, 4], [3]]
    assert _prompt_batches(model, mine.DecodingOptions(language="en"), by_index, members, 3) == [[0, 1, 2], [4], [3]]
    # beam search on the device batches like greedy; beyond 8 beams (host loop) only rows of equal length (no prompt ==
    # empty prompt) share a call
    assert _prompt_batches(model, mine.DecodingOptions(language="en", beam_size=2), by_index, members, 16) == [[0, 1, 2, 4], [3]]
    assert _prompt_batches(model, mine.DecodingOptions(language="en", beam_size=9), by_index, members, 16) == \
        [[0, 4], [1], [2], [3]]


class _ScriptedModel:
    """stands in for a Whisper model inside transcribe(): `decode` returns scripted DecodingResults"""

    def __init__(self, result_cls, tokenizer, scripts, multilingual=True):
        fm = _fake_model(multilingual)
        self.dims, self.is_multilingual, self.num_languages, self.device = fm.dims, fm.is_multilingual, fm.num_languages, fm.device
        self.result_cls, self.tk, self.scripts, self.calls = result_cls, tokenizer, scripts, []

    def decode(self, segment, options):
        i = min(len(self.calls), len(self.scripts) - 1)
        self.calls.append((tuple(segment.shape), options.temperature, tuple(options.prompt or ())))
        spec = self.scripts[i]
        toks = spec["tokens"]
        text = self.tk.decode(toks).strip()
        return self.result_cls(audio_features=torch.zeros(1), language="en", tokens=toks, text=text,
                               avg_logprob=spec.get("avg_logprob", -0.3), no_speech_prob=spec.get("no_speech_prob", 0.01),
                               temperature=options.temperature, compression_ratio=spec.get("compression_ratio", 1.2))


def test_transcribe_seek_logic_matches_reference(ref, monkeypatch):
    """the 30 s window state machine (transcribe.py:272-508): same scripted decoder outputs -> same segments,
    same seek sequence, same prompts, same fallback behaviour"""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]   # the package attribute of that name is the function
    from whisper_amd import decoding as mine
    tk, rtk = _tok(ref)
    TB = tk.timestamp_begin
    hello = tk.encode(" hello there")
    world = tk.encode(" general kenobi")
    scripts = [
        {"tokens": [TB, *hello, TB + 200, TB + 200, *world, TB + 450, TB + 450, *hello]},          # pairs, unfinished tail
        {"tokens": [TB, *world, TB + 300], "compression_ratio": 3.0},                                 # triggers fallback
        {"tokens": [TB, *world, TB + 300]},                                                           # single ts ending
        {"tokens": [*hello, *world], "no_speech_prob": 0.9, "avg_logprob": -2.0},                     # skipped: silence
        {"tokens": [TB + 10, *hello, TB + 100, TB + 100, *hello, TB + 700, TB + 700]},                # consecutive end
        {"tokens": [*hello]},                                                                          # no timestamps
    ]
    rng = np.random.default_rng(0)
    audio = (rng.standard_normal(16000 * 100) * 0.01).astype(np.float32)
    filt = oracle.mel_filterbank(80)

    def cpu_mel(a, n_mels=80, padding=0, device=None):
        return oracle.log_mel_spectrogram(a, filt, padding=padding)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", cpu_mel)

    for kw in (dict(temperature=(0.0, 0.4)), dict(temperature=0.0, condition_on_previous_text=False),
               dict(temperature=(0.0, 0.4), initial_prompt="Star Wars", carry_initial_prompt=True),
               dict(temperature=0.0, clip_timestamps="5,40,55"), dict(temperature=0.0, no_speech_threshold=None)):
        ma = _ScriptedModel(mine.DecodingResult, tk, scripts)
        mb = _ScriptedModel(ref.DecodingResult, rtk, scripts)
        ra = mine_tr.transcribe(ma, audio, language="en", fp16=False, **kw)
        rb = ref.transcribe(mb, audio, language="en", fp16=False, **kw)
        assert ma.calls == mb.calls
        assert ra["text"] == rb["text"] and ra["language"] == rb["language"]
        assert ra["segments"] == rb["segments"]


class _FunctionalModel:
    """decode() is a deterministic function of (window content, temperature, prompt): batched or one at a time, a
    window must come out the same.  A window "fails" (compression ratio 3.0 -> fallback) below a per-window
    temperature threshold, so windows climb the ladder to different rungs."""

    def __init__(self, result_cls, tokenizer):
        fm = _fake_model(True)
        self.dims, self.is_multilingual, self.num_languages, self.device = fm.dims, fm.is_multilingual, fm.num_languages, fm.device
        self.decoder = fm.decoder
        self.result_cls, self.tk, self.calls, self.detect_calls = result_cls, tokenizer, [], []

    def detect_language(self, mel):
        """language from the content of the first window: single (n_mels, 3000) or batched"""
        single = mel.ndim == 2
        mels = mel[None] if single else mel
        self.detect_calls.append(mels.shape[0])
        langs = [("en", "de", "fr")[int(abs(float(m.double().sum())) * 1000) % 3] for m in mels]
        probs = [{c: (0.9 if c == lang else 0.05) for c in ("en", "de", "fr")} for lang in langs]
        toks = torch.tensor([self.tk.to_language_token(lang) for lang in langs])
        return (toks[0], probs[0]) if single else (toks, probs)

    def _one(self, mel, options, prompt):
        key = int(abs(float(mel.double().sum())) * 1000) % 9973
        TB = self.tk.timestamp_begin
        words = self.tk.encode(" " + " ".join(["alpha", "bravo", "charlie", "delta", "echo"][: 1 + key % 5])
                               + {"en": "", "de": " und", "fr": " et", None: " ?"}[options.language])
        need = (0.0, 0.0, 0.2, 0.4, 0.2)[key % 5]                    # lowest temperature at which this window is fine
        ok = options.temperature >= need - 1e-9
        toks = [TB, *words, TB + 100 + key % 400, TB + 100 + key % 400, *words[: 1 + (len(prompt or ()) % 2)], TB + 700]
        return self.result_cls(audio_features=torch.zeros(1), language="en", tokens=toks,
                               text=self.tk.decode(toks).strip(), avg_logprob=-0.3 - 0.01 * options.temperature,
                               no_speech_prob=0.01, temperature=options.temperature,
                               compression_ratio=1.2 if ok else 3.0)

    def decode(self, segment, options, prompts=None):
        if segment.ndim == 2:
            self.calls.append((1, options.temperature))
            return self._one(segment, options, options.prompt)
        self.calls.append((segment.shape[0], options.temperature))
        assert options.prompt is None and prompts is not None and len(prompts) == segment.shape[0]
        return [self._one(m, options, p) for m, p in zip(segment, prompts)]


def test_transcribe_batch_ladder_equals_file_by_file(monkeypatch):
    """transcribe_batch: windows of several files are decoded in batches with per-row prompts and climb the
    temperature ladder together (transcribe.py:184-224), and still every file comes out exactly as from transcribe()"""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    filt = oracle.mel_filterbank(80)

    def cpu_mel(a, n_mels=80, padding=0, device=None):
        return oracle.log_mel_spectrogram(a, filt, padding=padding)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", cpu_mel)
    rng = np.random.default_rng(1)
    files = [(rng.standard_normal(16000 * n) * 0.01).astype(np.float32) for n in (95, 40, 130, 61, 20)]
    for kw in (dict(temperature=(0.0, 0.2, 0.4, 0.6)), dict(temperature=(0.0, 0.2), condition_on_previous_text=False),
               dict(temperature=(0.0, 0.2, 0.4), beam_size=3, best_of=2)):
        ma, mb = _FunctionalModel(mine.DecodingResult, tk), _FunctionalModel(mine.DecodingResult, tk)
        want = [mine_tr.transcribe(ma, a, language="en", fp16=False, **kw) for a in files]
        got = mine_tr.transcribe_batch(mb, files, language="en", fp16=False, batch_size=4, **kw)
        assert [g["segments"] for g in got] == [w["segments"] for w in want]
        assert [g["text"] for g in got] == [w["text"] for w in want]
        assert {s["temperature"] for w in want for s in w["segments"]} >= {0.0, 0.2}          # the ladder was climbed
        assert sum(n for n, _ in mb.calls) == sum(n for n, _ in ma.calls)                      # same decodes in total
        assert len(mb.calls) < len(ma.calls)                                                   # ... in fewer calls
        assert any(n > 1 and t > 0 for n, t in mb.calls)                                       # retries were batched
        assert all(n <= 4 for n, _ in mb.calls)


def _beam_update_model(state, cand_lp, cand_tok, first, G, K, eot, max_candidates):
    """Line-by-line Python model of beam_update_kernel (whisper_amd/csrc/beam.hip): one segment per workgroup, fp32
    scores, rank = stable descending order, walk until G sequences are kept.  `state`: tokens [R][len], sums [R] fp32,
    fin (list of (sequence, score) per segment, in insertion order), done (flags of the previous update)."""
    tokens, sums, fin, done_prev = state["tokens"], state["sums"], state["fin"], state["done"]
    B = len(fin)
    if all(done_prev):                                   # completed: later updates leave everything untouched
        return dict(tokens=[list(r) for r in tokens], sums=sums.copy(), fin=fin, done=[1] * B), list(range(len(tokens)))
    new_tokens, new_sums, src, done_next = [None] * len(tokens), sums.copy(), [None] * len(tokens), [0] * B
    for au in range(B):
        r0, N = au * G, G * K
        score = np.full(N, np.nan, np.float32)
        ctok, csrc = np.zeros(N, int), np.zeros(N, int)
        for c in range(N):
            j, kk = divmod(c, K)
            if (j == G - 1) if first else True:          # first update: every beam holds the same prefix
                score[c] = np.float32(sums[r0 + j]) + np.float32(cand_lp[r0 + j][kk])
            ctok[c], csrc[c] = cand_tok[r0 + j][kk], r0 + j
        order = {}
        for c in range(N):
            if score[c] != score[c]:
                continue
            rank = sum(1 for o in range(N) if score[o] == score[o] and (score[o] > score[c] or (score[o] == score[c] and o < c)))
            order[rank] = c
        kept, newly = [], []
        for i in range(K if first else N):
            if len(kept) >= G:
                break
            c = order[i]
            (newly if ctok[c] == eot else kept).append(c)
        for c in newly:
            if len(fin[au]) >= max_candidates:
                break
            fin[au].append((tuple(tokens[csrc[c]]) + (eot,), float(score[c])))
        for b, c in enumerate(kept):
            new_tokens[r0 + b] = list(tokens[csrc[c]]) + [int(ctok[c])]
            src[r0 + b], new_sums[r0 + b] = int(csrc[c]), score[c]
        done_next[au] = 1 if len(fin[au]) >= max_candidates else 0
    return dict(tokens=new_tokens, sums=new_sums, fin=fin, done=done_next), src


def test_device_beam_bookkeeping_model_equals_beam_search_decoder():
    """the algorithm beam_update_kernel implements (modelled above, candidate lists = log_softmax(...).topk(G + 1) as the
    row kernel produces them) against BeamSearchDecoder.update — itself checked against the reference in
    test_token_decoders_match_reference — over random multi-step runs: live beams, sums (bit-equal fp32), finished
    lists in dict order, KV source rows, completion flag; patience above and below 1; EOT made likely so that lists
    fill up and runs complete."""
    import torch.nn.functional as F
    from whisper_amd.decoding import BeamSearchDecoder

    class Inf:
        def rearrange_kv_cache(self, src):
            self.src = list(src)

    rng = np.random.default_rng(0)
    for trial in range(120):
        G, B, V = int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(12, 40))
        eot, patience = V - 3, float(rng.choice([1.0, 1.0, 2.0, 0.5]))
        inf = Inf()
        dec = BeamSearchDecoder(G, eot, inf, patience)
        K = G + 1
        tokens, sums = torch.tensor([[1, 2, 3]] * (B * G)), torch.zeros(B * G)
        st = dict(tokens=[[1, 2, 3] for _ in range(B * G)], sums=np.zeros(B * G, np.float32), fin=[[] for _ in range(B)], done=[0] * B)
        completed = False
        for step in range(12):
            lg = torch.tensor(rng.standard_normal((B, V)).astype(np.float32)).repeat_interleave(G, 0) if step == 0 \
                else torch.tensor(rng.standard_normal((B * G, V)).astype(np.float32))
            lg[:, eot] += float(rng.choice([0, 1.5, 3.0]))
            lp = F.log_softmax(lg.float(), -1).numpy()
            order = np.lexsort((np.arange(V)[None, :].repeat(B * G, 0), -lp), axis=-1)[:, :K]   # value desc, index asc
            if not completed:
                tokens, completed = dec.update(tokens, lg.clone(), sums)
            st, src = _beam_update_model(st, np.take_along_axis(lp, order, 1), order, step == 0, G, K, eot, dec.max_candidates)
            assert [list(r) for r in st["tokens"]] == tokens.tolist(), (trial, step)
            assert np.array_equal(st["sums"], sums.numpy()), (trial, step)
            for a in range(B):
                assert [k for k, _ in st["fin"][a]] == list(dec.finished_sequences[a].keys()), (trial, step, a)
                assert [v for _, v in st["fin"][a]] == list(dec.finished_sequences[a].values()), (trial, step, a)
            assert bool(all(st["done"])) == bool(completed)
            if not all(st["done"]) and step > 0:
                # same caches: a new beam continues the row the decoder names (prefixes are distinct after step 0)
                assert src == inf.src, (trial, step)
            if completed and step > 8:
                break


def test_word_timing_host_logic_matches_reference(ref, monkeypatch):
    """the host half of word timestamps (timing.py:57-80 backtrace, :245-276 merge_punctuations, :279-388
    add_word_timestamps with its duration-clipping heuristics): same inputs -> same words / boundaries as the live
    reference; the alignment itself (the device half) is stubbed with the same scripted WordTimings on both sides"""
    import copy
    import whisper_amd  # noqa: F401
    from whisper_amd import timing as mine
    rt = ref.timing
    rng = np.random.default_rng(5)

    # backtrace: random valid trace matrices (first row = 2, first column = 1, as dtw leaves them)
    for n, m in ((5, 9), (17, 40), (60, 33), (1, 7)):
        trace = rng.integers(0, 3, (n + 1, m + 1)).astype(np.float32)
        trace[0, :], trace[:, 0] = 2, 1
        assert np.array_equal(mine.backtrace(trace.copy()), rt.backtrace(trace.copy()))

    # merge_punctuations on random word lists with plenty of punctuation
    vocab = [" hello", " world", ",", ".", " \"", "\"", " (", ")", " -", "!", " there", "?", " ¿", " que", ":"]
    pre, app = "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、"
    for _ in range(200):
        words = [vocab[i] for i in rng.integers(0, len(vocab), int(rng.integers(1, 12)))]
        a = [mine.WordTiming(w, [i], 0.1 * i, 0.1 * i + 0.05, 0.9) for i, w in enumerate(words)]
        b = [rt.WordTiming(w, [i], 0.1 * i, 0.1 * i + 0.05, 0.9) for i, w in enumerate(words)]
        mine.merge_punctuations(a, pre, app)
        rt.merge_punctuations(b, pre, app)
        assert [(x.word, x.tokens) for x in a] == [(x.word, x.tokens) for x in b]

    # add_word_timestamps: scripted alignment with pauses, over-long words and sentence marks
    tk, rtk = _tok(ref)
    for trial in range(60):
        n_seg = int(rng.integers(1, 4))
        segments, all_words, t = [], [], float(rng.uniform(0, 3))
        for s in range(n_seg):
            n_w = int(rng.integers(1, 7))
            toks, seg_start = [], t
            for w in range(n_w):
                word = [" alpha", " beta", ".", " gamma", "!", " delta", ","][int(rng.integers(0, 7))]
                wt = tk.encode(word)
                dur = float(rng.choice([0.1, 0.2, 0.3, 0.5, 1.5, 3.0]))
                gap = float(rng.choice([0.0, 0.0, 0.1, 2.5]))
                all_words.append((word, wt, round(t + gap, 2), round(t + gap + dur, 2), float(rng.uniform(0.3, 1.0))))
                toks += wt
                t += gap + dur
            segments.append(dict(seek=int(rng.choice([0, 3000])), start=round(seg_start + float(rng.uniform(-0.8, 0.8)), 2),
                                 end=round(t + float(rng.uniform(-0.8, 0.8)), 2), tokens=[tk.timestamp_begin] + toks + [tk.timestamp_begin + 50]))
        for s in segments:
            s["seek"] = segments[0]["seek"]
        monkeypatch.setattr(mine, "find_alignment", lambda *a, **k: [mine.WordTiming(*w) for w in all_words])
        monkeypatch.setattr(rt, "find_alignment", lambda *a, **k: [rt.WordTiming(*w) for w in all_words])
        sa, sb = copy.deepcopy(segments), copy.deepcopy(segments)
        last = float(rng.choice([0.0, 1.0, 4.0]))
        mine.add_word_timestamps(segments=sa, model=None, tokenizer=tk, mel=None, num_frames=3000, last_speech_timestamp=last)
        rt.add_word_timestamps(segments=sb, model=None, tokenizer=rtk, mel=None, num_frames=3000, last_speech_timestamp=last)
        assert sa == sb, trial


def test_transcribe_word_timestamp_seeking_matches_reference(ref, monkeypatch):
    """transcribe.py:401-470 — word_timestamps=True changes how windows advance (seek to the last word's end) and
    enables hallucination_silence_threshold (anomaly scores, skipping silence before / after suspicious segments).
    The word aligner is stubbed identically on both sides (deterministic words with a few very long and very
    improbable ones), the decoder is scripted: same segments, words and seek sequence as the live reference."""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    ref_tr = sys.modules["whisper.transcribe"]
    from whisper_amd import decoding as mine
    tk, rtk = _tok(ref)
    TB = tk.timestamp_begin
    hello, world = tk.encode(" hello there"), tk.encode(" general kenobi you are bold")
    scripts = [
        {"tokens": [TB, *hello, TB + 200, TB + 200, *world, TB + 450, TB + 450, *hello]},
        {"tokens": [TB + 20, *world, TB + 300]},
        {"tokens": [TB + 600, *hello, TB + 700, TB + 700, *world, TB + 1400, TB + 1400]},
        {"tokens": [TB, *hello, TB + 100, TB + 100, *hello, TB + 220, TB + 400, *world, TB + 1000, TB + 1000]},
        {"tokens": [*hello, *world]},
        {"tokens": [TB + 50, *world, TB + 1200, TB + 1200, *hello, TB + 1490]},
    ]

    def fake_words(*, segments, model, tokenizer, mel, num_frames, prepend_punctuations="", append_punctuations="",
                   last_speech_timestamp, **kw):
        # words spread over the segment; every 5th word over-long, every 7th improbable -> anomaly scores vary
        counter = 0
        for seg in segments:
            toks = [t for t in seg["tokens"] if t < tokenizer.eot]
            n = max(1, len(toks) // 2)
            span = max(seg["end"] - seg["start"], 0.2)
            words, t0 = [], seg["start"]
            for i in range(n):
                counter += 1
                dur = span / n * (3.5 if counter % 5 == 0 else 0.6)
                words.append(dict(word=f" w{counter}", start=round(t0, 2), end=round(t0 + dur, 2),
                                  probability=0.05 if counter % 7 == 0 else 0.9))
                t0 += span / n
            seg["words"] = words
    monkeypatch.setattr(mine_tr, "add_word_timestamps", fake_words)
    monkeypatch.setattr(ref_tr, "add_word_timestamps", fake_words)
    # the state machine asks its driver for the alignment (so that transcribe_batch can batch it); stand-in aligner
    monkeypatch.setattr(mine_tr, "find_alignment", lambda *a, **k: [])
    filt = oracle.mel_filterbank(80)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", lambda a, n_mels=80, padding=0, device=None: oracle.log_mel_spectrogram(a, filt, padding=padding))
    rng = np.random.default_rng(2)
    audio = (rng.standard_normal(16000 * 140) * 0.01).astype(np.float32)
    for kw in (dict(word_timestamps=True), dict(word_timestamps=True, hallucination_silence_threshold=1.0),
               dict(word_timestamps=True, hallucination_silence_threshold=0.2, condition_on_previous_text=False),
               dict(word_timestamps=True, clip_timestamps="10,70,80,130", hallucination_silence_threshold=2.0)):
        ma = _ScriptedModel(mine.DecodingResult, tk, scripts)
        mb = _ScriptedModel(ref.DecodingResult, rtk, scripts)
        ra = mine_tr.transcribe(ma, audio, language="en", fp16=False, temperature=0.0, **kw)
        rb = ref.transcribe(mb, audio, language="en", fp16=False, temperature=0.0, **kw)
        assert ma.calls == mb.calls, kw
        assert ra["text"] == rb["text"]
        assert ra["segments"] == rb["segments"], kw
        assert len(ma.calls) >= 4 and any("words" in s for s in ra["segments"])


def test_transcribe_reuses_the_windows_encoder_output(monkeypatch):
    """Two places where the reference encodes a window again and this package does not: the temperature-fallback retry
    (transcribe.py:184-224 calls decode() on the mel every time) is handed DecodingResult.audio_features of the first
    attempt — decode() accepts encoded input (decoding.py:655-662) — and the word aligner receives the same tensor
    through `audio_features=` instead of running the encoder on the mel (timing.py:199)."""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    TB = tk.timestamp_begin
    hello = tk.encode(" hello there")
    fm = _fake_model(True)
    feats = torch.arange(fm.dims.n_audio_ctx * fm.dims.n_audio_state, dtype=torch.float32).reshape(
        fm.dims.n_audio_ctx, fm.dims.n_audio_state)
    seen_inputs, seen_align = [], []

    class M:
        dims, is_multilingual, num_languages, device = fm.dims, True, 99, fm.device

        def decode(self, segment, options):
            seen_inputs.append((tuple(segment.shape), options.temperature))
            bad = len(seen_inputs) == 1                     # the first attempt fails the compression-ratio test
            return mine.DecodingResult(audio_features=feats, language="en", tokens=[TB, *hello, TB + 300],
                                       text=" hello there", avg_logprob=-0.3, no_speech_prob=0.01,
                                       temperature=options.temperature, compression_ratio=3.0 if bad else 1.2)

    def aligner(model, tokenizer, text_tokens, mel, num_frames, **kw):
        seen_align.append(kw.get("audio_features"))
        return []
    monkeypatch.setattr(mine_tr, "find_alignment", aligner)
    filt = oracle.mel_filterbank(80)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", lambda a, n_mels=80, padding=0, device=None: oracle.log_mel_spectrogram(a, filt, padding=padding))
    audio = (np.random.default_rng(0).standard_normal(16000 * 8) * 0.01).astype(np.float32)
    r = mine_tr.transcribe(M(), audio, language="en", fp16=False, temperature=(0.0, 0.4), word_timestamps=True)
    assert r["text"].strip() == "hello there"
    assert seen_inputs[0] == ((80, 3000), 0.0)                                   # the mel window
    assert seen_inputs[1] == ((fm.dims.n_audio_ctx, fm.dims.n_audio_state), 0.4)  # the retry: encoded features
    assert len(seen_align) == 1 and seen_align[0] is feats


def test_detect_language_host_logic_matches_reference(ref):
    """decoding.py:18-77: given the same logits at the <|startoftranscript|> position, the language mask, arg-max and
    probability dictionaries equal the reference's (v2 = 99 and v3 = 100 languages; single and batched inputs;
    English-only models refuse)"""
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    rng = np.random.default_rng(4)
    for n_vocab in (51865, 51866):
        fm = _fake_model(True)
        dims = SimpleNamespace(**{**fm.dims.__dict__, "n_vocab": n_vocab})
        table = torch.tensor(rng.standard_normal((3, n_vocab)).astype(np.float32) * 3)

        class M:
            is_multilingual, num_languages = True, n_vocab - 51765 - 1

            def __init__(self):
                self.dims = dims

            def logits(self, x, feats):
                assert x.shape == (feats.shape[0], 1)
                return table[: feats.shape[0], None, :].clone()

        feats = torch.zeros(3, dims.n_audio_ctx, dims.n_audio_state)
        tk = get_tokenizer(True, num_languages=M.num_languages)
        rtk = ref.tokenizer.get_tokenizer(True, num_languages=M.num_languages)
        ta, pa = mine.detect_language(M(), feats, tk)
        tb, pb = ref.decoding.detect_language(M(), feats, rtk)
        assert ta.tolist() == tb.tolist() and pa == pb and len(pa[0]) == M.num_languages
        ta, pa = mine.detect_language(M(), feats[0], tk)
        tb, pb = ref.decoding.detect_language(M(), feats[0], rtk)
        assert int(ta) == int(tb) and pa == pb
    with pytest.raises(ValueError):
        mine.detect_language(_fake_model(False), torch.zeros(1500, 384), get_tokenizer(False))
    with pytest.raises(ValueError):
        ref.decoding.detect_language(_fake_model(False), torch.zeros(1500, 384), ref.tokenizer.get_tokenizer(False))


def test_transcribe_batch_in_flight_groups_and_merges(monkeypatch):
    """transcribe_batch(in_flight=k): the files are dealt round-robin into k groups, every group is an ordinary
    transcribe_batch call handed to decoding.run_in_lanes (one host thread + HIP stream per group on a GPU; a sequential
    stand-in here), and the results come back in INPUT order and equal to in_flight=1.  An exception in a lane surfaces."""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    filt = oracle.mel_filterbank(80)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", lambda a, n_mels=80, padding=0, device=None: oracle.log_mel_spectrogram(a, filt, padding=padding))
    seen = []

    def fake_lanes(model, jobs, in_flight=3, dtype=None):
        jobs = list(jobs)
        seen.append((len(jobs), in_flight, dtype))
        return [j() for j in jobs]
    monkeypatch.setattr(mine, "run_in_lanes", fake_lanes)
    rng = np.random.default_rng(3)
    files = [(rng.standard_normal(16000 * n) * 0.01).astype(np.float32) for n in (40, 95, 20, 61, 33)]
    kw = dict(language="en", fp16=False, temperature=(0.0, 0.2), batch_size=2)
    want = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), files, **kw)
    assert not seen                                                              # in_flight = 1: no lanes
    m = _FunctionalModel(mine.DecodingResult, tk)
    got = mine_tr.transcribe_batch(m, files, in_flight=2, **kw)
    assert seen == [(2, 2, torch.float32)]                                       # two groups: files 0, 2, 4 and 1, 3
    assert [g["text"] for g in got] == [w["text"] for w in want] and [g["segments"] for g in got] == [w["segments"] for w in want]
    assert all(n <= 2 for n, _ in m.calls)
    seen.clear()
    got = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), files, in_flight=8, **kw)
    assert seen == [(5, 5, torch.float32)] and [g["text"] for g in got] == [w["text"] for w in want]     # never more groups than files
    assert mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), files[:1], in_flight=3, **kw)[0]["text"] == want[0]["text"]


def test_transcribe_batch_loads_files_concurrently(monkeypatch, tmp_path):
    """paths given to transcribe_batch are decoded by a thread pool up front (same results as arrays, input order kept,
    a failing file raises the loader's error)"""
    import threading
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    filt = oracle.mel_filterbank(80)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram",
                        lambda a, n_mels=80, padding=0, device=None: oracle.log_mel_spectrogram(a, filt, padding=padding))
    rng = np.random.default_rng(3)
    arrays = {str(tmp_path / f"f{i}.wav"): (rng.standard_normal(16000 * n) * 0.01).astype(np.float32)
              for i, n in enumerate((35, 50, 20, 64))}
    seen = []

    def fake_load(path, sr=16000):
        seen.append((path, threading.get_ident()))
        if path.endswith("missing.wav"):
            raise RuntimeError(f"Failed to load audio: {path}")
        return arrays[path]
    monkeypatch.setattr(mine_tr, "load_audio", fake_load)
    kw = dict(language="en", fp16=False, temperature=(0.0, 0.2, 0.4))
    want = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), list(arrays.values()), **kw)
    got = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), list(arrays), **kw)
    assert [g["segments"] for g in got] == [w["segments"] for w in want]
    assert sorted(p for p, _ in seen) == sorted(arrays) and threading.get_ident() not in {t for _, t in seen}
    mixed = [list(arrays)[0], arrays[list(arrays)[1]], list(arrays)[2]]
    got = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), mixed, **kw)
    assert [g["text"] for g in got] == [w["text"] for w in want[:3]]
    with pytest.raises(RuntimeError, match="Failed to load audio"):
        mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), list(arrays) + [str(tmp_path / "missing.wav")], **kw)


def test_transcribe_batch_bounds_active_files(monkeypatch, tmp_path):
    """memory follows the window, not the total audio: with max_active_files = 2 at most two files are loaded and hold
    a whole-file spectrogram at any time, a new file starts only when one finishes, and the results equal the
    unbounded run (file by file state machines are independent)"""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    filt = oracle.mel_filterbank(80)
    live = {"mels": 0, "peak": 0, "loads": []}

    class CountedMel(torch.Tensor):
        pass

    def counted_mel(a, n_mels=80, padding=0, device=None):
        m = oracle.log_mel_spectrogram(a, filt, padding=padding)
        live["mels"] += 1
        live["peak"] = max(live["peak"], live["mels"])
        import weakref
        weakref.finalize(m, lambda: live.__setitem__("mels", live["mels"] - 1))
        return m
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram", counted_mel)
    rng = np.random.default_rng(5)
    arrays = {str(tmp_path / f"g{i}.wav"): (rng.standard_normal(16000 * n) * 0.01).astype(np.float32)
              for i, n in enumerate((35, 64, 20, 50, 31, 45))}

    def fake_load(path, sr=16000):
        live["loads"].append((path, live["mels"]))
        return arrays[path]
    monkeypatch.setattr(mine_tr, "load_audio", fake_load)
    kw = dict(language="en", fp16=False, temperature=(0.0, 0.2))
    want = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), list(arrays), **kw)
    import gc
    gc.collect()
    live.update(mels=0, peak=0, loads=[])
    got = mine_tr.transcribe_batch(_FunctionalModel(mine.DecodingResult, tk), list(arrays), max_active_files=2, **kw)
    gc.collect()
    assert [g["segments"] for g in got] == [w["segments"] for w in want]
    assert live["peak"] <= 2, live
    assert [p for p, _ in live["loads"]] == list(arrays)          # admitted in input order, later ones only after a finish
    assert all(n <= 1 for _, n in live["loads"][2:]), live["loads"]


def test_loader_and_filter_error_conventions(ref, tmp_path):
    """the error behaviour of the entry points a caller can hit without a GPU: unknown model name and
    checksum mismatch -> RuntimeError (whisper/__init__.py:143-145, 66-93), unsupported n_mels -> AssertionError
    (audio.py:103), with the reference raising the same type on the same input; a cached file with the right digest
    is reused without touching the network, a wrong one is re-fetched (here from a file:// URL)"""
    import hashlib
    import whisper_amd
    from whisper_amd import audio as mine_audio
    with pytest.raises(RuntimeError, match="not found; available models"):
        whisper_amd.load_model("no-such-model", device="cpu")
    with pytest.raises(RuntimeError, match="not found; available models"):
        ref.load_model("no-such-model", device="cpu")
    assert whisper_amd.available_models() == ref.available_models()
    for fn in (lambda: mine_audio.mel_filters("cpu", 64), lambda: ref.audio.mel_filters("cpu", 64)):
        with pytest.raises(AssertionError, match="Unsupported n_mels"):
            fn()

    payload = b"checkpoint bytes " * 1000
    digest = hashlib.sha256(payload).hexdigest()
    src = tmp_path / "srv" / digest
    src.mkdir(parents=True)
    (src / "tiny.pt").write_bytes(payload)
    url = f"file://{src}/tiny.pt"
    root = tmp_path / "cache"
    got = whisper_amd._fetch(url, str(root), in_memory=False)                     # fetched and verified
    assert got == str(root / "tiny.pt") and (root / "tiny.pt").read_bytes() == payload
    assert whisper_amd._fetch(url, str(root), in_memory=True) == payload           # cached copy, right digest
    (root / "tiny.pt").write_bytes(b"corrupted")
    with pytest.warns(UserWarning, match="checksum does not match"):
        assert whisper_amd._fetch(url, str(root), in_memory=True) == payload       # re-fetched
    bad = tmp_path / "srv" / ("0" * 64)
    bad.mkdir()
    (bad / "base.pt").write_bytes(payload)
    with pytest.raises(RuntimeError, match="checksum does not not match"):
        whisper_amd._fetch(f"file://{bad}/base.pt", str(root), in_memory=False)
    (root / "dir.pt").mkdir()
    with pytest.raises(RuntimeError, match="not a regular file"):
        whisper_amd._fetch(f"file://{src}/dir.pt", str(root), in_memory=False)


def test_kv_cache_hooks_surface(monkeypatch):
    """Whisper.install_kv_cache_hooks / model.decoder(..., kv_cache=) keep the reference's calling convention
    (model.py:227-249, 310-341): first call feeds every token, later calls the new ones, logits for every token fed,
    hook.remove() releases the cache.  The device task is replaced by a recorder here; the GPU test
    tests/test_api_gpu.py::test_incremental_decoder_with_kv_cache_hooks checks the numbers."""
    from whisper_amd import model as mm
    calls = []

    class FakeTask:
        def __init__(self, engine, n_audio, group, max_prefill):
            calls.append(("create", n_audio, group, max_prefill))
            self.position, self.closed = 0, False

        def set_audio(self, xa):
            calls.append(("audio", tuple(xa.shape)))

        def prefill(self, x):
            calls.append(("prefill", tuple(x.shape)))
            self.position += x.shape[1]
            return torch.zeros(x.shape[0], x.shape[1], 7)

        def step(self, last):
            calls.append(("step", tuple(last.shape)))
            self.position += 1
            return torch.zeros(last.shape[0], 7)

        def close(self):
            self.closed = True
            calls.append(("close",))
    class FakeEngine:          # tasks come from the engine's cache (HipModel.acquire_task)
        def acquire_task(self, n_audio, group, max_prefill, capture_q=False):
            return FakeTask(self, n_audio, group, max_prefill)
    fm = _fake_model(True)
    model = mm.Whisper(mm.ModelDimensions(**fm.dims.__dict__), {}, device="cpu")
    monkeypatch.setattr(model, "engine", lambda dtype: FakeEngine())
    cache, hooks = model.install_kv_cache_hooks()
    assert isinstance(cache, dict) and len(hooks) == 1
    xa = torch.zeros(2, 1500, 384)
    toks = torch.tensor([[1, 2, 3]] * 4)
    assert model.decoder(toks, xa, kv_cache=cache).shape == (4, 3, 7)
    assert model.decoder(toks[:, -1:], xa, kv_cache=cache).shape == (4, 1, 7)
    assert model.decoder(toks[:, :2], xa, kv_cache=cache).shape == (4, 2, 7)          # several new tokens at once
    task = cache[mm._TASK_KEY]
    for h in hooks:
        h.remove()
    assert task.closed and mm._TASK_KEY not in cache
    assert calls == [("create", 2, 2, 448), ("audio", (2, 1500, 384)), ("prefill", (4, 3)), ("step", (4,)),
                     ("prefill", (4, 2)), ("close",)]
    # a plain dict nobody hooked (reference model.py:96-104, 233-249: none of its modules is in it, so every key / value
    # is computed from the tokens passed; only the position offset is read from the first entry): an ordinary pass ...
    assert model.decoder(toks, xa, kv_cache={}).shape == (4, 3, 7)
    assert model.decoder(toks, xa, kv_cache={"note": torch.zeros(2, 0, 8)}).shape == (4, 3, 7)
    with pytest.raises(NotImplementedError):                 # ... and a non-zero offset without the earlier keys is refused
        model.decoder(toks, xa, kv_cache={"foreign": torch.zeros(2, 5, 8)})
    user = {"note": 1}
    cache2, hooks2 = model.install_kv_cache_hooks(user)
    assert cache2 is not user and cache2["note"] == 1                                   # copied, as the reference does


def test_api_surface_matches_reference(ref):
    """the names a caller of openai/whisper touches exist here with the same parameters, defaults and
    dataclass fields (extra keyword parameters of ours — `prompts`, `transcribe_batch` — are additions, never renames)"""
    import dataclasses
    import inspect
    import whisper_amd as mine

    def fields(cls):
        return [(f.name, f.default) for f in dataclasses.fields(cls)]
    assert fields(mine.DecodingOptions) == fields(ref.DecodingOptions)
    assert fields(mine.DecodingResult) == fields(ref.DecodingResult)
    assert fields(mine.ModelDimensions) == fields(ref.ModelDimensions)

    def params(fn):
        def plain(v):
            return dataclasses.asdict(v) if dataclasses.is_dataclass(v) and not isinstance(v, type) else v
        return [(p.name, plain(p.default), p.kind) for p in inspect.signature(fn).parameters.values()]
    for name in ("transcribe", "load_model", "pad_or_trim", "load_audio", "log_mel_spectrogram", "available_models",
                 "detect_language"):
        assert params(getattr(mine, name)) == params(getattr(ref, name)), name
    theirs, ours = params(ref.decode), params(mine.decode)
    assert [p for p in ours if p[0] != "prompts"] == theirs                       # + our per-segment prompts
    for attr in ("dims", "device", "is_multilingual", "num_languages", "encoder", "decoder", "logits", "embed_audio",
                 "alignment_heads", "set_alignment_heads", "install_kv_cache_hooks", "transcribe", "decode",
                 "detect_language", "forward"):
        assert hasattr(ref.model.Whisper, attr) or attr in ("dims", "encoder", "decoder", "alignment_heads"), attr
    fm = _fake_model(True)
    m = mine.Whisper(mine.ModelDimensions(**fm.dims.__dict__), {}, device="cpu")
    for attr in ("dims", "device", "is_multilingual", "num_languages", "encoder", "decoder", "logits", "embed_audio",
                 "alignment_heads", "set_alignment_heads", "install_kv_cache_hooks", "transcribe", "decode",
                 "detect_language", "forward"):
        assert hasattr(m, attr), attr
    assert m.is_multilingual and m.num_languages == 99 and m.device == torch.device("cpu")
    # the nn.Module surface a caller of the reference touches on the model object (model.py:252: Whisper is an nn.Module)
    for attr in ("parameters", "named_parameters", "state_dict", "load_state_dict", "half", "float", "to", "cuda", "cpu",
                 "eval", "train", "requires_grad_"):
        assert hasattr(ref.model.Whisper, attr) and callable(getattr(m, attr)), attr
    rdims = ref.ModelDimensions(**fm.dims.__dict__)
    theirs_sd = ref.model.Whisper(rdims).state_dict()
    from whisper_amd.model import expected_state_shapes
    want = expected_state_shapes(mine.ModelDimensions(**fm.dims.__dict__))
    assert {k: tuple(v.shape) for k, v in theirs_sd.items()} == want              # names and shapes = the reference's
    res = m.load_state_dict(theirs_sd)                                           # a reference state dict loads as is
    assert res.missing_keys == [] and res.unexpected_keys == []
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in ref.model.Whisper(rdims).named_parameters()]
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in ref.model.Whisper(rdims).parameters())
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in theirs_sd.items() if k != "decoder.ln.weight"})           # strict: missing key
    with pytest.raises(RuntimeError):
        m.load_state_dict({**theirs_sd, "decoder.ln.weight": torch.zeros(3)})                          # size mismatch
    loose = m.load_state_dict({"decoder.ln.weight": theirs_sd["decoder.ln.weight"], "extra": torch.zeros(1)}, strict=False)
    assert loose.unexpected_keys == ["extra"] and "decoder.ln.bias" in loose.missing_keys
    assert m.half() is m and m._half and m.float() is m and not m._half and m.eval() is m and m.train(False) is m
    # nn.Module.to as an inference user calls it: a dtype is half() / float(), a device moves, both at once work
    assert m.to(torch.float16) is m and m._half and m.to(dtype=torch.float32) is m and not m._half
    assert m.to("cpu", torch.float16) is m and m._half and m.device == torch.device("cpu") and m.to(device="cpu") is m
    m.float()
    with pytest.raises(TypeError):
        m.to(torch.int8)
    with pytest.raises(TypeError):
        m.to(torch.zeros(1))
    # load_state_dict COPIES (torch does): a later in-place edit of the caller's tensors does not reach the model
    mine_w = theirs_sd["decoder.ln.weight"].clone()
    m.load_state_dict({"decoder.ln.weight": mine_w}, strict=False)
    mine_w.add_(1.0)
    assert torch.equal(m.state_dict()["decoder.ln.weight"], theirs_sd["decoder.ln.weight"])
    with pytest.raises(TypeError):
        m.load_state_dict({"decoder.ln.weight": [1.0] * int(theirs_sd["decoder.ln.weight"].numel())}, strict=False)
    # alignment heads: default = upper half of the decoder layers (model.py:270-276); dumps decode like the reference's
    dense = m.alignment_heads.to_dense()
    assert dense.shape == (fm.dims.n_text_layer, fm.dims.n_text_head) and bool(dense[fm.dims.n_text_layer // 2:].all())
    assert not bool(dense[: fm.dims.n_text_layer // 2].any())
    from whisper_amd.registry import ALIGNMENT_HEADS, MODEL_URLS
    assert MODEL_URLS == ref._MODELS and ALIGNMENT_HEADS == ref._ALIGNMENT_HEADS
    import base64
    import gzip
    for name, (layers, heads) in (("tiny.en", (4, 6)), ("large-v3", (32, 20)), ("turbo", (4, 20))):
        mask = np.frombuffer(gzip.decompress(base64.b85decode(ALIGNMENT_HEADS[name])), dtype=bool)
        assert mask.size == layers * heads and mask.sum() == {"tiny.en": 8, "large-v3": 10, "turbo": 6}[name]


def test_transcribe_batch_detects_languages_in_one_pass(monkeypatch):
    """without `language`, transcribe_batch identifies the language of all files in batched passes (transcribe.py:139-152
    does it file by file); every file then decodes exactly as `transcribe` decodes it"""
    import oracle
    import whisper_amd  # noqa: F401
    mine_tr = sys.modules["whisper_amd.transcribe"]
    from whisper_amd import decoding as mine
    from whisper_amd.tokenizer import get_tokenizer
    tk = get_tokenizer(True, num_languages=99, language="en", task="transcribe")
    filt = oracle.mel_filterbank(80)
    monkeypatch.setattr(mine_tr, "log_mel_spectrogram",
                        lambda a, n_mels=80, padding=0, device=None: oracle.log_mel_spectrogram(a, filt, padding=padding))
    rng = np.random.default_rng(6)
    files = [(rng.standard_normal(16000 * n) * 0.01).astype(np.float32) for n in (33, 41, 25, 58, 36, 30, 47)]
    kw = dict(fp16=False, temperature=(0.0, 0.2, 0.4), verbose=None)
    ma, mb = _FunctionalModel(mine.DecodingResult, tk), _FunctionalModel(mine.DecodingResult, tk)
    want = [mine_tr.transcribe(ma, a, **kw) for a in files]
    got = mine_tr.transcribe_batch(mb, files, batch_size=4, **kw)
    assert [g["language"] for g in got] == [w["language"] for w in want] and len({w["language"] for w in want}) >= 2
    assert [g["segments"] for g in got] == [w["segments"] for w in want]
    assert ma.detect_calls == [1] * len(files) and mb.detect_calls == [4, 3]
    # a given language switches detection off, a single file keeps the reference's own flow
    mc = _FunctionalModel(mine.DecodingResult, tk)
    mine_tr.transcribe_batch(mc, files[:3], language="de", **kw)
    assert mc.detect_calls == []
    md = _FunctionalModel(mine.DecodingResult, tk)
    mine_tr.transcribe_batch(md, files[:1], **kw)
    assert md.detect_calls == [1]


def test_flash_attention_lds_tile_layouts_are_conflict_free_and_consistent():
    """Model of the encoder flash-attention kernel's LDS tiles (csrc/attention.hip, csrc/common.h::swz_byte) under the
    bank rules of the MI355X guide (64 banks x 4 B; ds_read_b128 serves four fixed groups of 16 lanes, ds_write_b64 two
    halves of 32): (1) the V^T tile, stored with the 4-key halves of neighbouring units exchanged, hands lane (row, hi)
    exactly the keys the score MFMA left in its P fragment; (2) its fragment reads and the K reads are conflict-free;
    (3) the plain layout read the same keys as 8-byte halves that collide two by two (what SQ_LDS_BANK_CONFLICT showed)."""
    swz = lambda row, unit: row * 128 + ((unit ^ ((row >> 1) & 7)) << 4)
    # (1) store: thread (row, cu) holds keys 8cu..8cu+7 of a 64-key tile row; low half -> unit cu&~1, high half -> cu|1
    lds = {}
    for row in range(64):
        for cu in range(8):
            for half, unit in ((0, cu & ~1), (1, cu | 1)):
                base = swz(row, unit) + (cu & 1) * 8
                for e in range(4):
                    addr = base + 2 * e
                    assert addr not in lds
                    lds[addr] = (row, 8 * cu + 4 * half + e)
    assert len(lds) == 64 * 64
    for lane in range(64):
        drow, hi = lane & 31, lane >> 5
        for dt in range(2):
            for kb in range(2):
                for s2 in range(2):
                    base = swz(dt * 32 + drow, kb * 4 + 2 * s2 + hi)
                    got = [lds[base + 2 * j] for j in range(8)]
                    # P fragment of (kb, s2): accumulator rows r = 8 s2 .. 8 s2 + 7 -> key (r & 3) + 8 (r >> 2) + 4 hi
                    want = [(dt * 32 + drow, kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) for r in range(8 * s2, 8 * s2 + 8)]
                    assert got == want
    # (2) bank conflicts of a 16-byte read: groups of 16 lanes, 16 slots of 4 banks
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for unit0 in range(0, 8, 2):
        for rowbase in (0, 32):
            for g in groups:
                slots = [(swz(rowbase + (l & 31), unit0 + (l >> 5)) // 16) % 16 for l in g]
                assert len(set(slots)) == 16
    # the two 8-byte stores of a staged unit: 32 lanes, 32 bank pairs
    for first in range(0, 512, 32):
        for which in (0, 1):
            pairs = []
            for u in range(first, first + 32):
                row, cu = u >> 3, u & 7
                unit = (cu & ~1) if which == 0 else (cu | 1)
                pairs.append(((swz(row, unit) + (cu & 1) * 8) // 8) % 32)
            assert len(set(pairs)) == 32
    # (3) the plain layout: lanes 0..31 read bytes hi*8.. of the same unit -> 16 distinct bank pairs for 32 lanes
    pairs = [((swz(l, 2) + 0) // 8) % 32 for l in range(32)]
    assert len(set(pairs)) == 16


def test_weight_packing_algebra():
    """What hip.pack_weights folds into the fp16 blob is exact algebra on the checkpoint tensors (float64 here):
    (1) LayerNorm affine parts folded into the consuming projection (WH_WEIGHTS_DEC_LN_FOLDED, model.py:39-50,142-171);
    (2) the encoder's softmax scale carried by the query / key projections (WH_WEIGHTS_ENC_QK_SCALED, model.py:118-121);
    (3) Conv1d as a GEMM over overlapping rows of the padded, transposed input (model.py:53-59,193-194)."""
    import math
    import torch
    from whisper_amd import hip
    from whisper_amd.synthetic import dims_for, synthetic_state_dict
    g = torch.Generator().manual_seed(3)
    D, N, R = 48, 80, 7
    W, b = torch.randn(N, D, generator=g, dtype=torch.float64), torch.randn(N, generator=g, dtype=torch.float64)
    gam, bet = torch.randn(D, generator=g, dtype=torch.float64), torch.randn(D, generator=g, dtype=torch.float64)
    x = torch.randn(R, D, generator=g, dtype=torch.float64) * 3 + 1
    xhat = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    want = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-5), W, b)
    Wf, bf = hip._fold_ln(W, b, gam, bet)
    got = xhat.float() @ Wf.T + bf                     # _fold_ln computes in fp32
    assert torch.allclose(got.double(), want, atol=2e-4, rtol=1e-5)

    # (2) pieces of an encoder block: rows [:2D] of qkv_w and the query bias carry ENC_QK_SCALE, the key bias is zero
    assert abs(hip.ENC_QK_SCALE ** 2 - 0.125 * math.log2(math.e)) < 1e-12
    dims = dims_for("tiny.en")
    sd = synthetic_state_dict(dims, seed=0, device="cpu")
    Da = dims.n_audio_state
    plain = dict((f, t) for f, t, _ in hip._block_pieces(sd, "encoder.blocks.0.", False, Da))
    scaled = dict((f, t) for f, t, _ in hip._block_pieces(sd, "encoder.blocks.0.", False, Da, qk_scale=hip.ENC_QK_SCALE))
    assert torch.equal(scaled["qkv_w"][2 * Da:].float(), plain["qkv_w"][2 * Da:].float())
    assert torch.allclose(scaled["qkv_w"][: 2 * Da].float(), plain["qkv_w"][: 2 * Da].float() * hip.ENC_QK_SCALE, rtol=1e-6)
    assert torch.all(scaled["qkv_b"][Da: 2 * Da] == 0) and torch.all(plain["qkv_b"][Da: 2 * Da] == 0)
    xe = torch.randn(5, Da, generator=g)
    q = (xe @ plain["qkv_w"][:Da].float().T + plain["qkv_b"][:Da]).double()
    k = (xe @ plain["qkv_w"][Da: 2 * Da].float().T).double()
    qs = (xe @ scaled["qkv_w"][:Da].float().T + scaled["qkv_b"][:Da]).double()
    ks = (xe @ scaled["qkv_w"][Da: 2 * Da].float().T).double()
    h = slice(0, 64)                                    # one head; model.py:118-121: (q * d^-.25) @ (k * d^-.25)^T, softmax
    ref_p = torch.softmax((q[:, h] * 64 ** -0.25) @ (k[:, h] * 64 ** -0.25).T, -1)
    e2 = torch.exp2(qs[:, h] @ ks[:, h].T)             # the kernel's form: p = 2^(k'.q'), normalised by the row sum
    assert torch.allclose(e2 / e2.sum(-1, keepdim=True), ref_p, atol=1e-6)

    # folded decoder block: stored gamma / beta become (1, 0)
    Dt = dims.n_text_state
    folded = dict((f, t) for f, t, _ in hip._block_pieces(sd, "decoder.blocks.0.", True, Dt, fold=True))
    for n in ("attn_ln", "cross_ln", "mlp_ln"):
        assert torch.all(folded[n + "_w"] == 1) and torch.all(folded[n + "_b"] == 0)

    # (3) conv as GEMM: rows of the GEMM input are 3 consecutive time steps of the zero-padded [time][channel] input
    C, T = 6, 11
    wc, xc = torch.randn(D, C, 3, generator=g), torch.randn(1, C, T, generator=g)
    wg = hip._conv_as_gemm(wc, 64)
    assert wg.shape == (D, 64) and torch.all(wg[:, 3 * C:] == 0)
    xt = torch.nn.functional.pad(xc[0].T, (0, 0, 1, 1))                                  # [T + 2][C]
    rows1 = torch.stack([xt[t: t + 3].reshape(-1) for t in range(T)])                    # stride 1 (conv1)
    assert torch.allclose(rows1 @ wg[:, : 3 * C].T, torch.nn.functional.conv1d(xc, wc, padding=1)[0].T, atol=1e-5)
    rows2 = torch.stack([xt[t: t + 3].reshape(-1) for t in range(0, T - 1, 2)])          # stride 2 (conv2): lda = 2 C
    assert torch.allclose(rows2 @ wg[:, : 3 * C].T, torch.nn.functional.conv1d(xc, wc, stride=2, padding=1)[0].T[: rows2.shape[0]], atol=1e-5)


def test_sampler_row_state_model_equals_timestamp_rules():
    """csrc/sampling.hip keeps three ints per row between steps (last sampled token is a timestamp, the one before is,
    last timestamp value + 1) instead of walking the sampled tokens; the partial kernel derives the masks of
    ApplyTimestampRules (decoding.py:441-505) from them and the decision kernel applies the "timestamp mass" rule to two
    range statistics.  Python model of exactly that, step by step against the host filter (itself checked against the
    reference in test_logit_filters_match_reference), greedy choice on random logits."""
    import types
    import torch
    from whisper_amd.decoding import ApplyTimestampRules
    V, TB, EOT, NOTS, T0, MAXI = 72, 40, 30, 35, 3, 5
    tk = types.SimpleNamespace(timestamp_begin=TB, eot=EOT, no_timestamps=NOTS)
    filt = ApplyTimestampRules(tk, T0, MAXI)
    g = torch.Generator().manual_seed(11)
    R, steps = 16, 40
    tokens = torch.randint(0, EOT, (R, T0), generator=g)
    state = [[0, 0, 0] for _ in range(R)]                       # row_state: last is ts, previous is ts, last ts value + 1
    NEG = float("-inf")
    for step in range(steps):
        # logits with a varying tilt towards timestamps so that every rule fires somewhere
        logits = torch.randn(R, V, generator=g) * 2.0
        logits[:, TB:] += torch.randn(R, 1, generator=g) * 2.0
        want = logits.clone()
        filt.apply(want, tokens)
        want_tok = want.argmax(-1)
        L = tokens.shape[1] - T0
        got = []
        for r in range(R):
            s0, s1, s2 = state[r]
            last_ts, pen_ts = (L >= 1) and s0 != 0, (L < 2) or s1 != 0
            lo = hi = 0
            if s2 > 0:
                lo, hi = TB, (s2 - 1) if (last_ts and not pen_ts) else s2
            x = logits[r].clone()
            for v in range(V):
                m = v == NOTS
                if last_ts:
                    m = m or (v >= TB if pen_ts else v < EOT)
                m = m or (lo <= v < hi)
                if L == 0:
                    m = m or v < TB or v > TB + MAXI
                if m:
                    x[v] = NEG
            # decision kernel: statistics of the text range [0, TB) and the timestamp range [TB, V)
            tx, ts = x[:TB], x[TB:]
            lse_all = torch.logsumexp(x, 0)
            ts_lse = torch.logsumexp(ts, 0) - lse_all if torch.isfinite(ts).any() else torch.tensor(NEG)
            text_best = tx.max() - lse_all if torch.isfinite(tx).any() else torch.tensor(NEG)
            nxt = int(ts.argmax()) + TB if ts_lse > text_best else int(x.argmax())
            got.append(nxt)
            is_ts = nxt >= TB
            state[r] = [1 if is_ts else 0, s0, (nxt + 1) if is_ts else s2]
        assert got == want_tok.tolist(), (step, got, want_tok.tolist())
        tokens = torch.cat([tokens, want_tok[:, None]], 1)
    sampled = tokens[:, T0:]
    assert (sampled >= TB).any() and (sampled < EOT).any()      # both kinds were produced


def test_beam_shared_history_permutation_model_is_exact():
    """The fused beam loop moves only part of the self-attention cache when the beams are re-ordered (csrc/beam.hip:
    lcp' / copy_from, csrc/elementwise.hip::permute_group_kernel): new row i takes from its source only the positions
    from lcp[i][src[i]] on, and lcp'[i][j] = lcp[src[i]][src[j]] (everything so far for equal sources).  Model with the
    cache content of (row, position) identified by the token history it was computed from: after every limited
    permutation each row must hold exactly what the full gather of decoding.py:172-176 would have given it."""
    rng = np.random.default_rng(5)
    moved = full = 0
    for G in (2, 5, 8):
        for trial in range(20):
            prompt = [int(t) for t in rng.integers(0, 9, size=int(rng.integers(1, 4)))]
            hist = [list(prompt) for _ in range(G)]                       # token history of each beam
            cache = [[tuple(prompt[: p + 1]) for p in range(len(prompt))] for _ in range(G)]
            lcp = [[0x7F7F7F7F] * 8 for _ in range(8)]                    # hipMemsetAsync(0x7f): "everything so far"
            for step in range(30):
                length = len(hist[0])                                     # cached positions == tokens in a row
                assert all(cache[i] == [tuple(hist[i][: p + 1]) for p in range(length)] for i in range(G))
                # the update: every new beam picks a source (sticky / collapsing / random phases) and a new token
                mode = rng.integers(0, 3)
                src = [i if mode == 0 and rng.random() < 0.8 else int(rng.integers(0, G if mode != 1 else min(G, 2)))
                       for i in range(G)]
                tok = [int(t) for t in rng.integers(0, 9, size=G)]
                new_lcp = [[0] * 8 for _ in range(8)]
                for i in range(G):
                    for j in range(G):
                        new_lcp[i][j] = min(length, length if src[i] == src[j] else lcp[src[i]][src[j]])
                copy_from = [min(length, length if src[i] == i else lcp[i][src[i]]) for i in range(G)]
                old = [list(c) for c in cache]
                for i in range(G):
                    if src[i] != i:
                        for p in range(copy_from[i], length):
                            cache[i][p] = old[src[i]][p]
                        moved += length - copy_from[i]
                    full += length if src[i] != i else 0
                hist = [hist[src[i]] + [tok[i]] for i in range(G)]
                # exactness: what the full gather would hold
                assert all(cache[i] == [tuple(hist[i][: p + 1]) for p in range(length)] for i in range(G)), (G, trial, step)
                # and the table keeps its meaning: rows i, j agree on their first lcp'[i][j] positions
                for i in range(G):
                    for j in range(G):
                        assert cache[i][: new_lcp[i][j]] == cache[j][: new_lcp[i][j]]
                lcp = new_lcp
                for i in range(G):                                        # the next decode step appends position `length`
                    cache[i].append(tuple(hist[i]))
    assert moved < full                                                   # and it does save copies


def test_coalesce_batches_and_job_seeds():
    """round 6 host logic of decode_many / the lanes (CPU): batches are coalesced into chains of at most chain_rows rows in order, a batch
    wider than the bound stays alone, None / 0 disables it; and the sampling seeds a job draws are a function of the base seed and the
    job's INDEX only (not of the order in which lanes run jobs), repeatable under torch.manual_seed, distinct between jobs."""
    from whisper_amd import decoding as D
    assert D.coalesce_batches([8, 8, 8], 24) == [[0, 1, 2]]
    assert D.coalesce_batches([8, 8, 8, 8, 8], 24) == [[0, 1, 2], [3, 4]]
    assert D.coalesce_batches([8, 16, 8, 1, 1], 24) == [[0, 1], [2, 3, 4]]
    assert D.coalesce_batches([40, 40, 8, 8], 24) == [[0], [1], [2, 3]]          # beam 5 x 8 clips: wider than a chain
    assert D.coalesce_batches([8, 8], None) == [[0], [1]] and D.coalesce_batches([8, 8], 0) == [[0], [1]]
    assert D.coalesce_batches([], 24) == []
    flat = [i for c in D.coalesce_batches([3, 1, 2, 3, 2, 24, 5, 19, 1], 24) for i in c]
    assert flat == list(range(9))

    def draws(order):
        torch.manual_seed(7)
        base = D._draw_seed()
        out = {}
        for i in order:                                   # jobs picked up in any order by the lanes
            with D._job_seeds(D._job_generator(base, i)):
                out[i] = (D._draw_seed(), D._draw_seed())
        return out
    a, b = draws([0, 1, 2, 3]), draws([3, 1, 0, 2])
    assert a == b
    assert len({v for pair in a.values() for v in pair}) == 8
    torch.manual_seed(8)
    assert D._draw_seed() != draws([0])[0][0]
    # outside a job the process-wide generator is used, as in the reference
    torch.manual_seed(3); x = D._draw_seed(); torch.manual_seed(3); assert D._draw_seed() == x


def test_greedy_decoder_update_returns_fresh_tensors_to_external_callers():
    """GreedyDecoder.update must not alias the tensor an earlier call returned (reference decoding.py:290 is a torch.cat);
    the in-place append is reserved for DecodingTask's own host loop."""
    from whisper_amd.decoding import GreedyDecoder
    dec = GreedyDecoder(0.0, eot=9)
    tokens = torch.tensor([[1, 2], [3, 4]])
    logits = torch.zeros(2, 10); logits[0, 5] = 1.0; logits[1, 6] = 1.0
    s0 = torch.zeros(2)
    t1, _ = dec.update(tokens, logits.clone(), s0)
    logits2 = torch.zeros(2, 10); logits2[0, 7] = 1.0; logits2[1, 8] = 1.0
    t2a, _ = dec.update(t1, logits2.clone(), s0)
    t2b, _ = dec.update(t1, logits.clone(), s0)           # branching from the same prefix again
    assert t1.tolist() == [[1, 2, 5], [3, 4, 6]]
    assert t2a.tolist() == [[1, 2, 5, 7], [3, 4, 6, 8]] and t2b.tolist() == [[1, 2, 5, 5], [3, 4, 6, 6]]
    assert t2a.data_ptr() != t2b.data_ptr() and t1.shape[1] == 3
