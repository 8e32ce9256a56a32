"""BASELINE.json's configurations at their OWN model dims, row counts and lengths, end to end on raw audio (VERDICT round 5,
"What's weak" 1): log-mel (HIP) -> AudioEncoder (HIP) -> cross K/V -> the fused greedy loop, against the CPU oracle doing the
same on the same samples.  The checkpoints are seeded random-init weights of the named architecture, margin-conditioned on the
test's own clips (oracle/condition.py `build_reference`: a trained model's peaked next-token distribution; on plain random
weights the top two logits tie to within rounding every few hundred steps and no reduced-precision engine can be id-exact).
All calls go through libwhisper_hip.so.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import condition
from whisper_amd import hip

pytestmark = pytest.mark.gpu


def _clips(n, seed0=0, samples=480000):
    """bench.py's synthetic clips (seeded noise + three tones)"""
    t = np.arange(samples) / 16000.0
    out = []
    for b in range(n):
        rng = np.random.default_rng(seed0 + b)
        x = rng.standard_normal(samples).astype(np.float32) * 0.05
        for f, a in ((220.0 + 20 * b, 0.2), (1300.0, 0.1), (3100.0, 0.05)):
            x += (a * np.sin(2 * np.pi * f * t)).astype(np.float32)
        out.append(x)
    return np.stack(out)


def _setup(dims, n_steps, device):
    from whisper_amd.tokenizer import get_tokenizer
    tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
    init = list(tok.sot_sequence)
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm,
                                                         tok.no_speech, tok.eot]))      # EOT suppressed: exactly n_steps tokens
    mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
    mask[suppress] = 1
    mask = mask.to(device)
    params = hip.GreedyParams(sample_begin=len(init), max_steps=n_steps, n_ctx=dims.n_text_ctx, eot=tok.eot,
                              timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                              max_initial_timestamp_index=50, suppress_blank=1, blank_token=tok.encode(" ")[0],
                              suppress_mask=mask.data_ptr())
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx,
                                 timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                                 suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)
    return tok, init, params, rules, mask


def test_base_one_clip_fp16_end_to_end_224_steps(gpu_device):
    """BASELINE configs[1]: base (6 + 6 layers, D = 512, 8 heads, 80 mels), ONE 30 s synthetic clip, greedy, fp16 — the shape whose
    decode step is 39 launches of one row (every projection on the 1-row path of the MFMA diagonal kernel, one key split per head
    in the attention launches).  Against the oracle on the same samples:
      * log-mel within 1e-4 (audio.py:110-157);
      * fp16 engine (what bench.py's `base_x1` leg times), its own log-mel and encoder in the loop: token ids of all 224 steps EXACT,
        sum_logprob within 0.15, no_speech_prob within 2e-3; encoder output within 2e-2 of the oracle's (measured 5e-3);
      * fp32 strict engine: the same ids; encoder within 2e-3; teacher-forced logits along the oracle's path (prompt pass + 12 steps,
        on the oracle's features) within 1e-3 — north_star's bar; the fp16 engine's error on the same path is measured, reported
        and bounded at about 2 x what was observed when the test was written (max 0.0087 / rms 0.0015)."""
    from conftest import write_report
    from whisper_amd.audio import log_mel_spectrogram
    from whisper_amd.synthetic import dims_for, synthetic_state_dict
    dims = dims_for("base")
    n_steps = 224
    sd = synthetic_state_dict(dims, seed=0, device="cpu")
    tok, init, params, rules, mask = _setup(dims, n_steps, gpu_device)
    T0 = len(init)
    audio = _clips(1)
    ref = condition.build_reference(dims, sd, audio, init, n_steps, rules, seed=2)
    want, mg = ref["dec"], ref["margins"]
    print("base x 1: oracle margins", mg, "edited rows", ref["edited_rows"])
    assert ref["consistent"] and mg["min"] >= 0.3 and mg["median"] >= 1.0 and mg.get("rule_min", 1.0) >= 0.3, mg
    wt = want["tokens"][0, T0:].tolist()
    assert len(wt) == n_steps and len(set(wt)) == n_steps and sum(t >= tok.timestamp_begin for t in wt) >= 4

    daudio = torch.from_numpy(audio).to(gpu_device)
    mel = log_mel_spectrogram(daudio, dims.n_mels)
    assert (mel.cpu() - ref["mel"]).abs().max().item() < 1e-4
    rep = {"model": "base dims, seed-0 weights, margin-conditioned on the clip", "rows": 1, "steps": n_steps, "oracle_margins": mg, "engines": {}}
    for dt, label, enc_tol in ((hip.WH_F16, "fp16", 2e-2), (hip.WH_F32, "fp32", 2e-3)):
        eng = hip.HipModel(dims, dt, hip.pack_weights(sd, dims, dt, gpu_device))
        try:
            feats = eng.encode(mel)
            enc_err = (feats.float().cpu() - ref["feats"]).abs().max().item()
            assert enc_err < enc_tol, (label, enc_err)
            task = hip.HipTask(eng, 1, 1, 8)
            try:
                task.set_audio(feats.contiguous())
                tokens = torch.zeros(1, T0 + n_steps + 1, dtype=torch.int64, device=gpu_device)
                tokens[:, :T0] = torch.tensor(init, device=gpu_device)
                n, sum_lp, nsp = task.greedy(tokens, params, 0, tok.no_speech)
                assert task.handoff_timeouts() == 0
            finally:
                task.close()
            got = tokens[0, T0:n].cpu().tolist()
            first = oracle.first_divergence(got, wt)
            lp_err = abs(float(sum_lp[0]) - want["sum_logprobs"][0])
            ns_err = abs(float(nsp[0]) - want["no_speech_probs"][0])
            # teacher-forced logits on the ORACLE's features (the decoder alone), prompt pass + 12 steps
            tf = hip.HipTask(eng, 1, 1, 8)
            worst, sq, cnt = 0.0, 0.0, 0
            try:
                tf.set_audio(ref["feats"].to(gpu_device, eng.torch_dtype).contiguous())
                toks = want["tokens"][:, : T0 + 12].to(gpu_device)
                outs = [tf.prefill(toks[:, :T0].contiguous())[:, -1].float().cpu()]
                for i in range(T0, T0 + 12):
                    outs.append(tf.step(toks[:, i].contiguous()).float().cpu())
            finally:
                tf.close()
            for i, g in enumerate(outs):
                w = want["step_logits"][i].float()
                ok = torch.isfinite(w)
                d = (g - w)[ok].abs()
                worst, sq, cnt = max(worst, float(d.max())), sq + float((d.double() ** 2).sum()), cnt + int(ok.sum())
            rms = (sq / cnt) ** 0.5
            rep["engines"][label] = {"first_divergence": first, "sum_logprob_err": lp_err, "no_speech_err": ns_err,
                                     "encoder_max_err": enc_err, "teacher_forced_max_abs_dlogit": worst, "teacher_forced_rms_dlogit": rms}
            print("base x 1", label, rep["engines"][label])
            assert n == T0 + n_steps
            assert first is None, (label, first, got[first], wt[first])            # token-id exact, all 224 steps
            assert lp_err < (2e-3 if dt == hip.WH_F32 else 0.15) and ns_err < 2e-3, (label, lp_err, ns_err)
            if dt == hip.WH_F32:
                assert worst < 1e-3, worst                                         # north_star: logits within 1e-3
            else:
                assert worst < 0.02 and rms < 3e-3, (worst, rms)
        finally:
            eng.drop_cached_tasks()
            del eng
            torch.cuda.empty_cache()
    write_report("conditioned_base_x1.json", rep)


def test_decode_many_coalesces_batches_into_one_chain(gpu_device):
    """whisper_amd.decode_many(chain_rows=24): three batches of 8 clips become ONE 24-row decode chain (one task, the decoder's
    weights streamed once per step) — the round-6 headline's schedule.  turbo-width micro model is not the point here: wide-v3 dims
    (D = 1280, 2 + 2 layers) so that the 17-24-row kernels (three row tiles per weight fragment) run.
      * fp32 engine: every clip's tokens / avg_logprob / no_speech_prob EXACTLY what decode() gives its own batch of 8;
      * fp16 engine: the same up to the order of fp32 partial sums — at least 22 of 24 clips identical over 24 steps on these
        random-init weights, avg_logprob within 2e-2 where the ids agree;
      * raw-audio batches take their log-mel per batch (its clamp is a maximum over the tensor given, audio.py:155), so coalescing
        does not change a spectrogram;
      * chain_rows=None decodes batch by batch (three 8-row chains, in_flight of them at once, one host thread): fp32 exact, fp16 as above."""
    import whisper_amd
    from whisper_amd.model import ModelDimensions, Whisper
    from oracle.model import dims_dict
    dims = oracle.dims_for("wide-v3")
    sd = oracle.synthetic_state_dict(dims, seed=3)
    model = Whisper(ModelDimensions(**dims_dict(dims)), sd, device=gpu_device)
    audio = torch.from_numpy(_clips(24, seed0=100)).to(gpu_device)
    batches = [audio[0:8], audio[8:16], audio[16:24]]
    for fp16 in (False, True):
        opts = whisper_amd.DecodingOptions(language="en", fp16=fp16, sample_len=24)
        want = [whisper_amd.decode(model, whisper_amd.log_mel_spectrogram(b, dims.n_mels), opts) for b in batches]
        eng = model.engine(torch.float16 if fp16 else torch.float32)
        eng.drop_cached_tasks()
        got = whisper_amd.decode_many(model, batches, opts, in_flight=1, chain_rows=24)
        shapes = sorted({t.n_rows for t in eng._task_cache})
        assert shapes == [24], shapes                                   # one chain of 24 rows did it
        assert [len(g) for g in got] == [8, 8, 8]
        same = sum(g.tokens == w.tokens for gs, ws in zip(got, want) for g, w in zip(gs, ws))
        if not fp16:
            assert same == 24
            for gs, ws in zip(got, want):
                assert np.allclose([g.avg_logprob for g in gs], [w.avg_logprob for w in ws], atol=1e-5)
                assert np.allclose([g.no_speech_prob for g in gs], [w.no_speech_prob for w in ws], atol=1e-6)
        else:
            assert same >= 22, same
            for gs, ws in zip(got, want):
                for g, w in zip(gs, ws):
                    if g.tokens == w.tokens:
                        assert abs(g.avg_logprob - w.avg_logprob) < 2e-2
        eng.drop_cached_tasks()
        lanes = whisper_amd.decode_many(model, batches, opts, in_flight=3, chain_rows=None)
        assert sorted({t.n_rows for t in eng._task_cache}) == [8]
        same = sum(g.tokens == w.tokens for gs, ws in zip(lanes, want) for g, w in zip(gs, ws))
        # same task shapes; in the fp16 engine a lane's task runs its cross attention as two launches (no spinning kernel beside other
        # chains) where decode() alone uses the fused launch: the fp32 partial sums meet in another order there
        assert same == 24 if not fp16 else same >= 22, same
        eng.drop_cached_tasks()


def test_ragged_prompts_in_a_20_row_chain(gpu_device):
    """Rows of ONE decode chain with previous-text prompts of different lengths (wh_task_set_lag) at the row counts of round 6's chains:
    20 rows at large-v3 widths (2 + 2 layers) — the row-tiled projections' cache append at `*d_pos - lag[r]`, the per-row self-attention
    lengths, the sampler's per-row sample_begin, with three row tiles per weight fragment.  tests/test_api_gpu.py covers this at 6 rows
    (the <= 8-row kernels) only; transcribe_batch(batch_size = 16 .. 24, condition_on_previous_text=True) runs exactly this.
    Every row against the same segment decoded ALONE with options.prompt: fp32 engine ids exact, avg_logprob to 1e-4; fp16 engine at
    least 18 of 20 rows identical over 16 steps on these random-init weights (a batched and a single-row decode sum in different
    orders), avg_logprob within 2e-2 where the ids agree."""
    import whisper_amd
    from whisper_amd.model import ModelDimensions, Whisper
    from oracle.model import dims_dict
    dims = oracle.dims_for("wide-v3")
    sd = oracle.synthetic_state_dict(dims, seed=3)
    model = Whisper(ModelDimensions(**dims_dict(dims)), sd, device=gpu_device)
    R = 20
    rng = np.random.default_rng(11)
    lengths = [0, 1, 5, 17, 40, 3, 0, 9, 60, 2] * 2
    prompts = [None if n == 0 else rng.integers(300, 40000, n).tolist() for n in lengths]
    audio = torch.from_numpy(_clips(R, seed0=300)).to(gpu_device)
    mel = whisper_amd.log_mel_spectrogram(audio, dims.n_mels)
    for fp16 in (False, True):
        opts = whisper_amd.DecodingOptions(language="en", fp16=fp16, sample_len=16)
        got = whisper_amd.decode(model, mel, opts, prompts=prompts)
        same = 0
        for i, p in enumerate(prompts):
            want = whisper_amd.decode(model, mel[i], opts, prompt=p)
            if got[i].tokens == want.tokens:
                same += 1
                assert abs(got[i].avg_logprob - want.avg_logprob) < (2e-2 if fp16 else 1e-4), (fp16, i)
            else:
                assert fp16, (i, got[i].tokens, want.tokens)          # the fp32 engine is exact
        assert same >= (18 if fp16 else R), (fp16, same)
