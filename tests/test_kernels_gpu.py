"""GPU parity tests of the individual HIP kernels / C-ABI entry points against the CPU oracle.
All calls go through libwhisper_hip.so (whisper_amd.hip -> ctypes).  Tolerances are stated per test."""
import numpy as np
import pytest
import torch

import oracle
from whisper_amd import hip

pytestmark = pytest.mark.gpu


def _audio(seed, n=480000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n).astype(np.float32) * 0.05
    x += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1870 * t)).astype(np.float32)
    return x


@pytest.fixture(scope="module")
def micro(gpu_device):
    out = {}
    for name in ("micro.en", "micro-v3"):
        dims = oracle.dims_for(name)
        sd = oracle.synthetic_state_dict(dims, seed=1)
        om = oracle.OracleModel(dims, sd)
        models = {}
        for dt in (hip.WH_F32, hip.WH_F16):
            blob = hip.pack_weights(sd, dims, dt, gpu_device)
            models[dt] = hip.HipModel(dims, dt, blob)
        out[name] = (dims, sd, om, models)
    return out


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("shape", [(1, 480000), (2, 160000), (1, 16000 * 7 + 80), (3, 640), (1, 201), (1, 16000 * 95 + 7)])
def test_log_mel(gpu_device, n_mels, shape):
    """fp32; atol 1e-4 (restatement-vs-torch.stft noise is 5.8e-5).  Sizes: one 30 s window, ragged
    frame counts (1000 and 700 frames against 16 frames per workgroup), 4 frames, the shortest input reflect padding
    accepts (201 samples: one frame, every tap reflected), and a 95 s file (9500 frames, global maximum over 594 workgroups)."""
    B, n = shape
    a = np.stack([_audio(10 + b, n) for b in range(B)])
    filt = oracle.mel_filterbank(n_mels)
    want = oracle.log_mel_spectrogram(a, filt, dtype=torch.float64)
    got = hip.log_mel(torch.from_numpy(a).to(gpu_device), torch.from_numpy(filt).to(gpu_device)).cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < 1e-4, err
    assert got.max() - got.min() <= 2.0 + 1e-6          # tests/test_audio.py:19 invariant of the reference


@pytest.mark.parametrize("name", ["micro.en", "micro-v3"])
@pytest.mark.parametrize("dt,tol", [(hip.WH_F32, 2e-4), (hip.WH_F16, 3e-2)])
def test_encoder(micro, gpu_device, name, dt, tol):
    dims, sd, om, models = micro[name]
    filt = oracle.mel_filterbank(dims.n_mels)
    mel = torch.stack([oracle.log_mel_spectrogram(_audio(3 + b), filt) for b in range(2)])
    want = om.encoder(mel)
    got = models[dt].encode(mel.to(gpu_device)).float().cpu()
    err = (got - want).abs().max().item()
    assert torch.isfinite(got).all()
    assert err < tol, err


def _feats(om, dims, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g)


@pytest.mark.parametrize("name", ["micro.en", "micro-v3"])
@pytest.mark.parametrize("dt,tol", [(hip.WH_F32, 1e-3), (hip.WH_F16, 6e-2)])   # fp16 engine vs the fp32 oracle: activations, K/V and LayerNorm-folded weights are rounded to fp16
@pytest.mark.parametrize("B,G,T0", [(2, 1, 5), (1, 3, 17), (3, 1, 1), (1, 1, 200), (16, 1, 2), (4, 10, 3)])
def test_prefill_and_steps(micro, gpu_device, name, dt, tol, B, G, T0):
    """Teacher-forced logits at every position: prefill (GEMM path) then 6 single-token steps (GEMV path,
    the 2nd onwards replayed from the hipGraph) vs the oracle's KV-cache decoder.  fp32: |dlogit| < 1e-3."""
    dims, sd, om, models = micro[name]
    model = models[dt]
    R = B * G
    feats = _feats(om, dims, B, seed=B * 7 + G)
    g = torch.Generator().manual_seed(5)
    toks = torch.randint(0, dims.n_vocab, (R, T0 + 6), generator=g)
    cache = om.new_cache()
    want0 = om.decoder(toks[:, :T0], feats, cache)
    task = hip.HipTask(model, B, G, max(T0, 8))
    try:
        task.set_audio(feats.to(gpu_device, model.torch_dtype).contiguous())
        dtoks = toks.to(gpu_device)
        got0 = task.prefill(dtoks[:, :T0].contiguous()).cpu()
        d0 = (got0 - want0).abs()
        if dt == hip.WH_F16 and T0 >= 100:
            # 200 positions x 51864 logits = 10^7 samples: the maximum of that many fp16-level errors sits higher than
            # over the 10^5..10^6 samples of the other cases, so the long prompt is bounded by max AND rms
            assert d0.max().item() < 1e-1 and d0.pow(2).mean().sqrt().item() < 1e-2, (d0.max().item(), d0.pow(2).mean().sqrt().item())
        else:
            assert d0.max().item() < tol, d0.max().item()
        for i in range(6):
            want = om.decoder(toks[:, T0 + i: T0 + i + 1], feats, cache)[:, -1]
            got = task.step(dtoks[:, T0 + i]).cpu()
            err = (got - want).abs().max().item()
            assert err < tol, (i, err)
        assert task.position == T0 + 6
        # selected positions only
        task.reset()
        sel = [0, T0 - 1]
        got_sel = task.prefill(dtoks[:, :T0].contiguous(), sel=sel).cpu()
        assert (got_sel - want0[:, sel]).abs().max().item() < tol
    finally:
        task.close()


def test_rearrange(micro, gpu_device):
    """rearrange_kv_cache (decoding.py:172-176): permuting rows of the self-attention cache == permuting rows."""
    dims, sd, om, models = micro["micro.en"]
    model = models[hip.WH_F32]
    B, G, T0 = 1, 4, 6
    feats = _feats(om, dims, B)
    g = torch.Generator().manual_seed(9)
    toks = torch.randint(0, dims.n_vocab, (G, T0 + 1), generator=g)
    src = [2, 0, 0, 3]
    cache = om.new_cache()
    om.decoder(toks[:, :T0], feats, cache)
    om.rearrange(cache, src)
    nxt = toks[:, T0:T0 + 1]
    want = om.decoder(nxt, feats, cache)[:, -1]
    task = hip.HipTask(model, B, G, 8)
    try:
        task.set_audio(feats.to(gpu_device).contiguous())
        task.prefill(toks[:, :T0].contiguous().to(gpu_device), sel=[T0 - 1])
        task.rearrange(src)
        got = task.step(nxt[:, 0].to(gpu_device)).cpu()
        assert (got - want).abs().max().item() < 1e-3
    finally:
        task.close()


def _rules(dims, T0, with_ts=True):
    multilingual = dims.n_vocab >= 51865
    eot = 50257 if multilingual else 50256
    nl = dims.n_vocab - 51765 - int(multilingual)
    sot = eot + 1
    transcribe = sot + 1 + nl + 1
    no_speech = transcribe + 3
    no_ts = no_speech + 1
    rng = np.random.default_rng(0)
    suppress = sorted(set(rng.integers(0, 50000, 80).tolist() + [sot, transcribe, transcribe - 1, no_speech]))
    return oracle.SamplingRules(sample_begin=T0, sot_index=0, eot=eot, n_ctx=dims.n_text_ctx,
                                timestamp_begin=(no_ts + 1) if with_ts else None, no_timestamps=no_ts,
                                max_initial_timestamp_index=50, suppress_blank=True, blank_token=220,
                                suppress_tokens=suppress, no_speech=no_speech)


@pytest.mark.parametrize("name", ["micro.en", "micro-v3"])
@pytest.mark.parametrize("with_ts", [True, False])
def test_fused_greedy(micro, gpu_device, name, with_ts):
    """wh_task_greedy (device-side filters + argmax) vs the oracle's row-wise loop: token ids exact,
    sum_logprobs within 1e-3, fp32 strict mode; 40 steps, 3 rows."""
    dims, sd, om, models = micro[name]
    model = models[hip.WH_F32]
    B, n_steps = 3, 40
    init = [50258 if dims.n_vocab >= 51865 else 50257]
    if dims.n_vocab >= 51865:
        init = [50258, 50259, 50258 + 1 + (dims.n_vocab - 51765 - 1) + 1]
    if not with_ts:
        init = init + [_rules(dims, 1).no_timestamps]
    T0 = len(init)
    rules = _rules(dims, T0, with_ts)
    feats = _feats(om, dims, B, seed=11)
    want = oracle.greedy_decode(om, feats, init, n_steps, rules)
    task = hip.HipTask(model, B, 1, 8)
    try:
        task.set_audio(feats.to(gpu_device).contiguous())
        tokens = torch.zeros(B, T0 + n_steps + 1, dtype=torch.int64, device=gpu_device)
        tokens[:, :T0] = torch.tensor(init)
        mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
        mask[rules.suppress_tokens] = 1
        mask = mask.to(gpu_device)
        p = hip.GreedyParams(sample_begin=T0, max_steps=n_steps, n_ctx=dims.n_text_ctx, eot=rules.eot,
                             timestamp_begin=rules.timestamp_begin if with_ts else -1,
                             no_timestamps=rules.no_timestamps, max_initial_timestamp_index=50,
                             suppress_blank=1, blank_token=220, suppress_mask=mask.data_ptr())
        n, sum_lp, nsp = task.greedy(tokens, p, 0, rules.no_speech)
        torch.cuda.synchronize()
        wt = want["tokens"]
        assert n == wt.shape[1], (n, wt.shape)
        assert torch.equal(tokens[:, :n].cpu(), wt)
        assert np.allclose(sum_lp.cpu().numpy(), np.array(want["sum_logprobs"]), atol=2e-3)
        assert np.allclose(nsp.cpu().numpy(), np.array(want["no_speech_probs"]), rtol=1e-3, atol=1e-7)
    finally:
        task.close()


def test_fused_greedy_to_the_end_of_the_context(micro, gpu_device):
    """Maximum size: EOT suppressed and sample_len = n_text_ctx - prompt, so the loop runs until the token row is full
    (decoding.py:699-705 stops at `tokens.shape[-1] > n_ctx`).  fp32 engine vs the oracle: the same number of tokens and
    the same ids over all 445 steps, 2 rows; the fp16 engine (fused step kernels: the last cache slot, the last key round)
    runs the same loop to the same length without a hand-off timeout, starts like the fp32 engine and samples allowed ids only."""
    dims, sd, om, models = micro["micro-v3"]
    B = 2
    init = [50258, 50259, 50258 + 1 + (dims.n_vocab - 51765 - 1) + 1]
    T0 = len(init)
    n_steps = dims.n_text_ctx - T0
    r0 = _rules(dims, T0, True)
    rules = oracle.SamplingRules(sample_begin=T0, sot_index=0, eot=r0.eot, n_ctx=dims.n_text_ctx, timestamp_begin=r0.timestamp_begin,
                                 no_timestamps=r0.no_timestamps, max_initial_timestamp_index=50, suppress_blank=True,
                                 blank_token=220, suppress_tokens=sorted(set(list(r0.suppress_tokens) + [r0.eot])),
                                 no_speech=r0.no_speech)
    feats = _feats(om, dims, B, seed=23)
    want = oracle.greedy_decode(om, feats, init, n_steps, rules)
    assert want["tokens"].shape[1] == dims.n_text_ctx
    mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
    mask[rules.suppress_tokens] = 1
    mask = mask.to(gpu_device)
    got = {}
    for dt in (hip.WH_F32, hip.WH_F16):
        task = hip.HipTask(models[dt], B, 1, 8)
        try:
            task.set_audio(feats.to(gpu_device, torch.float16 if dt == hip.WH_F16 else torch.float32).contiguous())
            tokens = torch.zeros(B, T0 + n_steps + 1, dtype=torch.int64, device=gpu_device)
            tokens[:, :T0] = torch.tensor(init)
            p = hip.GreedyParams(sample_begin=T0, max_steps=n_steps, n_ctx=dims.n_text_ctx, eot=rules.eot,
                                 timestamp_begin=rules.timestamp_begin, no_timestamps=rules.no_timestamps,
                                 max_initial_timestamp_index=50, suppress_blank=1, blank_token=220, suppress_mask=mask.data_ptr())
            n, sum_lp, nsp = task.greedy(tokens, p, 0, rules.no_speech)
            torch.cuda.synchronize()
            assert n == dims.n_text_ctx, n
            assert task.handoff_timeouts() == 0
            got[dt] = tokens[:, :n].cpu()
            # the cache holds n - 1 positions (the last sampled token has not been fed): one more step fills the last slot,
            # the one after it is refused, not written out of bounds
            assert task.position == dims.n_text_ctx - 1
            assert torch.isfinite(task.step(tokens[:, n - 1].contiguous())).all()
            with pytest.raises(hip.HipError):
                task.step(tokens[:, n - 1].contiguous())
        finally:
            task.close()
    assert torch.equal(got[hip.WH_F32], want["tokens"])
    # random weights: the logits are nearly flat, so the fp16 engine leaves the fp32 path at its first flipped near-tie
    # (parity of the fp16 engine is tests/test_wide_gpu.py's subject); here: same start, and every sampled id is allowed
    g16 = got[hip.WH_F16]
    assert torch.equal(g16[:, :T0 + 4], got[hip.WH_F32][:, :T0 + 4])
    assert int(g16.min()) >= 0 and int(g16.max()) < dims.n_vocab
    assert not bool(torch.isin(g16[:, T0:], torch.tensor(rules.suppress_tokens)).any())


@pytest.mark.parametrize("shape", [(10,), (1, 15), (4, 5, 345), (6, 12, 240, 512)])
def test_median_filter(gpu_device, shape):
    """shapes and widths of the reference's tests/test_timing.py:14-19,67-84; exact (order statistics)."""
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(0))
    for w in [3, 5, 7, 13]:
        want = oracle.median_filter(x.numpy(), w)
        got = hip.median_filter(x.to(gpu_device), w).cpu().numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("N,M", [(10, 20), (32, 16), (123, 1500), (234, 189)])
def test_dtw(gpu_device, N, M):
    """sizes of tests/test_timing.py:8-13; trace equal to dtw_cpu's on random input (integer codes: exact)."""
    x = np.random.default_rng(N * M).standard_normal((N, M)).astype(np.float32)
    want = oracle.dtw_trace(x)
    got = hip.dtw_trace(torch.from_numpy(x).to(gpu_device)).cpu().numpy()
    assert np.array_equal(got[1:, 1:], want[1:, 1:])
    assert np.array_equal(oracle.backtrace(got), oracle.backtrace(want))


@pytest.mark.parametrize("N,M", [(1, 1), (10, 20), (32, 16), (123, 1500), (234, 189), (447, 1500), (700, 1500)])
def test_dtw_backtrace_on_device(gpu_device, N, M):
    """wh_dtw_backtrace_batch (timing.py:57-79 on the device, a lane per clip on the 2-bit LDS copy of the trace; the
    700 x 1500 trace does not fit the LDS and is walked in global memory): the path equals the oracle's `dtw_path`
    entry by entry, the jump frames equal `time_indices[jumps]` (timing.py:226-228); also the planted monotone path of
    the reference's tests/test_timing.py:20-46 generator, and `whisper_amd.timing.dtw` end to end."""
    from whisper_amd.timing import dtw
    rng = np.random.default_rng(N * 31 + M)
    x = rng.standard_normal((N, M)).astype(np.float32)
    want = oracle.dtw_path(x)
    trace = hip.dtw_trace(torch.from_numpy(x).to(gpu_device))
    jumps, path = hip.dtw_backtrace(trace)
    assert np.array_equal(path.cpu().numpy(), want)
    ti, fi = want
    first = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
    assert np.array_equal(jumps.cpu().numpy(), fi[first])
    assert np.array_equal(dtw(torch.from_numpy(x).to(gpu_device)), want)
    # the reference's known-answer generator (tests/test_timing.py:22-48): random costs in [0, 1), minus 1 along a planted
    # monotone walk (a diagonal step wherever the walk turns a corner): dtw must recover exactly that walk
    steps = np.concatenate([np.zeros(N - 1), np.ones(M - 1)])
    rng.shuffle(steps)
    y = rng.random((N, M)).astype(np.float32)
    i, j, k, planted = 0, 0, 0, []
    while True:
        y[i, j] -= 1
        planted.append((i, j))
        if k == len(steps):
            break
        if k + 1 < len(steps) and steps[k] != steps[k + 1]:
            i, j, k = i + 1, j + 1, k + 2
            continue
        if steps[k] == 0:
            i += 1
        if steps[k] == 1:
            j += 1
        k += 1
    got = dtw(torch.from_numpy(y).to(gpu_device))
    assert np.array_equal(got, oracle.dtw_path(y))
    assert np.array_equal(got, np.array(planted).T)


def test_dtw_backtrace_batch_ragged(gpu_device):
    """several clips of different sizes in one launch, incl. a clip whose trace holds an invalid code (length -1)"""
    import ctypes as C
    rng = np.random.default_rng(5)
    sizes = [(12, 40), (57, 211), (3, 1500), (230, 1500)]
    Nm, Mm = max(n for n, _ in sizes), max(m for _, m in sizes)
    stride = (Nm + 1) * (Mm + 1)
    traces = torch.zeros(len(sizes) + 1, stride, dtype=torch.int8)
    wants = []
    for b, (n, m) in enumerate(sizes):
        x = rng.standard_normal((n, m)).astype(np.float32)
        tr = oracle.dtw_trace(x)
        traces[b, : tr.size] = torch.from_numpy(tr.reshape(-1))
        wants.append(oracle.backtrace(tr))
    bad = np.full((5, 7), 2, np.int8); bad[:, 0] = 1; bad[3, 4] = -1; bad[4, 6] = 0; bad[4, 5] = 0
    bad[4, 6] = 1                                                                # (4,6) up -> (3,6) left ... -> (3,4): invalid
    traces[len(sizes), : bad.size] = torch.from_numpy(bad.reshape(-1))
    rows = torch.tensor([n for n, _ in sizes] + [4], dtype=torch.int32).to(gpu_device)
    cols = torch.tensor([m for _, m in sizes] + [6], dtype=torch.int32).to(gpu_device)
    d_tr = traces.to(gpu_device)
    R = len(sizes) + 1
    jumps = torch.zeros(R, Nm, dtype=torch.int32, device=gpu_device)
    path = torch.zeros(R, 2, Nm + Mm, dtype=torch.int32, device=gpu_device)
    plen = torch.zeros(R, dtype=torch.int32, device=gpu_device)
    s = torch.cuda.current_stream(gpu_device)
    hip.check(hip.lib().wh_dtw_backtrace_batch(d_tr.data_ptr(), stride, rows.data_ptr(), cols.data_ptr(), R, Nm, Mm,
                                               jumps.data_ptr(), Nm, path.data_ptr(), Nm + Mm, plen.data_ptr(), hip.stream_ptr(s)))
    plen_h, path_h, jumps_h = plen.cpu().tolist(), path.cpu().numpy(), jumps.cpu().numpy()
    for b, want in enumerate(wants):
        assert plen_h[b] == want.shape[1]
        assert np.array_equal(path_h[b][:, Nm + Mm - plen_h[b]:], want)
        first = np.pad(np.diff(want[0]), (1, 0), constant_values=1).astype(bool)
        assert np.array_equal(jumps_h[b, : sizes[b][0]], want[1][first])
    assert plen_h[-1] == -1                                                      # the reference raises ValueError there


@pytest.mark.parametrize("n_steps", [1, 2, 7, 8, 9, 17])
def test_fused_greedy_step_count_edges(micro, gpu_device, n_steps):
    """the device loop polls for completion every 8 tokens: step counts around that period and the degenerate
    single-step call must give the oracle's ids and lengths (fp32 strict mode, EOT allowed)"""
    dims, sd, om, models = micro["micro.en"]
    model = models[hip.WH_F32]
    B = 2
    init = [50257]
    rules = _rules(dims, 1, True)
    feats = _feats(om, dims, B, seed=31)
    want = oracle.greedy_decode(om, feats, init, n_steps, rules)
    task = hip.HipTask(model, B, 1, 8)
    try:
        task.set_audio(feats.to(gpu_device).contiguous())
        tokens = torch.zeros(B, 1 + n_steps + 1, dtype=torch.int64, device=gpu_device)
        tokens[:, 0] = init[0]
        mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
        mask[rules.suppress_tokens] = 1
        mask = mask.to(gpu_device)
        p = hip.GreedyParams(sample_begin=1, max_steps=n_steps, n_ctx=dims.n_text_ctx, eot=rules.eot,
                             timestamp_begin=rules.timestamp_begin, no_timestamps=rules.no_timestamps,
                             max_initial_timestamp_index=50, suppress_blank=1, blank_token=220,
                             suppress_mask=mask.data_ptr())
        n, sum_lp, nsp = task.greedy(tokens, p, 0, rules.no_speech)
        wt = want["tokens"]
        assert n == wt.shape[1], (n, wt.shape)
        assert torch.equal(tokens[:, :n].cpu(), wt)
        assert np.allclose(sum_lp.cpu().numpy(), np.array(want["sum_logprobs"]), atol=2e-3)
    finally:
        task.close()


def test_task_handle_refuses_concurrent_callers(micro, gpu_device):
    """handles are not thread-safe (include/whisper_hip.h); this is enforced: while one thread is inside a call on a
    task (here: a long fused greedy decode), a second thread calling into the SAME task gets 'invalid call sequence'
    instead of corrupting it; afterwards the task works as before"""
    import threading
    dims, sd, om, models = micro["micro.en"]
    model = models[hip.WH_F16]
    from whisper_amd.tokenizer import get_tokenizer
    tok = get_tokenizer(False)
    init = list(tok.sot_sequence)
    task = hip.HipTask(model, 2, 1, 8)
    try:
        task.set_audio(_feats(om, dims, 2).to(gpu_device, torch.float16).contiguous())
        mask = torch.zeros(dims.n_vocab, dtype=torch.uint8, device=gpu_device)
        mask[tok.eot] = 1
        params = hip.GreedyParams(sample_begin=len(init), max_steps=400, n_ctx=dims.n_text_ctx, eot=tok.eot,
                                  timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                                  max_initial_timestamp_index=50, suppress_blank=1, blank_token=tok.encode(" ")[0],
                                  suppress_mask=mask.data_ptr())
        tokens = torch.zeros(2, len(init) + 401, dtype=torch.int64, device=gpu_device)
        tokens[:, :len(init)] = torch.tensor(init, device=gpu_device)
        seen = []

        def intruder():
            import time
            t_end = time.time() + 20
            while hip.lib().wh_task_position(task.handle) <= 0 and time.time() < t_end:      # unguarded query: wait until
                pass                                                                         # the decode is inside its call
            while "n" not in done and time.time() < t_end:
                rc = hip.lib().wh_task_reset(task.handle, None)
                if rc != 0:
                    seen.append(rc)
                    return
        th = threading.Thread(target=intruder)
        done = {}

        def decode():
            done["n"] = task.greedy(tokens, params, 0, -1)[0]
        td = threading.Thread(target=decode)
        td.start(); th.start(); td.join(); th.join()
        assert seen == [4], seen                                   # WH_ERR_STATE while the decode held the handle
        assert done["n"] == len(init) + 400
        task.reset()
        task.set_audio(_feats(om, dims, 2).to(gpu_device, torch.float16).contiguous())
        assert task.prefill(tokens[:, :len(init)].contiguous()).shape == (2, len(init), dims.n_vocab)
    finally:
        task.close()


# ---------------------------------------------------------------------------------------------------
# the fused loops without a blocked caller: wh_task_greedy_begin / wh_task_beam_begin + wh_task_poll
# ---------------------------------------------------------------------------------------------------
def _loop_setup(micro, gpu_device, dt, B, n_steps, eot_ok=True):
    dims, sd, om, models = micro["micro.en"]
    model = models[dt]
    rules = _rules(dims, 1, True)
    mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
    mask[list(rules.suppress_tokens) + ([] if eot_ok else [rules.eot])] = 1
    mask = mask.to(gpu_device)
    p = hip.GreedyParams(sample_begin=1, max_steps=n_steps, n_ctx=dims.n_text_ctx, eot=rules.eot,
                         timestamp_begin=rules.timestamp_begin, no_timestamps=rules.no_timestamps,
                         max_initial_timestamp_index=50, suppress_blank=1, blank_token=220, suppress_mask=mask.data_ptr())
    return dims, om, model, rules, p, mask


@pytest.mark.parametrize("dt", [hip.WH_F32, hip.WH_F16])
@pytest.mark.parametrize("n_steps", [1, 2, 9, 40])
def test_greedy_begin_poll_equals_blocking_call(micro, gpu_device, dt, n_steps):
    """wh_task_greedy_begin + wh_task_poll (the loop of decoding.py:680-710 split where the host would wait; nothing in them
    waits for the device) against wh_task_greedy on the same inputs: token ids, sum_logprobs and no-speech probabilities
    BIT-identical, same token count (both run one state machine, the blocking call with sleeping waits in place of the event
    queries).  While the loop runs the task refuses every other call (WH_ERR_STATE) and takes them again afterwards; a poll
    without a loop is refused too."""
    import time
    dims, om, model, rules, p, mask = _loop_setup(micro, gpu_device, dt, 3, n_steps)
    B = 3
    feats = _feats(om, dims, B, seed=31).to(gpu_device, model.torch_dtype).contiguous()

    def fresh():
        t = torch.zeros(B, 1 + n_steps + 1, dtype=torch.int64, device=gpu_device)
        t[:, 0] = 50257
        return t
    task = hip.HipTask(model, B, 1, 8)
    try:
        task.set_audio(feats)
        want_tok = fresh()
        n0, lp0, ns0 = task.greedy(want_tok, p, 0, rules.no_speech)
        assert hip.lib().wh_task_poll(task.handle, None) == 4                  # no loop begun: WH_ERR_STATE
        task.reset()
        got_tok = fresh()
        pend = task.greedy_begin(got_tok, p, 0, rules.no_speech)
        polls, refused = 0, hip.lib().wh_task_reset(task.handle, None)
        t_end = time.time() + 30
        while True:
            res = pend.poll()
            polls += 1
            if res is not None:
                break
            assert time.time() < t_end, "loop never ended"
        assert refused == 4                                                     # the task was busy with its loop
        n1, lp1, ns1 = res
        torch.cuda.synchronize()
        assert n1 == n0 and torch.equal(got_tok, want_tok)
        assert torch.equal(lp1, lp0) and torch.equal(ns1, ns0)
        assert pend.poll() is res                                               # idempotent once ended
        task.reset()                                                            # ... and the task takes calls again
        assert task.prefill(got_tok[:, :1].contiguous()).shape == (B, 1, dims.n_vocab)
    finally:
        task.close()


def test_two_loops_polled_from_one_thread(micro, gpu_device):
    """ONE host thread keeps two tasks on two streams going by polling them in turn (what whisper_amd.run_interleaved does):
    each task's 120-step result equals what it returns alone through the blocking call, and at some point both loops were
    running at once (neither had ended when the other was polled)."""
    dims, om, model, rules, p, mask = _loop_setup(micro, gpu_device, hip.WH_F16, 2, 120, eot_ok=False)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(2)]
    feats = [_feats(om, dims, 2, seed=40 + i).to(gpu_device, torch.float16).contiguous() for i in range(2)]
    torch.cuda.synchronize()
    tasks = [hip.HipTask(model, 2, 1, 8, stream=streams[i]) for i in range(2)]

    def fresh():
        t = torch.zeros(2, 1 + 120 + 1, dtype=torch.int64, device=gpu_device)
        t[:, 0] = 50257
        return t
    try:
        want = []
        for i in range(2):
            tasks[i].set_audio(feats[i])
            tk = fresh()
            n, lp, _ = tasks[i].greedy(tk, p, 0, -1)
            want.append((n, tk.clone(), lp.clone()))
            tasks[i].reset()
        toks = [fresh(), fresh()]
        pend = [tasks[i].greedy_begin(toks[i], p, 0, -1) for i in range(2)]
        res, both_running = [None, None], 0
        while res[0] is None or res[1] is None:
            for i in range(2):
                if res[i] is None:
                    res[i] = pend[i].poll()
            both_running += int(res[0] is None and res[1] is None)
        torch.cuda.synchronize()
        assert both_running > 0
        for i in range(2):
            assert res[i][0] == want[i][0] == 121
            assert torch.equal(toks[i], want[i][1]) and torch.equal(res[i][1], want[i][2])
    finally:
        for t in tasks:
            t.close()


@pytest.mark.parametrize("n_steps", [3, 20])
def test_beam_begin_poll_equals_blocking_call(micro, gpu_device, n_steps):
    """wh_task_beam_begin + wh_task_poll against wh_task_beam: live beams, sum_logprobs, finished lists (tokens, lengths, scores,
    counts) bit-identical; 2 clips x beam 3, fp32 engine."""
    dims, om, model, rules, p, mask = _loop_setup(micro, gpu_device, hip.WH_F32, 2, n_steps)
    B, G = 2, 3
    feats = _feats(om, dims, B, seed=33).to(gpu_device).contiguous()
    bp = hip.BeamParams(rules=p, beam_size=G, max_candidates=G)

    def fresh():
        t = torch.zeros(2, B * G, 1 + n_steps + 2, dtype=torch.int64, device=gpu_device)
        t[0, :, 0] = 50257
        return t
    task = hip.HipTask(model, B, G, 8)
    try:
        task.set_audio(feats)
        t0 = fresh()
        n0, lp0, ns0, fin0 = task.beam(t0, bp, 0, rules.no_speech)
        task.reset()
        t1 = fresh()
        n1, lp1, ns1, fin1 = task.beam_begin(t1, bp, 0, rules.no_speech).wait()
        torch.cuda.synchronize()
        assert n1 == n0 and torch.equal(t1[0, :, :n1], t0[0, :, :n0]) and torch.equal(lp1, lp0) and torch.equal(ns1, ns0)
        assert torch.equal(fin1[3], fin0[3]) and torch.equal(fin1[1], fin0[1]) and torch.equal(fin1[2], fin0[2])
        assert torch.equal(fin1[0], fin0[0])
    finally:
        task.close()
