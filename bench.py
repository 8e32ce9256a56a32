#!/usr/bin/env python
"""bench.py — audio-seconds transcribed per wall-second on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic 30 s clips per GPU:
log-mel (HIP) -> AudioEncoder (MFMA) -> cross-KV -> greedy decode with the device-side sampling loop.
Workload at N=1 = BASELINE.json configs[2]: large-v3 dims, batch = 8 x 30 s, greedy, fp16.  The step
count is fixed: `sample_len` forced tokens per clip with EOT suppressed, so the work is
identical run to run.  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, each rank decodes its own 8 clips (weak scaling, no collective in the step);
rank 0 builds the packed weight blob and RCCL-broadcasts it over xGMI at load.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def log(msg: str) -> None:
    """progress to stderr (stdout carries exactly one JSON line)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s float4-copy achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="large-v3")
    p.add_argument("--batch", type=int, default=8, help="30 s clips per GPU")
    p.add_argument("--sample-len", type=int, default=224, help="forced decode steps per clip (n_text_ctx // 2)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=12, help="decode steps timed on the CPU baseline")
    p.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = all usable cores)")
    return p.parse_args()


def synth_audio(batch: int, rank: int, device) -> torch.Tensor:
    """seeded noise + three tones so the mel is not flat; clip index = global clip id."""
    n = 480000
    t = np.arange(n) / 16000.0
    clips = []
    for b in range(batch):
        rng = np.random.default_rng(rank * batch + b)
        x = rng.standard_normal(n).astype(np.float32) * 0.05
        for f, a in ((220.0 + 20 * b, 0.2), (1300.0, 0.1), (3100.0, 0.05)):
            x += (a * np.sin(2 * np.pi * f * t)).astype(np.float32)
        clips.append(x)
    return torch.from_numpy(np.stack(clips)).to(device)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist = None

    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    log(f"rank {rank}/{world} device ready")
    from whisper_amd import hip
    from whisper_amd.audio import log_mel_spectrogram
    from whisper_amd.launcher import broadcast_weights
    from whisper_amd.synthetic import dims_for, synthetic_state_dict
    from whisper_amd.tokenizer import get_tokenizer

    dims = dims_for(args.model)
    dtype = hip.WH_F16
    # ---- weights: rank 0 packs, everyone else receives the blob over RCCL -----------------------
    blob = None
    if rank == 0:
        sd = synthetic_state_dict(dims, seed=0, device=device)
        blob = hip.pack_weights(sd, dims, dtype, device)
        del sd
        torch.cuda.empty_cache()
    blob = broadcast_weights(blob, dims, dtype, device, dist)
    model = hip.HipModel(dims, dtype, blob)
    log(f"weights packed: {blob.numel() / 1e9:.2f} GB")

    B, N = args.batch, args.sample_len
    multilingual = dims.n_vocab >= 51865
    tok = get_tokenizer(multilingual, num_languages=dims.n_vocab - 51765 - int(multilingual), language="en",
                        task="transcribe")
    init = list(tok.sot_sequence)
    T0 = len(init)
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev,
                                                         tok.sot_lm, tok.no_speech, tok.eot]))   # + EOT: fixed N
    mask = torch.zeros(dims.n_vocab, dtype=torch.uint8)
    mask[suppress] = 1
    mask = mask.to(device)
    params = hip.GreedyParams(sample_begin=T0, max_steps=N, n_ctx=dims.n_text_ctx, eot=tok.eot,
                              timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                              max_initial_timestamp_index=50, suppress_blank=1,
                              blank_token=tok.encode(" ")[0], suppress_mask=mask.data_ptr())

    audio = synth_audio(B, rank, device)
    task = hip.HipTask(model, B, 1, max(T0, 8))
    tokens = torch.zeros(B, T0 + N + 1, dtype=torch.int64, device=device)
    init_t = torch.tensor(init, device=device)
    sot_index = tok.sot_sequence.index(tok.sot)

    def one_pass():
        mel = log_mel_spectrogram(audio, dims.n_mels)            # (B, n_mels, 3000) fp32 on device
        feats = model.encode(mel)
        task.reset()
        task.set_audio(feats)
        tokens.zero_()
        tokens[:, :T0] = init_t
        n, sum_lp, nsp = task.greedy(tokens, params, sot_index, tok.no_speech)
        return n

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        n_tok = one_pass()
        torch.cuda.synchronize(device)
        log("warmup pass done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_tok = one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert n_tok == T0 + N, (n_tok, T0, N)
    ms_per_step = elapsed / args.steps * 1e3
    log(f"timed: {ms_per_step:.1f} ms per pass")
    audio_s = 30.0 * B * world * args.steps
    value = audio_s / elapsed

    out = {
        "metric": "audio-seconds transcribed per wall-second (large-v3 greedy)",
        "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{args.model} dims (random-init weights), {B} x 30 s synthetic clips per GPU, "
                               f"greedy, fp16 weights/KV + fp32 accumulate, {N} forced decode steps per clip "
                               f"(EOT suppressed), log-mel + encoder + cross-KV + decode timed",
                   "clips_per_gpu": B, "sample_len": N, "parallelism": f"dp{world} (clips sharded, no step collective)"},
    }

    # ---- roofline of the dominant kernels: HIP events on the launch stream, layer-rotated (HBM-cold) ----
    if rank == 0 and not args.no_roofline:
        kinds = {"decode_step": 0, "attn_decode_cross": 1, "attn_decode_self": 2, "gemv_qkv": 3, "gemv_fc1": 4, "gemv_fc2": 5,
                 "gemv_logits": 6, "gemv_out": 7}
        kern = {}
        for name, kind in kinds.items():
            ms, nbytes = task.bench_kernel(kind, 64 if kind else 16)
            log(f"kernel {name}: {ms * 1e3:.1f} us, {nbytes / (ms * 1e-3) / 1e9:.0f} GB/s")
            kern[name] = {"avg_us": round(ms * 1e3, 2), "bytes": nbytes, "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1)}
        # the MFMA-bound side of the pass, for orientation: AudioEncoder.forward on the batch (SURVEY.md §8d FLOP count)
        mel = log_mel_spectrogram(audio, dims.n_mels)
        model.encode(mel)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(3):
            model.encode(mel)
        torch.cuda.synchronize(device)
        enc_ms = (time.perf_counter() - t0) / 3 * 1e3
        D_, L_, M_ = dims.n_audio_state, dims.n_audio_layer, dims.n_mels
        enc_flop = B * (2 * 3000 * M_ * 3 * D_ + 2 * 1500 * D_ * 3 * D_ + L_ * (24 * 1500 * D_ * D_ + 4 * 1500 * 1500 * D_))
        kern["encoder_forward"] = {"avg_us": round(enc_ms * 1e3, 1), "flop": enc_flop,
                                   "TFLOPs": round(enc_flop / (enc_ms * 1e-3) / 1e12, 1), "mfma_peak_TFLOPs": 2500.0}
        log(f"encoder forward: {enc_ms:.1f} ms, {enc_flop / (enc_ms * 1e-3) / 1e12:.0f} TFLOP/s")
        dom = kern["attn_decode_cross"]
        # HBM traffic per launch from the PMC counters: they cannot be read from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/, per round),
        # and only when it was taken on this very workload; otherwise null.
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.isfile(tf) and args.model == "large-v3" and B == 8:
            with open(tf) as f:
                pm = json.load(f)["kernels"].get("attn_decode_cross")
            if pm:
                traffic = pm["hbm_read_bytes_corrected"] + pm["hbm_write_bytes_raw"]
        out["roofline"] = {"bound": "hbm", "kernel": "attn_decode_kernel<half> (cross-attention KV stream)",
                           "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 "
                                             "correction + WRITE_SIZE raw, bytes per launch)" if traffic else None,
                           "bytes_per_launch": dom["bytes"], "avg_us": dom["avg_us"], "all_kernels": kern}
    task.close()

    # ---- CPU baseline: the oracle (fp32 torch-CPU restatement of the reference) on this box's host cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, dims, init, suppress, tok, audio[:1].cpu().numpy())

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, dims, init, suppress, tok, audio_np):
    """Oracle = "port": same algorithm as the reference's CPU fp32 path.  Bounded sample: 1 clip, log-mel +
    encoder + `cpu_steps` decode steps; audio-s/s extrapolated linearly to `sample_len` steps."""
    import oracle
    from whisper_amd.synthetic import synthetic_state_dict
    from whisper_amd.utils import usable_cores
    cores = getattr(args, "cpu_threads", 0) or usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: building {args.model} fp32 oracle on {cores} host threads")
    sd = synthetic_state_dict(dims, seed=0, device="cpu")
    om = oracle.OracleModel(dims, sd)
    log("cpu_baseline: oracle model ready")
    filt = oracle.mel_filterbank(dims.n_mels)
    t0 = time.perf_counter()
    mel = oracle.log_mel_spectrogram(audio_np[0], filt)
    t_mel = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        feats = om.encoder(mel[None])
    t_enc = time.perf_counter() - t0
    log(f"cpu_baseline: log-mel {t_mel:.2f}s encoder {t_enc:.2f}s")
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx,
                                 timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                                 suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)
    k = args.cpu_steps
    t0 = time.perf_counter()
    with torch.no_grad():
        oracle.greedy_decode(om, feats, init, k, rules)
    t_dec = time.perf_counter() - t0
    per_step = t_dec / k
    log(f"cpu_baseline: {k} decode steps in {t_dec:.2f}s")
    total = t_mel + t_enc + per_step * args.sample_len
    return {"value": round(30.0 / total, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"1 clip of the same workload: log-mel {t_mel:.2f}s + encoder {t_enc:.2f}s + {k} decode steps "
                      f"at {per_step * 1e3:.0f} ms/step, extrapolated to {args.sample_len} steps (fp32, torch CPU, "
                      f"{cores} threads)"}


if __name__ == "__main__":
    main()
