# In-process A/B of decode-step switches that api.cpp reads per call / per capture: one process, one model, a fresh task
# (fresh step graphs) per variant, large-v3 x 8 rows, 224 greedy steps; token ids compared with the first variant.
#   python tools/step_env_ab.py base SOME_SWITCH=1 base
# The product's own switches (tools/README.md) are read once per process: for those run the script once per setting
# (`WH_GEMV_DOT2=1 python tools/step_env_ab.py base`).
# (WH_SAMPLER_EAGER of profiles/r02_probe_sampler_in_graph.txt existed in the library of commit dc21f41 only)
# (the WH_PREFETCH experiment of profiles/r02_probe_prefetch.txt ran through an earlier form of this script on the
#  library of commit 9ffca92, variants `0 1 2 3 7 3:128 3:512 0` = WH_PREFETCH[:WH_PREFETCH_WGS]; that switch was removed)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
variants = sys.argv[1:] or ["base", "base"]
B, N = 8, 224
dims = dims_for("large-v3")
sd = synthetic_state_dict(dims, seed=0, device=dev)
model = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
init = list(tok.sot_sequence); T0 = len(init)
suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm,
                                                     tok.no_speech, tok.eot]))
mask = torch.zeros(dims.n_vocab, dtype=torch.uint8); mask[suppress] = 1; mask = mask.to(dev)
params = hip.GreedyParams(sample_begin=T0, max_steps=N, n_ctx=dims.n_text_ctx, eot=tok.eot,
                          timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                          max_initial_timestamp_index=50, suppress_blank=1, blank_token=tok.encode(" ")[0],
                          suppress_mask=mask.data_ptr())
g = torch.Generator(device=dev).manual_seed(4)
feats = (torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev)
         + 3.0 * torch.randn(B, 1, dims.n_audio_state, generator=g, device=dev)).half()
init_t = torch.tensor(init, device=dev)
sot_index = tok.sot_sequence.index(tok.sot)
ref = None
touched = set()
for v in variants:
    for k in touched: os.environ.pop(k, None)
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("="); os.environ[k] = val; touched.add(k)
    task = hip.HipTask(model, B, 1, max(T0, 8))
    bufs = [torch.zeros(B, T0 + N + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    times = []
    for it in range(4):
        tokens = bufs[it & 1]            # another token buffer every pass: the captured sampler nodes get new arguments
        task.reset(); task.set_audio(feats); tokens.zero_(); tokens[:, :T0] = init_t
        torch.cuda.synchronize(); t0 = time.perf_counter()
        task.greedy(tokens, params, sot_index, tok.no_speech)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    if ref is None: ref = tokens.clone()
    assert bool((bufs[0] == bufs[1]).all()), "the two token buffers of one variant differ"
    print(f"{v:28s}: {min(times[1:]) / N * 1e3:8.1f} us per step (best of 3; passes {[round(x, 1) for x in times]} ms)"
          f"  tokens equal to first variant: {bool((tokens == ref).all())}", flush=True)
    del task
