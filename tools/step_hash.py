# A/B helper for bit-identical kernel variants: a short greedy decode at several row counts, printing a digest of the tokens and of
# the summed log-probabilities — run it under two settings of a developer switch (libwhisper_hip_dev.so) and compare the lines.
#   WHISPER_AMD_LIB=whisper_amd/libwhisper_hip_dev.so python tools/step_hash.py [model] ; WH_NO_TAIL_MERGE=1 ... python tools/step_hash.py
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
N = 40
dims = dims_for(name)
sd = synthetic_state_dict(dims, seed=0, device=dev)
model = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
init = list(tok.sot_sequence); T0 = len(init)
suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.eot]))
mask = torch.zeros(dims.n_vocab, dtype=torch.uint8); mask[suppress] = 1; mask = mask.to(dev)
params = hip.GreedyParams(sample_begin=T0, max_steps=N, n_ctx=dims.n_text_ctx, eot=tok.eot, timestamp_begin=tok.timestamp_begin,
                          no_timestamps=tok.no_timestamps, max_initial_timestamp_index=50, suppress_blank=1,
                          blank_token=tok.encode(" ")[0], suppress_mask=mask.data_ptr())
g = torch.Generator(device=dev).manual_seed(4)
feats = (torch.randn(24, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev)
         + 3.0 * torch.randn(24, 1, dims.n_audio_state, generator=g, device=dev)).half()
sot_index = tok.sot_sequence.index(tok.sot)
for B, two in ((24, False), (20, False), (16, False), (9, False), (8, True), (8, False), (3, True), (1, True)):
    task = hip.HipTask(model, B, 1, max(T0, 8), two_launch_cross=two)
    tokens = torch.zeros(B, T0 + N + 1, dtype=torch.int64, device=dev)
    tokens[:, :T0] = torch.tensor(init, device=dev)
    task.reset(); task.set_audio(feats[:B].contiguous())
    out = task.greedy(tokens, params, sot_index, tok.no_speech)
    torch.cuda.synchronize()
    slp = out[0] if isinstance(out, tuple) else out
    h = hashlib.sha1(tokens.cpu().numpy().tobytes()).hexdigest()[:12]
    extra = ""
    try:
        extra = hashlib.sha1(torch.as_tensor(slp).float().cpu().numpy().tobytes()).hexdigest()[:12]
    except Exception:
        pass
    print(f"{B:2d} rows{' two-launch cross' if two else '':17s} tokens {h}  sum_logprobs {extra}", flush=True)
    task.destroy()
