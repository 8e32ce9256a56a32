# Round 6: 24 resident clips — how should they be cut into decode chains?  
# 224-step greedy decodes (decode only: no log-mel / encoder) of 24 clips as 3 x 8 (round 5's lanes), 2 x 12, 16 + 8, 1 x 24, 4 x 6,
# each chain = its own task, HIP stream and host thread; then the same with an encoder of 8 clips looping on the engine's stream.
#   python tools/chains_ab.py [model]
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
N = 224
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
init_t = torch.tensor(init, device=dev)
sot_index = tok.sot_sequence.index(tok.sot)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


class Job:
    def __init__(self, lo, hi, stream, two_self=True):
        self.f = feats[lo:hi].contiguous()
        self.B = hi - lo
        self.task = hip.HipTask(model, self.B, 1, max(T0, 8), stream=stream, two_launch_self=two_self)
        self.tokens = torch.zeros(self.B, T0 + N + 1, dtype=torch.int64, device=dev)

    def run(self):
        torch.cuda.set_device(dev)
        self.task.reset(); self.task.set_audio(self.f); self.tokens.zero_(); self.tokens[:, :T0] = init_t
        self.task.greedy(self.tokens, params, sot_index, tok.no_speech)


stop_enc = [False]
mel = torch.randn(8, dims.n_mels, 3000, device=dev).half()


def enc_loop(count):
    torch.cuda.set_device(dev)
    while not stop_enc[0]:
        model.encode(mel)
        model.stream.synchronize()
        count[0] += 1


def timed(jobs, reps=3, with_encoder=False):
    ts, encs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        cnt = [0]
        stop_enc[0] = False
        et = threading.Thread(target=enc_loop, args=(cnt,)) if with_encoder else None
        t0 = time.perf_counter()
        if et: et.start()
        th = [threading.Thread(target=j.run) for j in jobs]
        for t in th: t.start()
        for t in th: t.join()
        stop_enc[0] = True
        if et: et.join()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); encs.append(cnt[0])
    return min(ts[1:]), ts, encs


print("model", name, flush=True)
ref = None
for label, cuts in (("3 x 8", [(0, 8), (8, 16), (16, 24)]), ("2 x 12", [(0, 12), (12, 24)]), ("16 + 8", [(0, 16), (16, 24)]),
                    ("1 x 24", [(0, 24)]), ("4 x 6", [(0, 6), (6, 12), (12, 18), (18, 24)]), ("1 x 12", [(0, 12)]), ("1 x 16", [(0, 16)])):
    jobs = [Job(lo, hi, streams[i]) for i, (lo, hi) in enumerate(cuts)]
    clips = sum(j.B for j in jobs)
    for enc in (False, True):
        best, ts, encs = timed(jobs, with_encoder=enc)
        print(f"  {label:8s} {'+ encoder loop' if enc else '              '}: {best:7.1f} ms per 224-step decode, {clips * 30.0 / (best * 1e-3):7.1f} audio-s/s of decode"
              f"  {[round(x, 1) for x in ts]}" + (f"  encoders of 8 clips finished meanwhile: {encs}" if enc else ""), flush=True)
    toks = torch.cat([j.tokens for j in jobs])[:, : T0 + N]
    if ref is None:
        ref = toks.clone()
    elif clips == 24:
        print(f"           rows equal to the 3 x 8 cut's: {int((toks == ref).all(dim=1).sum())} of 24", flush=True)
    for j in jobs:
        j.task.destroy()
    del jobs
