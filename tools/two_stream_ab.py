# Do two half-batch decode chains overlap on one GPU?  (round 5)  The decode step of 8 rows is a chain of ~193 dependent launches
# that keeps the chip's memory pipes busy about two thirds of the time; two INDEPENDENT chains of 4 rows each (two tasks, two
# streams, two host threads) could fill each other's gaps — at the price of streaming the weights twice.
#   python tools/two_stream_ab.py [model]
# Prints wall time of a 224-step greedy decode for: one task of 8 rows | one task of 4 rows | two tasks of 4 rows at once |
# two tasks of 8 rows at once (16 clips: what two pipelined batches would do).
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
feats = (torch.randn(48, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev)
         + 3.0 * torch.randn(48, 1, dims.n_audio_state, generator=g, device=dev)).half()
init_t = torch.tensor(init, device=dev)
sot_index = tok.sot_sequence.index(tok.sot)


class Job:
    def __init__(self, rows, stream):
        self.f = feats[rows].contiguous()
        self.B = self.f.shape[0]
        self.task = hip.HipTask(model, self.B, 1, max(T0, 8), stream=stream)
        self.tokens = torch.zeros(self.B, T0 + N + 1, dtype=torch.int64, device=dev)

    def run(self):
        torch.cuda.set_device(dev)
        self.task.reset(); self.task.set_audio(self.f); self.tokens.zero_(); self.tokens[:, :T0] = init_t
        self.task.greedy(self.tokens, params, sot_index, tok.no_speech)


def timed(jobs, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=j.run) for j in jobs]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:]), ts


s = [torch.cuda.Stream(device=dev) for _ in range(4)]
j8 = Job(slice(0, 8), s[0]); j8b = Job(slice(8, 16), s[1])
j4a = Job(slice(0, 4), s[2]); j4b = Job(slice(4, 8), s[3])
print("model", name, flush=True)
for label, jobs, clips in (("one task, 8 rows", [j8], 8), ("one task, 4 rows", [j4a], 4), ("two tasks of 4 rows at once", [j4a, j4b], 8),
                           ("two tasks of 8 rows at once", [j8, j8b], 16), ("one task, 8 rows (again)", [j8], 8)):
    best, ts = timed(jobs)
    print(f"  {label:32s}: {best:7.1f} ms per 224-step decode = {best / N * 1e3:7.1f} us per step, {clips * 30.0 / (best * 1e-3):7.1f} audio-s/s of decode  {[round(x, 1) for x in ts]}", flush=True)
ref = j8.tokens[:, : T0 + N].clone()
both = torch.cat([j4a.tokens, j4b.tokens])[:, : T0 + N]
print("  tokens of the two 4-row tasks equal to the 8-row task's:", bool((both == ref).all()), flush=True)
# round 5, after 9 - 24 rows moved to gemv8_kernel's row tiles: do chains of 16 rows overlap as chains of 8 do?
j16 = [Job(slice(16 * i, 16 * i + 16), s[i]) for i in range(3)]
for label, jobs, clips in (("one task, 16 rows", j16[:1], 16), ("two tasks of 16 rows at once", j16[:2], 32), ("three tasks of 16 rows at once", j16, 48)):
    best, ts = timed(jobs)
    print(f"  {label:32s}: {best:7.1f} ms per 224-step decode = {best / N * 1e3:7.1f} us per step, {clips * 30.0 / (best * 1e-3):7.1f} audio-s/s of decode  {[round(x, 1) for x in ts]}", flush=True)
