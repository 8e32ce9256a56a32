# Why is one find_alignment_batch call in five 40-60 ms slower (bench extras.word_timestamps: [75.8, 76.2, 77.2, 75.9, 134.3])?
# Hypothesis: a full (generation 2) pass of Python's cyclic garbage collector inside the call's host part.  Logs every collection
# (generation, duration) around ten calls, then repeats with gc.freeze() after the warm-up.
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_amd
from whisper_amd import hip
from whisper_amd.model import ModelDimensions, Whisper
from whisper_amd.synthetic import dims_dict, dims_for, synthetic_state_dict
from whisper_amd.timing import find_alignment_batch
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
dims = dims_for(name)
sd = synthetic_state_dict(dims, seed=0, device=dev)
eng = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
m = Whisper(ModelDimensions(**dims_dict(dims)), {}, device=dev); m.adopt_engine(torch.float16, eng)
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
B = 8
g = torch.Generator(device=dev).manual_seed(1)
audio = torch.randn(B, 480000, generator=g, device=dev) * 0.1
mel = whisper_amd.log_mel_spectrogram(audio, dims.n_mels)
text = [[int(x) for x in torch.randint(300, 40000, (182,), generator=torch.Generator().manual_seed(b)).tolist()] for b in range(B)]
events = []
t_gc = [0.0]
def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info["collected"]))
gc.callbacks.append(cb)
print("tracked objects:", len(gc.get_objects()), "thresholds", gc.get_threshold(), flush=True)
for label in ("as is", "after gc.freeze()"):
    find_alignment_batch(m, tok, text, mel.half(), [3000] * B)
    torch.cuda.synchronize()
    if label != "as is":
        gc.collect(); gc.freeze()
    rows = []
    for i in range(10):
        events.clear()
        t0 = time.perf_counter()
        find_alignment_batch(m, tok, text, mel.half(), [3000] * B)
        torch.cuda.synchronize()
        rows.append(((time.perf_counter() - t0) * 1e3, [(gen, round(ms, 1)) for gen, ms, _ in events if ms > 0.5]))
    print(label)
    for ms, ev in rows:
        print(f"   call {ms:6.1f} ms   gc passes > 0.5 ms (generation, ms): {ev}", flush=True)
