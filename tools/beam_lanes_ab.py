# beam 5 x 8 clips x 64 steps, three passes in flight: ONE host thread polling (decode_many -> run_interleaved) vs three host threads
# (run_in_lanes), same process, same box.   python tools/beam_lanes_ab.py [sleep_us]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_amd
from whisper_amd import hip
from whisper_amd.model import ModelDimensions, Whisper
from whisper_amd.synthetic import dims_dict, dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
from whisper_amd import decoding
dev = torch.device("cuda:0")
dims = dims_for("large-v3")
sd = synthetic_state_dict(dims, seed=0, device=dev)
eng = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
m = Whisper(ModelDimensions(**dims_dict(dims)), {}, device=dev); m.adopt_engine(torch.float16, eng)
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
g = torch.Generator(device=dev).manual_seed(1)
audio = torch.randn(8, 480000, generator=g, device=dev) * 0.1
for label, opts in (("beam 5 x 64 steps", whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=64, beam_size=5, suppress_tokens=[-1, tok.eot])),
                    ("greedy x 64 steps", whisper_amd.DecodingOptions(language="en", fp16=True, sample_len=64, suppress_tokens=[-1, tok.eot]))):
    def one():
        return whisper_amd.decode(m, whisper_amd.log_mel_spectrogram(audio, dims.n_mels), opts)
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter(); one(); one(); torch.cuda.synchronize(); alone = (time.perf_counter() - t0) / 2 * 1e3
    res = {}
    for mode in ("one thread", "three threads", "one thread", "three threads"):
        def run(n):
            if mode == "one thread":
                return whisper_amd.decode_many(m, [audio] * n, opts, in_flight=3, chain_rows=None)
            return decoding.run_in_lanes(m, [one] * n, 3, torch.float16)
        run(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(9); torch.cuda.synchronize()
        res.setdefault(mode, []).append((time.perf_counter() - t0) / 9 * 1e3)
    print(f"{label}: one at a time {alone:.1f} ms per pass; three in flight: " + "; ".join(f"{k} {[round(x, 1) for x in v]}" for k, v in res.items()), flush=True)
