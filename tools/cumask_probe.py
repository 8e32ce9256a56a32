# Round 6 probe: can the MFMA-bound encoder of the NEXT chain run BESIDE the HBM/latency-bound decode of the current one if the two
# are kept on disjoint CUs (hipExtStreamCreateWithCUMask)?  Unmasked they only time-slice: a persistent 256-workgroup GEMM holds every
# CU's LDS, so each of the decode chain's ~230 launches per token waits for a tile to end (tools/chains_ab.py "+ encoder loop").
#   python tools/cumask_probe.py [encoder CUs, default 32]
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
rt = C.CDLL("libamdhip64.so")
N_CU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = C.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


n_enc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc_bits = set(range(n_enc)); dec_bits = set(range(n_enc, N_CU))
print(f"{N_CU} CUs: encoder stream on {n_enc}, decode stream on {N_CU - n_enc}", flush=True)
s_enc, s_dec, s_all = masked_stream(enc_bits), masked_stream(dec_bits), torch.cuda.Stream(device=dev)
N = 224
dims = dims_for("large-v3")
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
mel = torch.randn(8, dims.n_mels, 3000, device=dev).half()
init_t = torch.tensor(init, device=dev)
torch.cuda.synchronize()


def decode_on(stream, reps=2):
    task = hip.HipTask(model, 24, 1, max(T0, 8), stream=stream)
    toks = torch.zeros(24, T0 + N + 1, dtype=torch.int64, device=dev)
    ts = []
    for _ in range(reps + 1):
        with torch.cuda.stream(stream):
            task.reset(); task.set_audio(feats); toks.zero_(); toks[:, :T0] = init_t
            torch.cuda.synchronize(); t0 = time.perf_counter()
            task.greedy(toks, params, 0, tok.no_speech)
            stream.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    task.destroy()
    return min(ts[1:]), toks[:, : T0 + N].clone()


def encode_on(stream, n=3):
    model.stream = stream
    model.encode(mel); stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.encode(mel)
    stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base, ref = decode_on(s_all)
print(f"decode 24 rows x {N} steps, all CUs, alone          : {base:7.1f} ms", flush=True)
dm, t2 = decode_on(s_dec)
print(f"decode on the {N_CU - n_enc}-CU stream, alone (grids sized for 256): {dm:7.1f} ms   tokens equal {bool((t2 == ref).all())}", flush=True)
print(f"encoder of 8 clips, all CUs, alone                : {encode_on(s_all):7.1f} ms", flush=True)
e_m = encode_on(s_enc)
print(f"encoder of 8 clips on the {n_enc}-CU stream, alone       : {e_m:7.1f} ms", flush=True)
for label, sd_, se_ in (("masked: decode | encoder on disjoint CUs", s_dec, s_enc), ("unmasked: both on all CUs              ", s_all, torch.cuda.Stream(device=dev))):
    stop, cnt = [False], [0]

    def enc_loop():
        torch.cuda.set_device(dev)
        model.stream = se_
        while not stop[0]:
            model.encode(mel); se_.synchronize(); cnt[0] += 1
    th = threading.Thread(target=enc_loop); th.start()
    time.sleep(0.2)
    c0 = cnt[0]
    t, _ = decode_on(sd_, reps=1)
    done = cnt[0] - c0
    stop[0] = True; th.join()
    print(f"{label}: decode {t:7.1f} ms per chain while the encoder loops (about {done / 2:.1f} encoders of 8 clips finish per chain)", flush=True)
