# Two decode chains of R rows each on two streams, driven from one thread (wh_task_greedy_begin + wh_task_poll), N steps — to be run
# under `rocprofv3 --kernel-trace`; tools/overlap_from_trace.py then measures how much of the time kernels of BOTH streams were running.
#   python tools/overlap_probe.py R [steps] [chains] [form: 0 default, 1 fused self attention too, 2 cross attention as two launches]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
from whisper_amd.tokenizer import get_tokenizer
dev = torch.device("cuda:0")
R = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 48; C_ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
FORM = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
feats = (torch.randn(R * C_, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev)
         + 3.0 * torch.randn(R * C_, 1, dims.n_audio_state, generator=g, device=dev)).half()
init_t = torch.tensor(init, device=dev)
SKIP = int(os.environ.get('OVERLAP_SKIP_STREAMS', '0'))          # create this many streams first and leave them idle (hardware-queue mapping)
spare = [torch.cuda.Stream(device=dev) for _ in range(SKIP)]
for sp in spare:                                   # ... used once, so that the runtime has bound them to a hardware queue
    with torch.cuda.stream(sp):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(C_)]
tasks = [hip.HipTask(model, R, 1, max(T0, 8), stream=streams[i], fused_self=bool(FORM & 1), two_launch_cross=bool(FORM & 2)) for i in range(C_)]
toks = [torch.zeros(R, T0 + N + 1, dtype=torch.int64, device=dev) for _ in range(C_)]


def run(which):
    pend = {}
    for i in which:
        with torch.cuda.stream(streams[i]):
            tasks[i].reset(); tasks[i].set_audio(feats[i * R:(i + 1) * R].contiguous()); toks[i].zero_(); toks[i][:, :T0] = init_t
            pend[i] = tasks[i].greedy_begin(toks[i], params, 0, tok.no_speech)
    while pend:
        for i in list(pend):
            if pend[i].poll() is not None:
                del pend[i]
    torch.cuda.synchronize()


if os.environ.get("OVERLAP_THREADS") == "1":       # one BLOCKED host thread per chain (wh_task_greedy) instead of one thread polling
    import threading

    def run(which):                                # noqa: F811
        def w(i):
            torch.cuda.set_device(dev)
            cur = torch.cuda.Stream(device=dev) if os.environ.get("OVERLAP_OWN_CURRENT") == "1" else None
            with (torch.cuda.stream(cur) if cur is not None else torch.cuda.stream(torch.cuda.current_stream(dev))):
                tasks[i].reset(); tasks[i].set_audio(feats[i * R:(i + 1) * R].contiguous()); toks[i].zero_(); toks[i][:, :T0] = init_t
                tasks[i].greedy(toks[i], params, 0, tok.no_speech)
        th = [threading.Thread(target=w, args=(i,)) for i in which]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        torch.cuda.synchronize()

run(range(C_)); run(range(C_))                    # warm-up: graphs captured
t0 = time.perf_counter(); run([0]); one = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); run(range(C_)); both = (time.perf_counter() - t0) * 1e3
print(f"rows {R} form {FORM}: one chain {one:.1f} ms, {C_} chains at once {both:.1f} ms ({both / one:.2f} x) for {N} steps", flush=True)
