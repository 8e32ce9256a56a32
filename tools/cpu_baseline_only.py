"""developer helper: time bench.cpu_baseline alone (no GPU needed) with progress + stack dumps"""
import faulthandler, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, repeat=True, file=sys.stderr)
import numpy as np
import bench
from whisper_amd.synthetic import dims_for
from whisper_amd.tokenizer import get_tokenizer
model = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
args = types.SimpleNamespace(model=model, cpu_steps=12, cpu_repeats=3, sample_len=224, parity_steps=0, cpu_threads=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dims = dims_for(model)
tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
init = list(tok.sot_sequence)
suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.eot]))
audio = (np.random.default_rng(0).standard_normal((1, 480000)) * 0.05).astype(np.float32)
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), file=sys.stderr, flush=True)
from whisper_amd.synthetic import synthetic_state_dict
sd = synthetic_state_dict(dims, seed=0)
base, parity = bench.cpu_baseline(args, dims, init, suppress, tok, audio, sd, None)      # no HIP pass here: parity is skipped
print(base)
