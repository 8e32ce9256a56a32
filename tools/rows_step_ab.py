# One decode step at several row counts (graph replay, HIP events: bench.step_roofline) for the library selected with
# WHISPER_AMD_LIB — run once per build to A/B a compile-time switch on the 9..48-row kernels:
#   python tools/rows_step_ab.py                      (product)
#   WHISPER_AMD_LIB=whisper_amd/libwhisper_hip_plain.so python tools/rows_step_ab.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = sys.argv[:1]
import bench
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
dev = torch.device("cuda:0")
name = os.environ.get("ROWS_MODEL", "large-v3")
dims = dims_for(name)
sd = synthetic_state_dict(dims, seed=0, device=dev)
model = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
g = torch.Generator(device=dev).manual_seed(4)
print("library:", os.environ.get("WHISPER_AMD_LIB", "whisper_amd/libwhisper_hip.so"), "model:", name, flush=True)
shapes = ((8, 1), (8, 5), (16, 1), (32, 1), (4, 5)) if name == "large-v3" else ((1, 1), (1, 5), (8, 1))
if os.environ.get("ROWS_SHAPES"):                # e.g. ROWS_SHAPES=24x1,20x1,17x1
    shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["ROWS_SHAPES"].split(","))
for B, G in shapes:
    feats = torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev).half()
    r = bench.step_roofline(model, feats, B, G, 35)
    print(f"  {B:2d} audio x {G} rows: step {r['step_us']:8.1f} us  {r['step_GBps']:7.1f} GB/s  frac {r['step_frac']:.3f}", flush=True)
    model.drop_cached_tasks()
