# Does the Infinity Cache (256 MB) serve the cross-attention K/V stream when the SAME layer's K/V is read again?  A one-decoder-layer
# model at large-v3 widths: wh_task_bench_kernel(kind 1) replays the cross-attention launch 64 times on the same K/V (B x 7.68 MB),
# against the 32-layer model where the launches rotate over 32 layers' K/V (HBM-cold).   python tools/mall_probe.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
dev = torch.device("cuda:0")
for layers in (1, 2, 32):
    dims = copy.copy(dims_for("large-v3"))
    dims.n_text_layer = layers; dims.n_audio_layer = 1
    sd = synthetic_state_dict(dims, seed=0, device=dev)
    model = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
    g = torch.Generator(device=dev).manual_seed(4)
    for B in (8, 12, 16, 24, 32):
        feats = torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev).half()
        task = hip.HipTask(model, B, 1, 8, two_launch_cross=True)
        task.set_audio(feats)
        toks = torch.randint(0, 50000, (B, 4), generator=g, device=dev)
        task.prefill(toks, sel=[3])
        ms, nbytes = task.bench_kernel(1, 64)
        print(f"decoder layers {layers:2d}, {B:2d} rows: cross attention {ms * 1e3:6.2f} us per launch, {nbytes / 1e6:6.1f} MB -> {nbytes / (ms * 1e-3) / 1e12:5.2f} TB/s"
              f"   (K/V of all layers: {nbytes * layers / 1e6:7.1f} MB)", flush=True)
        task.destroy()
    del model
    torch.cuda.empty_cache()
