#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, subprocess
for lib in ("", "whisper_amd/libwhisper_hip_r40.so"):
    env = dict(os.environ)
    if lib: env["WHISPER_AMD_LIB"] = lib
    code = '''
import sys; sys.argv=["x"]
import torch, bench
from whisper_amd import hip
from whisper_amd.synthetic import dims_for, synthetic_state_dict
dev = torch.device("cuda:0")
for name, shapes in (("large-v3", ((16, 1), (4, 5), (24, 1), (32, 1), (8, 5), (48, 1))), ("turbo", ((32, 1), (16, 1)))):
    dims = dims_for(name)
    sd = synthetic_state_dict(dims, seed=0, device=dev)
    model = hip.HipModel(dims, hip.WH_F16, hip.pack_weights(sd, dims, hip.WH_F16, dev)); del sd
    g = torch.Generator(device=dev).manual_seed(4)
    for B, G in shapes:
        feats = torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g, device=dev).half()
        r = bench.step_roofline(model, feats, B, G, 35)
        print(f"  {name} {B:2d} audio x {G} rows: step {r['step_us']:8.1f} us", flush=True)
        model.drop_cached_tasks()
    del model
'''
    print("library:", lib or "product (9-24 rows on gemv8 x 2 / 3 row tiles)", flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
PY
