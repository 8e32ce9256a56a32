"""CPU-only checks of the drop-in boundary: libwhisper_hip.so loads, exports every symbol that
include/whisper_hip.h declares (and nothing in the binding is missing from the header), the pure-host entry
points behave, and the product refuses to run without a GPU instead of falling back."""
import os
import re

import pytest
import torch

from whisper_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "whisper_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wh_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_header():
    lib = hip.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in whisper_hip.h but not exported"
    assert sorted(hip.SIGNATURES) == syms, "ctypes binding and header disagree"


def test_library_exports_nothing_but_the_header():
    """the dynamic symbol table of libwhisper_hip.so holds exactly the wh_* functions include/whisper_hip.h
    declares — no whk::launch_* host wrappers, no kernel stubs (linker version script csrc/exports.map)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", hip.lib_path()], check=True, capture_output=True, text=True).stdout
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == header_symbols(), sorted(set(defined) ^ set(header_symbols()))


def test_status_strings_and_version():
    lib = hip.lib()
    assert lib.wh_abi_version() == 1
    assert lib.wh_status_string(0) == b"ok"
    assert b"workspace" in lib.wh_status_string(2)
    assert b"hand-off" in lib.wh_status_string(6) and lib.wh_status_string(99) == b"unknown status"
    assert b"running" in lib.wh_status_string(7)                     # WH_RUNNING: wh_task_poll, not an error


def test_argument_validation_without_gpu():
    """entry points reject bad arguments before touching the device"""
    lib = hip.lib()
    assert lib.wh_log_mel(None, 480000, 1, 80, None, None, None, None) == 1
    assert lib.wh_median_filter(None, None, 1, 10, 7, None) == 1
    assert lib.wh_dtw_trace(None, 4, 4, None, None) == 1
    assert lib.wh_encoder_workspace_bytes(None, 1) == 0
    assert lib.wh_task_workspace_bytes(None, 1, 1, 8, 0) == 0
    assert lib.wh_task_poll(None, None) == 1
    assert lib.wh_task_greedy_begin(None, None, None, 0, 0, -1, None, None, None) == 1       # null task: refused before any device work


def test_blob_layout_is_deterministic():
    from whisper_amd.synthetic import dims_for
    d = dims_for("base")
    a, na = hip.blob_layout(d, hip.WH_F16)
    b, nb = hip.blob_layout(d, hip.WH_F16)
    assert a == b and na == nb
    offs = sorted(v[0] for v in a.values())
    assert all(o % 256 == 0 for o in offs) and len(set(offs)) == len(offs)
    # fp16 step-weight bytes of SURVEY.md Appendix A: 14*D^2*L + V*D parameters
    D, L, V = d.n_text_state, d.n_text_layer, d.n_vocab
    mats = sum(2 * int(torch.tensor(s).prod()) for n, (o, s, m) in a.items()
               if m and n.startswith("dec.") and not n.endswith("ckv_w")) + 2 * V * D    # ckv_w is per-window, not per-step
    assert mats == 2 * (14 * D * D * L + V * D)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    import numpy as np
    import whisper_amd
    with pytest.raises(hip.HipError):
        whisper_amd.log_mel_spectrogram(np.zeros(16000, dtype=np.float32))
    from whisper_amd.synthetic import dims_for, synthetic_state_dict
    from whisper_amd.model import ModelDimensions, Whisper
    from oracle.model import dims_dict
    dims = dims_for("micro.en")
    model = Whisper(ModelDimensions(**dims_dict(dims)), {}, device="cpu")
    with pytest.raises(hip.HipError):
        model.encoder(torch.zeros(1, 80, 3000))


def test_audio_library_exports_header():
    """libwhisper_audio.so exports every function include/whisper_audio.h declares"""
    import re
    from whisper_amd import audio
    text = open(os.path.join(ROOT, "include", "whisper_audio.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", text)))
    assert syms == ["wh_flac_decode", "wh_flac_error", "wh_flac_free"]
    lib = audio._audio_lib()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in whisper_audio.h but not exported"


def test_struct_layouts_match_the_header(tmp_path):
    """every struct that crosses the boundary has the same size and field offsets in the ctypes binding as a C
    compiler gives the declarations in include/whisper_hip.h (gcc, plain C99 — the header must stay a C header)"""
    import ctypes as C
    import subprocess
    structs = {"wh_dims": hip.Dims, "wh_layer_weights": hip.LayerWeights, "wh_model_weights": hip.ModelWeights,
               "wh_greedy_params": hip.GreedyParams, "wh_beam_params": hip.BeamParams}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "whisper_hip.h"', '#include "whisper_audio.h"',
             'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_shipped_library_reads_no_experiment_switches():
    """developer A/B switches and rejected experiments live in the -DWH_DEV build only
    (`make -C whisper_amd/csrc dev`).  The shipped library's data must not contain a single `WH_*` environment-variable
    name besides WH_NO_GRAPH (decode steps launched eagerly: a support switch)."""
    import re
    from whisper_amd import hip
    with open(hip.lib_path(), "rb") as f:
        blob = f.read()
    names = {m.decode() for m in re.findall(rb"WH_[A-Z0-9_]{3,}", blob)}
    assert names == {"WH_NO_GRAPH"}, names
