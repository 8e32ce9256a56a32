"""Compile every kernel file to gfx950 assembly and look for the two code-generation traps this project has met
(DESIGN.md §3 / §5b):
  1. an empty predicated block — `s_and_saveexec_b64` immediately followed by `s_or_b64 exec, exec` — which is what a
     silently dropped conditional store looks like (beam.hip once lost every logit filter this way);
  2. for the kernels of the decode step, more than ONE group of kernel-argument loads (`s_load ... s[0:1]` ... `s_waitcnt
     lgkmcnt`): every further group is a serialized round trip to the cold kernarg segment (`pin_kernargs`);
  3. register spills in ANY kernel: `.vgpr_spill_count` / `.sgpr_spill_count` / `.private_segment_fixed_size` of the
     kernel descriptors must be 0 (a scratch access per lane per unit is what turned the beam-search row tiles slow).
Usage: python tools/isa_lint.py [file.hip ...]      (no GPU needed; hipcc cross-compiles; ~20-60 s per file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whisper_amd", "csrc")
STEP_KERNELS = ("gemv_kernel", "gemv8_kernel", "gemv_stream_kernel", "attn_decode_kernel", "attn_decode_group_kernel",
                "greedy_partial_kernel", "greedy_final_kernel", "beam_partial_kernel", "beam_row_kernel",
                "beam_update_kernel")


def kernels(asm_path):
    cur, out = None, {}
    for ln in open(asm_path):
        m = re.match(r"^(_Z\S+):\s*;\s*@", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith("s_endpgm"):
            cur = None
        elif s and not s.startswith(";"):
            out[cur].append(s)
    return out


def spills(asm_path):
    """kernel name -> (vgpr spills, sgpr spills, scratch bytes) from the .amdhsa / metadata block of the assembly"""
    out, cur = {}, None
    for ln in open(asm_path):
        m = re.match(r"\s*\.name:\s+(\S+)", ln)
        if m:
            cur = m.group(1).strip("'\"")
            out.setdefault(cur, [0, 0, 0])
        for i, key in enumerate((".vgpr_spill_count:", ".sgpr_spill_count:", ".private_segment_fixed_size:")):
            m = re.match(r"\s*" + re.escape(key) + r"\s+(\d+)", ln)
            if m and cur:
                out[cur][i] = int(m.group(1))
    return out


def lint(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                        "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", path, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        bad = 0
        for name, body in kernels(asm).items():
            empty = sum(1 for a, b in zip(body, body[1:]) if a.startswith("s_and_saveexec_b64") and b.startswith("s_or_b64 exec, exec"))
            groups, pending = 0, False
            for s in body:
                if s.startswith("s_load") and "s[0:1]" in s:
                    pending = True
                elif s.startswith("s_waitcnt") and "lgkmcnt" in s and pending:
                    groups, pending = groups + 1, False
            step = any(k in name for k in STEP_KERNELS)
            if empty or (step and groups > 1):
                bad += 1
                print(f"{os.path.basename(path)}: {name[:110]}: empty predicated blocks {empty}, kernarg load groups {groups}")
        for name, (vs, ss, scratch) in spills(asm).items():
            if (vs or scratch) and "median_generic" not in name:      # the any-width median keeps its window in scratch by design
                bad += 1
                print(f"{os.path.basename(path)}: {name[:110]}: vgpr spills {vs}, sgpr spills {ss}, scratch {scratch} B/lane")
        return bad


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = sum(lint(f) for f in files)
    print(f"{len(files)} files, {bad} kernels flagged")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
