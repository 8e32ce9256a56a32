"""profiles/rNN_pmc_fetch_size.csv + rNN_pmc_write_size.csv (tools/prof_summary.py --pmc) -> rNN_pmc_traffic.json:
HBM bytes per launch of the decode-step kernels.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of 16 B/lane streaming reads at 64 B);
WRITE_SIZE is uncalibrated and kept raw.  Usage: python tools/pmc_traffic.py fetch.csv write.csv out.json"""
import csv
import json
import re
import sys

# bench name -> regex on "mangled name grid=(...)" rows (large-v3, 8 clips, fp16)
PATTERNS = {
    # round 3: LN + projection + attention in one launch (xattn.hip); the two-launch kernels otherwise
    "attn_decode_cross": r"(xattn8_kernel<8>|xattn8_kernel<8, false>|xattn8_kernelILi8E|attn_decode_kernelIDF16_Li16ELi4ELb0E.*grid=\(768,20,8\))",
    "attn_decode_self": r"(sattn8_kernel|attn_decode_kernelIDF16_Li8ELi8ELb1E.*grid=\(512,20,8\))",
    # gemv8_kernel<PRO, GS, KS, NU, CSm, XW, NRT>: LN = 1, PLAIN = 0, COMBINE = 2 (qkv / cq only in the two-launch form)
    "gemv_qkv": r"gemv8_kernel<1, 2, 4, 5, 1, 8, 1>",
    "gemv_fc1": r"gemv8_kernel<1, 3, 4, 5, 1, 4, 1>",
    "gemv_fc2": r"gemv8_kernel<0, 1, 16, 5, 1, 0, 1>",
    "gemv_out": r"gemv8_kernel<0, 1, 4, 5, 1, 0, 1>",
    "gemv_cq": r"gemv8_kernel<1, 1, 4, 5, 1, 8, 1>",
    "gemv_cout": r"gemv8_kernel<2, 1, 4, 5, 3, 8, 1>",
    "gemv_logits": r"gemv_stream_kernelIDF16_",
}


def read(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r["kernel"]] = (float(r["avg"]), int(r["dispatches"]))
    return rows


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0 "
                       "--sample-len 24 --no-cpu-baseline --no-roofline`, large-v3 B=8 fp16, per launch.  "
                       "FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 "
                       "tallies 128-B requests at 64 B for 16 B/lane streaming reads); WRITE_SIZE is uncalibrated, raw.",
           "kernels": {}}
    for name, pat in PATTERNS.items():
        fk = [k for k in fetch if re.search(pat, k)]
        wk = [k for k in write if re.search(pat, k)]
        if not fk or not wk:
            print(f"no rows for {name} (not launched in this configuration)", file=sys.stderr)
            continue
        fk, wk = max(fk, key=lambda k: fetch[k][1]), max(wk, key=lambda k: write[k][1])
        f_kb, n = fetch[fk]
        w_kb, _ = write[wk]
        out["kernels"][name] = {"fetch_size_kb_raw": round(f_kb, 1), "write_size_kb_raw": round(w_kb, 1),
                                "dispatches": n, "hbm_read_bytes_corrected": int(f_kb * 1024 * 2),
                                "hbm_write_bytes_raw": int(w_kb * 1024)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["hbm_read_bytes_corrected"] + v["hbm_write_bytes_raw"] for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
