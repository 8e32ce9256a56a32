"""profiles/rNN_pmc_fetch_size.csv + rNN_pmc_write_size.csv (tools/prof_summary.py --pmc) -> rNN_pmc_traffic.json:
HBM bytes per launch of the decode-step kernels.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of 16 B/lane streaming reads at 64 B);
WRITE_SIZE is uncalibrated and kept raw.  Usage: python tools/pmc_traffic.py fetch.csv write.csv out.json [rows of the chain, default 24]"""
import csv
import json
import re
import sys

ROWS = int(sys.argv[4]) if len(sys.argv) > 4 else 24
NRT = 1 if ROWS <= 8 else 2 if ROWS <= 16 else 3          # row tiles of gemv8_kernel (csrc/gemv.hip)
# bench name -> regex on "mangled name grid=(...)" rows (large-v3, fp16, one decode chain of ROWS rows)
PATTERNS = {
    # <= 8 rows: LN + projection + attention in one launch (xattn.hip); the two-launch kernels otherwise
    "attn_decode_cross": (r"(xattn8_kernel<8>|xattn8_kernel<8, false>|xattn8_kernelILi8E)" if ROWS <= 8 else
                          rf"attn_decode_kernelIDF16_Li16ELi4ELb0E.*grid=\(768,20,{ROWS}\)"),
    "attn_decode_self": rf"(sattn8_kernel|attn_decode_kernelIDF16_Li8ELi8ELb1E.*grid=\(512,20,{ROWS}\))",
    # gemv8_kernel<PRO, GS, KS, NU, CSm, XW, NRT>: LN = 1, PLAIN = 0, COMBINE = 2
    "gemv_qkv": rf"gemv8_kernel<1, 2, 4, 5, 1, 8, {NRT}>",
    "gemv_fc1": rf"gemv8_kernel<1, 3, 4, 5, 1, 4, {NRT}>" if ROWS <= 8 else rf"gemv8_kernel<1, 3, 2, 10, 1, 8, {NRT}>",   # 9+ rows: 8 LayerNorm waves
    "gemv_fc2": rf"gemv8_kernel<0, 1, 16, 5, 1, 0, {NRT}>",
    "gemv_out": rf"gemv8_kernel<0, 1, 4, 5, 1, 0, {NRT}>",
    "gemv_cq": rf"gemv8_kernel<1, 1, 4, 5, 1, 8, {NRT}>",
    "gemv_cout": rf"gemv8_kernel<2, 1, 4, 5, 3, 8, {NRT}>",
    "merge_partials": r"merge_partials_kernel",
    "gemv_logits": r"(gemv_stream_kernelIDF16_|gemv_rows48_stream_kernel)",
}


def read(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r["kernel"]] = (float(r["avg"]), int(r["dispatches"]))
    return rows


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 0 "
                       f"--sample-len 24 --no-cpu-baseline --no-roofline --no-extras`, large-v3 fp16, one decode chain of {ROWS} rows, per launch.  "
                       "FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 "
                       "tallies 128-B requests at 64 B for 16 B/lane streaming reads); WRITE_SIZE is uncalibrated, raw.",
           "rows": ROWS, "kernels": {}}
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
