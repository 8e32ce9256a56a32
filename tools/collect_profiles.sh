#!/bin/bash
# Copy what tools/final_profile.sh (and the GPU parity tests) left under gpurun_out/ into profiles/ as round NN's evidence:
#   bash tools/collect_profiles.sh 03
set -e
R=${1:?round number, e.g. 03}
cd "$(dirname "$0")/.."
F=gpurun_out/final
P=profiles/r${R}
cat $F/bench.err $F/bench.out > ${P}_bench.log
cp $F/kernel_stats.csv ${P}_kernel_stats.csv
cp $F/kernel_stats_extras.csv ${P}_kernel_stats_extras.csv
cp $F/pmc_fetch_size.csv ${P}_pmc_fetch_size.csv
cp $F/pmc_write_size.csv ${P}_pmc_write_size.csv
cp $F/pmc_traffic.json ${P}_pmc_traffic.json
cp $F/pmc_lds.csv ${P}_pmc_lds.csv
cp $F/final_pmc_mfma.csv ${P}_pmc_mfma.csv
cp $F/final_pmc_mfma_raw.csv ${P}_pmc_mfma_raw.csv
cp $F/probe_gemm.txt ${P}_probe_gemm.txt
cp $F/probe_flash_layout.txt ${P}_probe_flash_layout.txt
python - "$R" <<'PY'
import json, os, sys
out = {"source": "tests/test_wide_gpu.py (pytest -m gpu) via tests/conftest.py::write_report, MI355X"}
for name in ("fp16_large_v3_greedy", "fp16_large_v3_beam5", "turbo_dims", "conditioned_large_v3", "conditioned_turbo",
             "conditioned_large_v3_beam5", "alignment_conditioned_turbo", "alignment_conditioned_large_v3",
             "conditioned_base_x1", "lanes_one_thread", "conditioned_large_v3_24_rows"):
    p = os.path.join("gpurun_out", "parity", name + ".json")
    if os.path.exists(p):
        out[name] = json.load(open(p))
json.dump(out, open(f"profiles/r{sys.argv[1]}_parity_fp16.json", "w"), indent=1)
PY
ls -la ${P}_*
