# mel tests + per-kernel time of the log-mel kernels (rocprofv3 kernel trace of one short bench pass); run through gpurun
R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest $R/tests -m gpu -q -k "mel" < /dev/null 2>&1 | tail -n 3
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_mel -o mel -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > /tmp/b_mel.log 2>&1 < /dev/null
K=$(find /tmp/p_mel -name "*_results.db" | head -n 1)
timeout 60 python $R/tools/prof_summary.py "$K" --grid --csv /tmp/k_mel.csv > /dev/null 2>&1 < /dev/null
grep -h "mel" /tmp/k_mel.csv < /dev/null | cut -c1-200
