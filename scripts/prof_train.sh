#!/bin/bash
# rocprofv3 kernel statistics of the two train legs, one trace per mode (profiles/rNN/bench_kernel_stats_{full,lora}.csv)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ECGB_BENCH_WORKERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -o full -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c1 --no-c5 --no-lora-leg --no-extras --train-steps 3 > gpurun_out/prof_full.json 2> gpurun_out/prof_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lora -o lora -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c1 --no-c5 --lora --no-extras --train-steps 3 > gpurun_out/prof_lora.json 2> gpurun_out/prof_lora.err
find gpurun_out/prof_full gpurun_out/prof_lora -name "*kernel_stats.csv" | head
for f in $(find gpurun_out/prof_full gpurun_out/prof_lora -name "*kernel_stats.csv"); do echo == $f; head -25 $f | cut -c1-200; done
# keep only the stats (the traces are large)
find gpurun_out/prof_full gpurun_out/prof_lora -type f ! -name "*stats.csv" -delete
