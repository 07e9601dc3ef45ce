#!/bin/bash
# rocprofv3 kernel statistics of the C5 shapes (Gemma-2B dims, S 2048, LoRA r16, batch 8: three steps + two greedy generates, scripts/dev_gemma_step.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python3 scripts/dev_gemma_step.py 8 > gpurun_out/prof_c5.log 2> gpurun_out/prof_c5.err
head -34 gpurun_out/prof_c5/c5_kernel_stats.csv | cut -c1-200
find gpurun_out/prof_c5 -type f ! -name "*stats.csv" -delete
