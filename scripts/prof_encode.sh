#!/bin/bash
# rocprofv3 passes for the encode kernel of the bench workload (profiles/rNN): kernel stats of the bench command, then the HBM and SQ
# counters in passes of their own (--pmc never together with --stats traces beyond --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ECGB_BENCH_WORKERS=1   # no fork pool under the profiler
P=${1:-r2}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${P}_stats -o enc -- python3 bench.py --no-cpu-baseline --no-train --no-c1 --no-extras > gpurun_out/${P}_bench_encode.txt 2> gpurun_out/${P}_bench_encode.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${P}_FETCH_SIZE -- python3 scripts/dev_encode_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${P}_WRITE_SIZE -- python3 scripts/dev_encode_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${P}_sq1 -- python3 scripts/dev_encode_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${P}_sq2 -- python3 scripts/dev_encode_only.py > /dev/null 2>&1
python3 scripts/pmc_summary.py encode_flow_kernel gpurun_out/${P}_FETCH_SIZE gpurun_out/${P}_WRITE_SIZE > gpurun_out/${P}_hbm.json
python3 scripts/pmc_summary.py encode_flow_kernel gpurun_out/${P}_sq1 gpurun_out/${P}_sq2 > gpurun_out/${P}_sq.json
head -4 gpurun_out/${P}_stats/*kernel_stats.csv | cut -c1-200
cat gpurun_out/${P}_hbm.json | head -30
# keep the summaries only
find gpurun_out/${P}_FETCH_SIZE gpurun_out/${P}_WRITE_SIZE gpurun_out/${P}_sq1 gpurun_out/${P}_sq2 -type f -delete
find gpurun_out/${P}_stats -type f ! -name "*stats.csv" -delete
