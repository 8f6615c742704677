#!/bin/bash
# The round's profile evidence, one call on the GPU box (writes under gpurun_out/<tag>_*; copy what is judged into profiles/):
#   tools/gpu_profile_round.sh r03
# (before a re-run, delete the local copies of gpurun_out/<tag>_{stats,pmc_*} as well: gpurun merges, it does not replace.)
# 1. identity of the kernel sources; 2. rocprofv3 kernel trace + stats of the default bench; 3.-5. PMC passes (FETCH_SIZE,
# WRITE_SIZE, VALU issue), each in its own run with --kernel-trace only; 6. the default bench line with the CPU baseline.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_valu   # the post-processors read every CSV below these
python3 $R/tools/src_id.py > $O/${TAG}_srcid.json
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --pcie-steps 0"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats --output-format csv -- $B --steps 5 --warmup 1 > $O/${TAG}_stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch --output-format csv -- $B --steps 1 --warmup 0 > $O/${TAG}_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write --output-format csv -- $B --steps 1 --warmup 0 > $O/${TAG}_pmcw.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_valu --output-format csv -- $B --steps 1 --warmup 0 > $O/${TAG}_pmcv.log 2>&1
cd $R
python3 bench.py > $O/${TAG}_bench_final.log 2>&1
