# Round profile on the GPU box: kernel stats of the default bench command, then the two PMC passes (separate runs, no
# other trace domains).  Outputs under gpurun_out/; copy the summaries into profiles/.
export TMPDIR=/tmp
TAG=${1:-r01_d}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/prof_${TAG}_bench.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_fetch_$TAG.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_write_$TAG.log 2>&1 &&
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG 128 > gpurun_out/traffic_$TAG.json &&
tail -1 gpurun_out/prof_${TAG}_bench.log && ls gpurun_out/prof_$TAG/*/ | head
