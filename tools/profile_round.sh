# Round profile on the GPU box: kernel stats of the default bench command, then the PMC passes (each its own run, with
# --kernel-trace only: gpurun refuses counter collection combined with other trace domains).  Outputs under gpurun_out/;
# the summaries are copied into profiles/ by the last lines.
export TMPDIR=/tmp
TAG=${1:-r02}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --pcie-steps 0 > gpurun_out/prof_${TAG}_bench.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pcie-steps 0 > gpurun_out/pmc_fetch_$TAG.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pcie-steps 0 > gpurun_out/pmc_write_$TAG.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d gpurun_out/pmc_valu_$TAG --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pcie-steps 0 > gpurun_out/pmc_valu_$TAG.log 2>&1 &&
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG 128 > gpurun_out/${TAG}_traffic.json &&
python3 tools/pmc_valu.py gpurun_out/pmc_valu_$TAG > gpurun_out/${TAG}_valu.json &&
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats_bench_steps5.csv &&
tail -1 gpurun_out/prof_${TAG}_bench.log | cut -c1-300
