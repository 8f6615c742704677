# stream / chunk / batch sweep of bench.py on one GPU (results: gpurun_out/sweep_r02.log)
rm -f gpurun_out/sweep_r02.log
for cfg in "2 128 256" "3 128 384" "2 192 384" "2 256 512" "3 96 288" "4 64 256" "1 256 256"; do
  set -- $cfg
  echo "streams=$1 chunk=$2 batch=$3" >> gpurun_out/sweep_r02.log
  P2AES_STREAMS=$1 P2AES_CHUNK=$2 timeout -k 10 300 python3 bench.py --batch $3 --steps 4 --warmup 1 --no-cpu-baseline --pcie-steps 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep_r02.log
done
cat gpurun_out/sweep_r02.log
