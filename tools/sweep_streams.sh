# stream / chunk / batch sweep of bench.py on one GPU (results: gpurun_out/sweep.log)
for cfg in "2 64 128" "2 96 192" "2 128 256" "1 128 128" "3 64 192" "2 48 96"; do
  set -- $cfg
  echo "streams=$1 chunk=$2 batch=$3" >> gpurun_out/sweep.log
  P2AES_STREAMS=$1 P2AES_CHUNK=$2 timeout -k 10 300 python3 bench.py --batch $3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
