out=gpurun_out/r03ev7; mkdir -p $out
for i in 1 2 3; do python3 bench.py > $out/bench_final_$i.json 2>/dev/null; done
for i in 1 2; do BNPC_HOST_THREADS=4 python3 bench.py --cpu-steps 0 > $out/bench_threads4_$i.json 2>/dev/null
BNPC_HOST_THREADS=1 python3 bench.py --cpu-steps 0 > $out/bench_threads1_$i.json 2>/dev/null; done
uptime
