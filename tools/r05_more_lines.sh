out=gpurun_out/r05ev8; mkdir -p $out
uptime > $out/box_load.log
for i in 1 2 3; do
  python3 bench.py --config c5 --steps 60 --warmup 10 --cpu-steps 0 > $out/bench_config5_$i.json 2>/dev/null
  python3 bench.py --config c4 --steps 100 --cpu-steps 0 > $out/bench_config4_$i.json 2>/dev/null
  python3 bench.py --config c3k --steps 100 --cpu-steps 0 > $out/bench_c3k_$i.json 2>/dev/null
  python3 bench.py --config k150 --steps 200 --cpu-steps 0 > $out/bench_k150_$i.json 2>/dev/null
  python3 bench.py --cpu-steps 0 > $out/bench_c3_$i.json 2>/dev/null
done
python3 bench.py --steps 20 --warmup 10 --cpu-steps 0 > $out/bench_20steps.json 2>/dev/null
for n in 1 2 4 8; do
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus $n --steps 200 --warmup 10 --cpu-steps 0 2>/dev/null | tail -1
done > $out/bench_ranks_sharing_one_gpu.jsonl
uptime >> $out/box_load.log
for f in $out/bench_*.json; do python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j['value'], j.get('first_step_s'))"; done
python3 -c "
import json
for l in open('$out/bench_ranks_sharing_one_gpu.jsonl'):
    j = json.loads(l); print(j['ranks'], j['n_gpus'], j['value'])"
cat $out/box_load.log
