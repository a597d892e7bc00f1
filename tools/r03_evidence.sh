#!/bin/bash
# Measurement set of round 3, run on the GPU box (gpurun): writes under
# gpurun_out/$1 (default r03ev); the summaries that are kept are copied into
# profiles/r03 afterwards.  rocprofv3 needs TMPDIR=/tmp and the program itself
# after "--" (python3, no wrappers).
out=gpurun_out/${1:-r03ev}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT

# 1. the bench line: default team, 4 threads, 1 thread; under the kernel trace
python3 bench.py > $out/bench_final.json 2> $out/bench_final.err
BNPC_HOST_THREADS=4 python3 bench.py --cpu-steps 0 > $out/bench_threads4.json 2>/dev/null
BNPC_HOST_THREADS=1 python3 bench.py --cpu-steps 0 > $out/bench_threads1.json 2>/dev/null
python3 bench.py --steps 20 --warmup 10 --cpu-steps 0 > $out/bench_20steps.json 2>/dev/null
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_bench -o bench -f csv -- \
    python3 bench.py --cpu-steps 0 > $out/bench_under_rocprof.json 2> $out/rocprof_bench.err

# 2. PMC passes (separate runs: FETCH_SIZE and WRITE_SIZE do not fit together)
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_$tag -o pmc -f csv -- \
        python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > /dev/null 2> $out/pmc_$tag.err
done
python3 tools/pmc_collect.py $out/pmc_final.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_WAVES > $out/pmc_collect.log 2>&1
# the same launch with the L2 prefetch off: its scalar streams alone (the
# calibration of the doubled FETCH_SIZE)
BNPC_LL_PREFETCH=0 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_FETCH_nopf -o pmc -f csv -- \
    python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > /dev/null 2> $out/pmc_FETCH_nopf.err
python3 tools/pmc_collect.py $out/pmc_noprefetch.json $out/pmc_FETCH_nopf $out/pmc_WRITE_SIZE $out/pmc_SQ_WAVES > /dev/null 2>&1

# 3. host breakdown, microbench, call overheads, screen trace
BNPC_HOST_THREADS=1 python3 tools/profile_steps.py c3 200 > $out/host_step_breakdown_threads1.log 2>&1
python3 tools/profile_steps.py c3 200 > $out/host_step_breakdown_default.log 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/call_overhead.py > $out/call_overhead.log 2>&1
python3 tools/mh_dev_trace.py c3 30 > /dev/null 2> $out/mh_screen_trace_c3.log
tools/ubench/sync_probe > $out/sync_probe.log 2>&1

# 4. the posterior pipeline
rocprofv3 --kernel-trace --stats -d $out/prof_posterior -o post -f csv -- \
    python3 tools/posterior_bench.py 10000 400 20 > $out/posterior_bench_10000.log 2>&1
python3 tools/posterior_bench.py 5000 400 10 x > $out/posterior_bench_5000.log 2>&1

# 5. ranks sharing the one GPU (bench.py's own harness), other configs
for n in 1 2 4 8; do
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus $n --steps 200 --warmup 10 --cpu-steps 0 2>/dev/null | tail -1
done > $out/bench_ranks_sharing_one_gpu.jsonl
for n in 8; do
    BNPC_HOST_THREADS=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port 29534 bench.py --gpus $n --steps 200 --warmup 10 --cpu-steps 0 2>/dev/null | tail -1
done > $out/bench_8ranks_1thread_each.jsonl
python3 tools/multichain_bench.py c3 2000 1 2 4 8 > $out/multichain_c3.log 2>&1
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 0 > $out/bench_config4.json 2> /dev/null
python3 bench.py --config c5 --steps 60 --warmup 5 --cpu-steps 0 > $out/bench_config5.json 2> /dev/null
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1
ls $out
