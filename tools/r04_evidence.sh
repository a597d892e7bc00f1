#!/bin/bash
# Measurement set of round 4, ONE command in ONE lease of a GPU box (gpurun):
# the bench line, the rocprofv3 kernel / copy statistics of the same command
# and the PMC passes whose counters bench.py quotes (pmc_final.json carries
# the digest of the sources it was taken with; bench.py refuses another
# build's).  Writes under gpurun_out/$1 (default r04ev); what is kept is copied
# into profiles/r04 afterwards.  rocprofv3 needs TMPDIR=/tmp and the program
# itself after "--" (python3, no wrappers).
out=gpurun_out/${1:-r04ev}
mkdir -p $out
uptime > $out/box_load.log; nproc >> $out/box_load.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT

# 1. PMC passes first (separate runs: FETCH_SIZE and WRITE_SIZE do not fit
#    together), so that the bench lines below find counters of THIS build
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_$tag -o pmc -f csv -- \
        python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > /dev/null 2> $out/pmc_$tag.err
done
python3 tools/pmc_collect.py $out/pmc_final.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_WAVES > $out/pmc_collect.log 2>&1
mkdir -p profiles/r04 && cp $out/pmc_final.json profiles/r04/pmc_final.json

# 2. the bench line: default team (x3), 4 threads, 1 thread, 20 steps, all
#    fast paths off; under the kernel / copy trace
for i in 1 2 3; do python3 bench.py --cpu-steps $([ $i = 1 ] && echo 12 || echo 0) > $out/bench_final_$i.json 2> $out/bench_final_$i.err; done
cp $out/bench_final_1.json $out/bench_final.json
BNPC_HOST_THREADS=4 python3 bench.py --cpu-steps 0 > $out/bench_threads4.json 2>/dev/null
for i in 1 2 3; do BNPC_HOST_THREADS=1 python3 bench.py --cpu-steps 0 > $out/bench_threads1_$i.json 2>/dev/null; done
python3 bench.py --steps 20 --warmup 10 --cpu-steps 0 > $out/bench_20steps.json 2>/dev/null
BNPC_STREAM_LIVE=0 BNPC_NATIVE_MH=0 BNPC_NATIVE_BETA=0 python3 bench.py --steps 100 --cpu-steps 0 > $out/bench_fallbacks.json 2>/dev/null
BNPC_NATIVE_STEP=0 python3 bench.py --cpu-steps 0 > $out/bench_step_by_methods.json 2>/dev/null
BNPC_DONE_WORDS=0 python3 bench.py --cpu-steps 0 > $out/bench_no_done_words.json 2>/dev/null
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_bench -o bench -f csv -- \
    python3 bench.py --cpu-steps 0 > $out/bench_under_rocprof.json 2> $out/rocprof_bench.err

# 3. host side: interpreter share, screen traces, microbench, first sweep
python3 tools/python_overhead.py c3 300 > $out/python_overhead.log 2>&1
BNPC_NATIVE_STEP=0 python3 tools/python_overhead.py c3 300 > $out/python_overhead_by_method.log 2>&1
python3 tools/mh_dev_trace.py c3 30 > /dev/null 2> $out/mh_screen_trace_c3.log
python3 tools/mh_dev_trace.py c5 20 > /dev/null 2> $out/mh_screen_trace_c5.log
BNPC_TIMING=gibbs,params python3 bench.py --config c5 --steps 20 --warmup 10 --cpu-steps 0 > /dev/null 2> $out/host_phase_trace_c5.log
BNPC_TIMING=gibbs,params python3 bench.py --steps 40 --warmup 10 --cpu-steps 0 > /dev/null 2> $out/host_phase_trace_c3.log
BNPC_TIMING=move python3 bench.py --steps 60 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c3.log
BNPC_TIMING=move python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c5.log
rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o c5 -f csv -- \
    python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 > /dev/null 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/tile_shape_bench.py 50000 5000 31608 > $out/tile_shape_c5.log 2>&1
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_first_sweep -o fs -f csv -- \
    python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5_rocprof.log 2>&1
[ -x tools/ubench/sync_probe ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/sync_probe tools/ubench/sync_probe.hip
tools/ubench/sync_probe > $out/sync_probe.log 2>&1

# 4. the posterior pipeline
BNPC_WARD_DEVICE=plain rocprofv3 --kernel-trace --stats -d $out/prof_posterior -o post -f csv -- \
    python3 tools/posterior_bench.py 10000 400 20 > $out/posterior_bench_10000_rocprof.log 2>&1
WARD_CHECK=1 python3 tools/posterior_bench.py 10000 400 20 > $out/posterior_bench_10000.log 2>&1
python3 tools/posterior_bench.py 50000 200 50 > $out/posterior_bench_50000.log 2>&1

# 5. ranks sharing the one GPU (bench.py's own harness), other configs, the
#    CLI at config-5 size with config 5's own flags
for n in 1 2 4 8; do
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus $n --steps 200 --warmup 10 --cpu-steps 0 2>/dev/null | tail -1
done > $out/bench_ranks_sharing_one_gpu.jsonl
python3 tools/multichain_bench.py c3 2000 1 2 4 8 > $out/multichain_c3.log 2>&1
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 0 > $out/bench_config4.json 2> /dev/null
python3 bench.py --config c5 --steps 60 --warmup 10 --cpu-steps 0 > $out/bench_config5.json 2> /dev/null
E2E_FLAGS="-smp 0.5 -sms 5" python3 tools/e2e_cli_big.py 50000 5000 50 8 200 posterior ML MAP > $out/e2e_cli_config5_posterior.log 2>&1
uptime >> $out/box_load.log
find $out -name "*_trace.csv" -size +4M -delete
ls $out
