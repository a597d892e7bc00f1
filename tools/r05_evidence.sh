#!/bin/bash
# Measurement set of round 5, ONE command in ONE lease of a GPU box (gpurun):
# PMC passes (config 3's bench run; config 5's converged shape on its own),
# the bench lines of every config with their parity gates and CPU legs, the
# rocprofv3 kernel / copy statistics of the bench command, phase traces.
# Writes under gpurun_out/$1 (default r05ev); what is kept is copied into
# profiles/r05 afterwards.  rocprofv3 needs TMPDIR=/tmp and the program itself
# after "--" (python3, no wrappers).
out=gpurun_out/${1:-r05ev}
mkdir -p $out profiles/r05
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
cp $out/pmc_final.json profiles/r05/pmc_final.json
#    config 5's converged shape on its own (the bench run of config 5 mixes it
#    with the small launches of the moves): K_end of the bench chain = 54
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
    "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_c5_$tag -o pmc -f csv -- \
        python3 tools/ll_shape_run.py 50000 5000 54 5 > /dev/null 2> $out/pmc_c5_$tag.err
done
PMC_COMMAND="rocprofv3 --pmc <counter> -- python3 tools/ll_shape_run.py 50000 5000 54 5 (one pass per counter set)" \
    python3 tools/pmc_collect.py $out/pmc_c5.json $out/pmc_c5_FETCH_SIZE $out/pmc_c5_WRITE_SIZE $out/pmc_c5_SQ_WAVES $out/pmc_c5_TCC_HIT_sum >> $out/pmc_collect.log 2>&1
cp $out/pmc_c5.json profiles/r05/pmc_c5.json

# 2. the bench line of config 3: default team (x3, the first with its CPU leg
#    and parity gate), 1 thread, 20 steps, fallbacks; under the kernel trace
for i in 1 2 3; do python3 bench.py --cpu-steps $([ $i = 1 ] && echo 12 || echo 0) > $out/bench_final_$i.json 2> $out/bench_final_$i.err; done
cp $out/bench_final_1.json $out/bench_final.json
for i in 1 2 3; do BNPC_HOST_THREADS=1 python3 bench.py --cpu-steps 0 > $out/bench_threads1_$i.json 2>/dev/null; done
python3 bench.py --steps 20 --warmup 10 --cpu-steps 0 > $out/bench_20steps.json 2>/dev/null
BNPC_STREAM_LIVE=0 BNPC_NATIVE_MH=0 BNPC_NATIVE_BETA=0 python3 bench.py --steps 100 --cpu-steps 0 > $out/bench_fallbacks.json 2>/dev/null
BNPC_NATIVE_STEP=0 python3 bench.py --cpu-steps 0 > $out/bench_step_by_methods.json 2>/dev/null
BNPC_MH_SCREEN=2 python3 bench.py --cpu-steps 0 > $out/bench_screen_without_theta.json 2>/dev/null
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_bench -o bench -f csv -- \
    python3 bench.py --cpu-steps 0 > $out/bench_under_rocprof.json 2> $out/rocprof_bench.err

# 3. the other configs, each with a bounded CPU leg and its parity gate
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c3k --steps 100 --cpu-steps 3 > $out/bench_c3k.json 2> /dev/null
python3 bench.py --config k150 --steps 200 --cpu-steps 6 > $out/bench_k150.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 4 > $out/bench_config4.json 2> /dev/null
python3 bench.py --config c5 --steps 60 --warmup 10 --cpu-steps 2 --cpu-seconds 400 > $out/bench_config5.json 2> $out/bench_config5.err

# 4. host side: phase traces, screen timelines, moves, microbench, first sweep
python3 tools/python_overhead.py c3 300 > $out/python_overhead.log 2>&1
for c in c3 c3k c4 c5; do
    BNPC_TIMING=gibbs,params python3 bench.py --config $c --steps 20 --warmup 6 --cpu-steps 0 > /dev/null 2> $out/host_phase_trace_$c.log
done
python3 tools/mh_dev_trace.py c3k 12 > /dev/null 2> $out/mh_screen_trace_c3k.log
python3 tools/mh_dev_trace.py c5 12 > /dev/null 2> $out/mh_screen_trace_c5.log
BNPC_TIMING=move python3 bench.py --steps 60 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c3.log
BNPC_TIMING=move python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c5.log
rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o c5 -f csv -- \
    python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_c3k -o c3k -f csv -- \
    python3 bench.py --config c3k --steps 50 --warmup 10 --cpu-steps 0 > /dev/null 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/msplit_sweep.py > $out/msplit_sweep.md 2>&1
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1

# 5. ranks sharing the one GPU (bench.py's own harness: n_gpus says 1)
for n in 1 2 4 8; do
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus $n --steps 200 --warmup 10 --cpu-steps 0 2>/dev/null | tail -1
done > $out/bench_ranks_sharing_one_gpu.jsonl
uptime >> $out/box_load.log
find $out -name "*_trace.csv" -size +4M -delete
ls $out
