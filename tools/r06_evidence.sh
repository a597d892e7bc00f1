#!/bin/bash
# Measurement set of round 6, ONE command in ONE lease of a GPU box (gpurun):
# PMC passes (config 3's bench run; config 5's converged shape on its own),
# the bench lines of every config with their parity gates and CPU legs, the
# A/B lines of the round's changes (BNPC_MH_AHEAD=0, BNPC_SWEEP_LANE=nostride),
# rocprofv3 kernel / copy statistics of the bench command, phase traces, the
# `--gpus N` lines (the command starts its ranks itself).  Writes under
# gpurun_out/$1 (default r06ev); what is kept is copied into profiles/r06
# afterwards.  rocprofv3 needs TMPDIR=/tmp and the program itself after "--".
out=gpurun_out/${1:-r06ev}
mkdir -p $out
uptime > $out/box_load.log; nproc >> $out/box_load.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT

# 1. PMC passes first, so that the bench lines below find counters of THIS build
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_$tag -o pmc -f csv -- \
        python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > /dev/null 2> $out/pmc_$tag.err
done
python3 tools/pmc_collect.py $out/pmc_final.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_WAVES > $out/pmc_collect.log 2>&1
mkdir -p profiles/r06; cp $out/pmc_final.json profiles/r06/pmc_final.json
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
    "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_c5_$tag -o pmc -f csv -- \
        python3 tools/ll_shape_run.py 50000 5000 54 5 > /dev/null 2> $out/pmc_c5_$tag.err
done
PMC_COMMAND="rocprofv3 --pmc <counter> -- python3 tools/ll_shape_run.py 50000 5000 54 5 (one pass per counter set)" \
    python3 tools/pmc_collect.py $out/pmc_c5.json $out/pmc_c5_FETCH_SIZE $out/pmc_c5_WRITE_SIZE $out/pmc_c5_SQ_WAVES $out/pmc_c5_TCC_HIT_sum >> $out/pmc_collect.log 2>&1
cp $out/pmc_c5.json profiles/r06/pmc_c5.json

# 2. config 3: the line as the driver runs it, the default line x3 (the first
#    with its CPU leg and parity gate), 1 thread, A/Bs, fallbacks; kernel trace
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
for i in 1 2 3; do python3 bench.py --cpu-steps $([ $i = 1 ] && echo 12 || echo 0) > $out/bench_final_$i.json 2> $out/bench_final_$i.err; done
cp $out/bench_final_1.json $out/bench_final.json
nodev="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do BNPC_HOST_THREADS=1 python3 bench.py $nodev > $out/bench_threads1_$i.json 2>/dev/null; done
BNPC_STREAM_LIVE=0 BNPC_NATIVE_MH=0 BNPC_NATIVE_BETA=0 python3 bench.py --steps 100 $nodev > $out/bench_fallbacks.json 2>/dev/null
BNPC_NATIVE_STEP=0 python3 bench.py $nodev > $out/bench_step_by_methods.json 2>/dev/null
BNPC_SWEEP_LANE=nostride python3 bench.py $nodev > $out/bench_without_the_stride.json 2>/dev/null
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_bench -o bench -f csv -- \
    python3 bench.py $nodev > $out/bench_under_rocprof.json 2> $out/rocprof_bench.err

# 3. the other configs, each with a bounded CPU leg and its parity gate
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c3k --steps 100 --cpu-steps 3 > $out/bench_c3k.json 2> /dev/null
python3 bench.py --config k150 --steps 200 --cpu-steps 6 > $out/bench_k150.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 4 > $out/bench_config4.json 2> /dev/null
python3 bench.py --config c5 --steps 100 --warmup 10 --cpu-steps 2 --cpu-seconds 400 > $out/bench_config5.json 2> $out/bench_config5.err
#    ... and three more of config 5 / 4 / c3k as the target is quoted (median
#    of 3), interleaved with the round's two host-side changes switched off
for i in 1 2 3; do
    python3 bench.py --config c5 --steps 100 $nodev > $out/bench_c5_$i.json 2>/dev/null
    BNPC_MH_AHEAD=0 python3 bench.py --config c5 --steps 100 $nodev > $out/bench_c5_ahead0_$i.json 2>/dev/null
    BNPC_SWEEP_LANE=nostride python3 bench.py --config c5 --steps 100 $nodev > $out/bench_c5_nostride_$i.json 2>/dev/null
    python3 bench.py --config c4 --steps 100 $nodev > $out/bench_c4_$i.json 2>/dev/null
    python3 bench.py --config c3k --steps 100 $nodev > $out/bench_c3k_$i.json 2>/dev/null
done

# 4. host side: phase traces over the bench window, moves, microbench
python3 tools/python_overhead.py c3 300 > $out/python_overhead.log 2>&1
for c in c3 c3k c4 c5; do
    steps=100; [ $c = c3 ] && steps=200
    BNPC_TIMING=gibbs,params python3 bench.py --config $c --steps $steps --warmup 10 $nodev > $out/bench_traced_$c.json 2> $out/host_phase_trace_$c.log
    python3 tools/trace_means.py $out/host_phase_trace_$c.log $out/bench_traced_$c.json > $out/host_phase_means_$c.txt 2>&1
done
python3 tools/mh_dev_trace.py c5 12 > /dev/null 2> $out/mh_screen_trace_c5.log
BNPC_TIMING=move python3 bench.py --steps 60 --warmup 10 $nodev 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c3.log
BNPC_TIMING=move python3 bench.py --config c5 --steps 30 --warmup 10 $nodev 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c5.log
rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o c5 -f csv -- \
    python3 bench.py --config c5 --steps 30 --warmup 10 $nodev > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_c3k -o c3k -f csv -- \
    python3 bench.py --config c3k --steps 50 --warmup 10 $nodev > /dev/null 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1
for lane in on nostride; do
  export BNPC_SWEEP_LANE=$lane; [ $lane = on ] && unset BNPC_SWEEP_LANE
  echo "== lane $lane" >> $out/hinted_loop_bench.log
  python3 tools/hinted_loop_bench.py 5000 14 400 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 5000 14 400 0.25 0.08 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 10000 20 300 0.2 0.02 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 50000 50 100 0.05 0.003 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 5000 200 300 0.02 0.0 >> $out/hinted_loop_bench.log 2>&1
done
unset BNPC_SWEEP_LANE
uptime >> $out/box_load.log
find $out -name "*_trace.csv" -size +4M -delete
ls $out
