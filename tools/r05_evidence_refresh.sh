#!/bin/bash
# After a change of the library's sources: the part of tools/r05_evidence.sh
# that is tied to the build by its digest - the PMC passes - and the lines
# that quote them (config 3 x 3, config 5), plus the method-by-method line.
out=gpurun_out/${1:-r05ev2}
mkdir -p $out profiles/r05
uptime > $out/box_load.log; nproc >> $out/box_load.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_$tag -o pmc -f csv -- \
        python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > /dev/null 2> $out/pmc_$tag.err
done
python3 tools/pmc_collect.py $out/pmc_final.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_WAVES > $out/pmc_collect.log 2>&1
cp $out/pmc_final.json profiles/r05/pmc_final.json
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
for i in 1 2 3; do python3 bench.py --cpu-steps $([ $i = 1 ] && echo 12 || echo 0) > $out/bench_final_$i.json 2> $out/bench_final_$i.err; done
cp $out/bench_final_1.json $out/bench_final.json
BNPC_NATIVE_STEP=0 python3 bench.py --cpu-steps 0 > $out/bench_step_by_methods.json 2>/dev/null
python3 bench.py --config c5 --steps 60 --warmup 10 --cpu-steps 2 --cpu-seconds 400 > $out/bench_config5.json 2> $out/bench_config5.err
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_bench -o bench -f csv -- \
    python3 bench.py --cpu-steps 0 > $out/bench_under_rocprof.json 2> $out/rocprof_bench.err
uptime >> $out/box_load.log
find $out -name "*_trace.csv" -size +4M -delete
ls $out
