#!/bin/bash
# Measurement set of a round, run on the GPU box (gpurun): writes everything
# under gpurun_out/$1 (default r02ev2); the summaries that are kept are copied
# into profiles/ by hand afterwards.  rocprofv3 needs TMPDIR=/tmp and the
# program itself after "--" (python3, no wrappers).
out=gpurun_out/${1:-r02ev2}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT

# 1. the bench line, plain and under the kernel / copy trace
python3 bench.py > $out/bench_final.json 2> $out/bench_final.err
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

# 3. the other mapping (lane <-> mutation, wavefront reduction) next to the shipped kernel
[ -x tools/ubench/wavereduce_peak ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/wavereduce_peak tools/ubench/wavereduce_peak.hip
rocprofv3 --kernel-trace --stats -d $out/prof_wavereduce -o wr -f csv -- \
    tools/ubench/wavereduce_peak > $out/wavereduce_peak.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_kernel_ab -o ab -f csv -- \
    python3 tools/kernel_ab.py 5120 1024 3152 > $out/kernel_ab_5120x1024x3152.log 2>&1

# 4. posterior co-clustering kernel, column counts
rocprofv3 --kernel-trace --stats -d $out/prof_codist -o codist -f csv -- \
    python3 tools/codist_bench.py > $out/codist_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_colcounts -o cc -f csv -- \
    python3 tools/colcounts_bench.py > $out/colcounts_bench.log 2>&1

# 5. call overheads, microbench, host breakdown, other configs
python3 tools/call_overhead.py > $out/call_overhead.log 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/profile_steps.py c3 200 > $out/host_step_breakdown_after.log 2>&1
python3 tools/gibbs_bench.py > $out/gibbs_bench.log 2>&1
python3 tools/mh_bench.py > $out/mh_bench.log 2>&1
python3 tools/msplit_tune_big.py > $out/msplit_big.log 2>&1
python3 tools/prefetch_ab.py > $out/prefetch_ab.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof_first_sweep -o fs -f csv -- \
    python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5_rocprof.log 2>&1
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1
python3 tools/first_sweep_profile.py c4 > $out/first_sweep_c4.log 2>&1
for p in pin_probe sync_probe; do
    [ -x tools/ubench/$p ] || hipcc --offload-arch=gfx950 -O2 -pthread -o tools/ubench/$p tools/ubench/$p.hip
    tools/ubench/$p > $out/$p.log 2>&1
done
python3 tools/multichain_bench.py c3 2000 1 2 4 8 > $out/multichain_c3.log 2>&1
python3 tools/multichain_bench.py c2 4000 1 2 4 8 > $out/multichain_c2.log 2>&1
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 0 > $out/bench_config4.json 2> /dev/null
python3 bench.py --config c5 --steps 60 --warmup 5 --cpu-steps 0 > $out/bench_config5.json 2> /dev/null
ls $out
