#!/bin/bash
# Copy what is kept of one tools/r06_evidence.sh run (gpurun_out/<tag>) into a
# directory under profiles/: the lines, logs and counter summaries as they
# are, the rocprofv3 statistics under the names profiles/r06/README.md uses.
#   tools/install_evidence.sh gpurun_out/r06ev5 profiles/r06
src=$1; dst=$2
mkdir -p $dst
for f in $src/*.json $src/*.log $src/*.txt $src/*.md; do
    case $(basename $f) in pmc_collect.log) continue;; esac
    [ -s $f ] && cp $f $dst/
done
stats() { find $src/$1 -name "*_$2.csv" | head -n 1; }
f=$(stats prof_bench kernel_stats);      [ -n "$f" ] && cp $f $dst/bench_kernel_stats.csv
f=$(stats prof_bench memory_copy_stats); [ -n "$f" ] && cp $f $dst/bench_memory_copy_stats.csv
f=$(stats prof_c5 kernel_stats);         [ -n "$f" ] && cp $f $dst/bench_config5_kernel_stats.csv
f=$(stats prof_c3k kernel_stats);        [ -n "$f" ] && cp $f $dst/bench_c3k_kernel_stats.csv
ls $dst | wc -l
