out=gpurun_out/r05pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for shape in "50000 5000 54" "50000 5000 50"; do
tag=$(echo $shape | tr ' ' 'x')
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr -d $out/pmc_${tag}_$t -o pmc -f csv -- python3 tools/ll_shape_run.py $shape 5 > /dev/null 2> $out/pmc_${tag}_$t.err
done
PMC_COMMAND="rocprofv3 --pmc <counter> -- python3 tools/ll_shape_run.py $shape 5 (one pass per counter set)" python3 tools/pmc_collect.py $out/pmc_shape_$tag.json $out/pmc_${tag}_FETCH_SIZE $out/pmc_${tag}_WRITE_SIZE $out/pmc_${tag}_SQ_WAVES $out/pmc_${tag}_TCC_HIT_sum
done
rocprofv3 --kernel-trace --stats -d $out/trace_shape -o t -f csv -- python3 tools/ll_shape_run.py 50000 5000 54 5 > /dev/null 2>&1
find $out -name "*_trace.csv" -size +4M -delete
python3 - <<'PY'
import json
for tag in ('50000x5000x54','50000x5000x50'):
    d=json.load(open(f'gpurun_out/r05pmc/pmc_shape_{tag}.json'))
    for k,v in d.items():
        if k.startswith('_'): continue
        if 'k_ll8' in k or 'combine' in k or 'tables' in k:
            print(tag, k[:60], {c:round(x['mean'],1) for c,x in v.items()})
PY
