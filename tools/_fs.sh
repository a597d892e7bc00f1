cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "issued_tiles" 2>&1 | tail -n 2
rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/r02j/fs -o fs -f csv -- python3 tools/first_sweep_profile.py c5 > gpurun_out/r02j/first_sweep_c5.log 2>&1
grep -v "^[EWI]2026" gpurun_out/r02j/first_sweep_c5.log | tail -n 45
python3 - <<'PY'
import csv
for f in ('kernel_stats','memory_copy_stats'):
    for row in csv.DictReader(open(f'gpurun_out/r02j/fs/fs_{f}.csv')):
        print(row['Name'][:50].ljust(50), row['Calls'].rjust(6), f"{float(row['TotalDurationNs'])/1e6:9.2f} ms", f"{float(row['AverageNs'])/1e3:10.1f} us")
PY
