set -x
mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tables or rg_init or tiling or split_merge or new_cluster" 2>&1 | tail -3
for v in "2 1" "2 0" "1 0"; do set -- $v; echo "== BNPC_SEQ_KERNEL=$1 BNPC_SEQ_STAGE=$2"; BNPC_SEQ_KERNEL=$1 BNPC_SEQ_STAGE=$2 python tools/call_overhead.py 2>&1 | grep -i "tables\|k_ll_seq"; done > gpurun_out/r02f/seqp_ab.log 2>&1
cat gpurun_out/r02f/seqp_ab.log
./tools/ubench/bar_write > gpurun_out/r02f/bar_write.log 2>&1; cat gpurun_out/r02f/bar_write.log
