# round-6 parity soaks (GPU chain - whole steps made natively - vs CPU oracle
# chain, same seed), run in parallel on the GPU box's host cores; logs ->
# gpurun_out/r06soak (copied to profiles/r06/)
out=gpurun_out/r06soak; mkdir -p $out
python3 tools/parity_soak.py c3 600 142 > $out/soak_c3_600_seed142.log 2>&1 &
python3 tools/parity_soak.py c3 300 111 0.6 > $out/soak_c3_300_seed111_smp06.log 2>&1 &
python3 tools/parity_soak.py c3 300 15 0.1 > $out/soak_c3_300_seed15_smp01.log 2>&1 &
python3 tools/parity_soak.py c2 600 17 0.5 > $out/soak_c2_600_seed17_smp05.log 2>&1 &
python3 tools/parity_soak.py c2 800 205 0.9 > $out/soak_c2_800_seed205_smp09.log 2>&1 &
python3 tools/parity_soak.py k150 300 52 > $out/soak_k150_300_seed52.log 2>&1 &
python3 tools/parity_soak.py k150 200 18 0.6 > $out/soak_k150_200_seed18_smp06.log 2>&1 &
python3 tools/parity_soak.py c3k 40 52 > $out/soak_c3k_40_seed52.log 2>&1 &
python3 tools/parity_soak.py c4 40 52 > $out/soak_c4_40_seed52.log 2>&1 &
python3 tools/parity_soak.py c4 30 19 0.5 > $out/soak_c4_30_seed19_smp05.log 2>&1 &
python3 tools/parity_soak_c5.py 17 8 > $out/soak_c5_full_size_seed17.log 2>&1 &
wait
tail -n 1 $out/*.log
