# round-5 parity soaks (GPU chain - whole steps made natively - vs CPU oracle
# chain, same seed), run in parallel on the GPU box's host cores; logs ->
# gpurun_out/r05soak (copied to profiles/r05/)
out=gpurun_out/r05soak; mkdir -p $out
python3 tools/parity_soak.py c3 600 42 > $out/soak_c3_600_seed42.log 2>&1 &
python3 tools/parity_soak.py c3 300 11 0.6 > $out/soak_c3_300_seed11_smp06.log 2>&1 &
python3 tools/parity_soak.py c3 300 5 0.1 > $out/soak_c3_300_seed5_smp01.log 2>&1 &
python3 tools/parity_soak.py c2 600 7 0.5 > $out/soak_c2_600_seed7_smp05.log 2>&1 &
python3 tools/parity_soak.py c2 800 105 0.9 > $out/soak_c2_800_seed105_smp09.log 2>&1 &
python3 tools/parity_soak.py k150 300 42 > $out/soak_k150_300_seed42.log 2>&1 &
python3 tools/parity_soak.py k150 200 8 0.6 > $out/soak_k150_200_seed8_smp06.log 2>&1 &
python3 tools/parity_soak.py c3k 40 42 > $out/soak_c3k_40_seed42.log 2>&1 &
python3 tools/parity_soak.py c4 40 42 > $out/soak_c4_40_seed42.log 2>&1 &
python3 tools/parity_soak.py c4 30 9 0.5 > $out/soak_c4_30_seed9_smp05.log 2>&1 &
python3 tools/parity_soak_c5.py 7 8 > $out/soak_c5_full_size_seed7.log 2>&1 &
wait
tail -n 1 $out/*.log
