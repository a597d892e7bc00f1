# round-4 parity soaks (GPU chain - whole steps made natively, bnpc_chain_step -
# vs CPU oracle chain, same seed), run in parallel on the GPU box's host cores;
# logs -> gpurun_out/r04soak (copied to profiles/r04/)
out=gpurun_out/r04soak; mkdir -p $out
python3 tools/parity_soak.py c3 600 42 > $out/soak_c3_600_seed42.log 2>&1 &
python3 tools/parity_soak.py c3 300 11 > $out/soak_c3_300_seed11.log 2>&1 &
python3 tools/parity_soak.py c2 400 7 0.5 > $out/soak_c2_400_seed7_smp05.log 2>&1 &
python3 tools/parity_soak.py c2 400 3 0.5 > $out/soak_c2_400_seed3_smp05.log 2>&1 &
python3 tools/parity_soak.py c4 40 42 > $out/soak_c4_40_seed42.log 2>&1 &
python3 tools/parity_soak.py c3 250 5 > $out/soak_c3_250_seed5.log 2>&1 &
python3 tools/parity_soak.py c3 250 6 0.6 > $out/soak_c3_250_seed6_smp06.log 2>&1 &
python3 tools/parity_soak.py c3 250 7 0.1 > $out/soak_c3_250_seed7_smp01.log 2>&1 &
python3 tools/parity_soak.py c2 600 11 0.7 > $out/soak_c2_600_seed11_smp07.log 2>&1 &
python3 tools/parity_soak.py c2 600 12 0.33 > $out/soak_c2_600_seed12.log 2>&1 &
python3 tools/parity_soak.py c4 30 9 > $out/soak_c4_30_seed9.log 2>&1 &
python3 tools/parity_soak_c5.py 7 8 > $out/soak_c5_full_size_seed7.log 2>&1 &
wait
tail -n 1 $out/*.log
