# round-3 parity soaks (GPU chain vs CPU oracle chain, same seed), run in
# parallel on the GPU box's host cores; logs -> gpurun_out/r03soak
out=gpurun_out/r03soak; mkdir -p $out
python3 tools/parity_soak.py c3 600 42 > $out/soak_c3_600_seed42.log 2>&1 &
python3 tools/parity_soak.py c3 300 11 > $out/soak_c3_300_seed11.log 2>&1 &
python3 tools/parity_soak.py c2 400 7 0.5 > $out/soak_c2_400_seed7_smp05.log 2>&1 &
python3 tools/parity_soak.py c2 400 3 0.5 > $out/soak_c2_400_seed3_smp05.log 2>&1 &
python3 tools/parity_soak.py c4 40 42 > $out/soak_c4_40_seed42.log 2>&1 &
wait
tail -n 2 $out/*.log
