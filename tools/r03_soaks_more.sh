out=gpurun_out/r03soak2; mkdir -p $out
python3 tools/parity_soak.py c3 250 5 > $out/soak_c3_250_seed5.log 2>&1 &
python3 tools/parity_soak.py c3 250 6 0.6 > $out/soak_c3_250_seed6_smp06.log 2>&1 &
python3 tools/parity_soak.py c3 250 7 0.1 > $out/soak_c3_250_seed7_smp01.log 2>&1 &
python3 tools/parity_soak.py c2 600 11 0.7 > $out/soak_c2_600_seed11_smp07.log 2>&1 &
python3 tools/parity_soak.py c2 600 12 0.33 > $out/soak_c2_600_seed12.log 2>&1 &
python3 tools/parity_soak.py c4 30 9 > $out/soak_c4_30_seed9.log 2>&1 &
wait
tail -n 1 $out/*.log
