# a second set of round-4 parity soaks (other seeds and move mixes than
# tools/r04_soaks.sh); logs -> gpurun_out/r04soak2 (copied to profiles/r04/)
out=gpurun_out/r04soak2; mkdir -p $out
python3 tools/parity_soak.py c3 400 101 > $out/soak2_c3_400_seed101.log 2>&1 &
python3 tools/parity_soak.py c3 300 102 0.5 > $out/soak2_c3_300_seed102_smp05.log 2>&1 &
python3 tools/parity_soak.py c3 300 103 0.8 > $out/soak2_c3_300_seed103_smp08.log 2>&1 &
python3 tools/parity_soak.py c2 800 104 0.33 > $out/soak2_c2_800_seed104.log 2>&1 &
python3 tools/parity_soak.py c2 800 105 0.9 > $out/soak2_c2_800_seed105_smp09.log 2>&1 &
python3 tools/parity_soak.py c4 40 106 0.5 > $out/soak2_c4_40_seed106_smp05.log 2>&1 &
python3 tools/parity_soak_c5.py 11 8 > $out/soak2_c5_full_size_seed11.log 2>&1 &
wait
tail -n 1 $out/*.log
