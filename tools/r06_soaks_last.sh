# Round 6, the last code (the large traces written by the team): ten more
# GPU-vs-oracle chains on new seeds, sized for one short lease; logs ->
# gpurun_out/r06soak4 (copied to profiles/r06/soak4_*.log)
out=gpurun_out/r06soak4; mkdir -p $out
uptime > $out/box_load.log
python3 tools/parity_soak.py c3 400 301 > $out/soak_c3_400_seed301.log 2>&1 &
python3 tools/parity_soak.py c3 300 302 0.7 > $out/soak_c3_300_seed302_smp07.log 2>&1 &
python3 tools/parity_soak.py c2 1500 303 0.5 > $out/soak_c2_1500_seed303_smp05.log 2>&1 &
python3 tools/parity_soak.py c2 1500 304 0.9 > $out/soak_c2_1500_seed304_smp09.log 2>&1 &
python3 tools/parity_soak.py k150 500 305 > $out/soak_k150_500_seed305.log 2>&1 &
python3 tools/parity_soak.py k150 400 306 0.7 > $out/soak_k150_400_seed306_smp07.log 2>&1 &
python3 tools/parity_soak.py c3k 60 307 > $out/soak_c3k_60_seed307.log 2>&1 &
python3 tools/parity_soak.py c4 60 308 > $out/soak_c4_60_seed308.log 2>&1 &
python3 tools/parity_soak.py c4 40 309 0.6 > $out/soak_c4_40_seed309_smp06.log 2>&1 &
python3 tools/parity_soak_c5.py 310 4 > $out/soak_c5_full_size_seed310.log 2>&1 &
wait
uptime >> $out/box_load.log
tail -q -n 1 $out/soak_*.log | cut -c1-170
