# round-5 parity soaks, third set: LONG chains (thousands of steps where the
# CPU oracle allows) - GPU chain vs CPU oracle chain on the same seed, in
# parallel on the GPU box's host cores; logs -> gpurun_out/r05soak3
out=gpurun_out/r05soak3; mkdir -p $out
run() { name=$1; shift; python3 tools/parity_soak.py "$@" > $out/soak_$name.log 2>&1 & }
run c3_1500_seed1                   c3 1500 1 0.33
run c3_1200_seed2_data11            c3 1200 2 0.5 data=11
run c3_1200_seed3_data12_fixed      c3 1200 3 0.33 data=12 learned=0
run c2_6000_seed4                   c2 6000 4 0.33
run c2_6000_seed5_data13_learned    c2 6000 5 0.6 data=13 learned=1
run k150_1500_seed6                 k150 1500 6 0.33
run k150_1200_seed7_data14          k150 1200 7 0.5 data=14
run c4_300_seed8                    c4 300 8 0.33
run c3k_200_seed9                   c3k 200 9 0.33
python3 tools/parity_soak_c5.py 13 12 > $out/soak_c5_full_size_seed13_12steps.log 2>&1 &
wait
tail -n 1 $out/*.log
