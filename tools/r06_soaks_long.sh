# round-6 parity soaks, third set: LONG chains (thousands of steps where the
# CPU oracle allows) - GPU chain vs CPU oracle chain on the same seed, in
# parallel on the GPU box's host cores; logs -> gpurun_out/r06soak3
out=gpurun_out/r06soak3; mkdir -p $out
# (the draws of every parameter batch taken ahead, whatever its size: the
# walker and its discard path under thousands of steps)
export BNPC_MH_AHEAD=2
run() { name=$1; shift; python3 tools/parity_soak.py "$@" > $out/soak_$name.log 2>&1 & }
run c3_1500_seed21                   c3 1500 21 0.33
run c3_1200_seed22_data11            c3 1200 22 0.5 data=11
run c3_1200_seed23_data12_fixed      c3 1200 23 0.33 data=12 learned=0
run c2_6000_seed24                   c2 6000 24 0.33
run c2_6000_seed25_data13_learned    c2 6000 25 0.6 data=13 learned=1
run k150_1500_seed26                 k150 1500 26 0.33
run k150_1200_seed27_data14          k150 1200 27 0.5 data=14
run c4_300_seed28                    c4 300 28 0.33
run c3k_200_seed29                   c3k 200 29 0.33
python3 tools/parity_soak_c5.py 23 12 > $out/soak_c5_full_size_seed23_12steps.log 2>&1 &
wait
tail -n 1 $out/*.log
