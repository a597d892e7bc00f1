# round-5 parity soaks, second set: other synthetic matrices, parameter priors
# (uniform Beta(1, 1), skewed), fixed / learned error rates, other split-merge
# settings and DP-alpha priors - GPU chain vs CPU oracle chain on the same
# seed, in parallel on the GPU box's host cores; logs -> gpurun_out/r05soak2
out=gpurun_out/r05soak2; mkdir -p $out
run() { name=$1; shift; python3 tools/parity_soak.py "$@" > $out/soak_$name.log 2>&1 & }
run c3_data1_uniform_prior          c3 300 21 0.33 data=1 beta=1,1
run c3_data2_fixed_errors           c3 300 22 0.5 data=2 learned=0
run c3_data3_skewed_prior_5scans    c3 250 23 0.4 data=3 beta=2,0.5 sm_steps=5 ratios=.5,.5
run c3_data4_alpha_prior            c3 250 24 0.33 data=4 alpha=10,2
run c2_data5_uniform_prior_learned  c2 600 25 0.7 data=5 beta=1,1 learned=1
run c2_data6_merge_heavy            c2 600 26 0.9 data=6 ratios=.2,.8 sm_steps=1
run k150_data7_uniform_prior        k150 200 27 0.33 data=7 beta=1,1
run k150_data8_learned              k150 150 28 0.5 data=8 learned=1
run c3k_data9                       c3k 30 29 0.33 data=9
run c4_data10_uniform_prior         c4 30 30 0.5 data=10 beta=1,1
wait
tail -n 1 $out/*.log
