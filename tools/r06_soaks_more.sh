# round-6 parity soaks, second set: other synthetic matrices, parameter priors
# (uniform Beta(1, 1), skewed), fixed / learned error rates, other split-merge
# settings and DP-alpha priors - GPU chain vs CPU oracle chain on the same
# seed, in parallel on the GPU box's host cores; logs -> gpurun_out/r06soak2
out=gpurun_out/r06soak2; mkdir -p $out
run() { name=$1; shift; python3 tools/parity_soak.py "$@" > $out/soak_$name.log 2>&1 & }
run c3_data21_uniform_prior          c3 300 121 0.33 data=21 beta=1,1
run c3_data22_fixed_errors           c3 300 122 0.5 data=22 learned=0
run c3_data23_skewed_prior_5scans    c3 250 123 0.4 data=23 beta=2,0.5 sm_steps=5 ratios=.5,.5
run c3_data24_alpha_prior            c3 250 124 0.33 data=24 alpha=10,2
run c2_data25_uniform_prior_learned  c2 600 125 0.7 data=25 beta=1,1 learned=1
run c2_data26_merge_heavy            c2 600 126 0.9 data=26 ratios=.2,.8 sm_steps=1
run k150_data27_uniform_prior        k150 200 127 0.33 data=27 beta=1,1
run k150_data28_learned              k150 150 128 0.5 data=28 learned=1
run c3k_data29                       c3k 30 129 0.33 data=29
run c4_data30_uniform_prior         c4 30 130 0.5 data=30 beta=1,1
wait
tail -n 1 $out/*.log
