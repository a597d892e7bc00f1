#!/bin/bash
# Round 6, the last lease: the GPU suite, the evidence set and the three soak
# sets on the round's last code.
cd $GRAFT_REPO_ROOT
bash tools/r06_final_check.sh r06final3 > gpurun_out/r06final3.log 2>&1
tail -n 12 gpurun_out/r06final3.log
bash tools/r06_evidence.sh r06ev4 > gpurun_out/r06ev4.log 2>&1
uptime > gpurun_out/r06soak_box_load.log
bash tools/r06_soaks.sh > gpurun_out/r06soak.log 2>&1
bash tools/r06_soaks_more.sh > gpurun_out/r06soak2.log 2>&1
uptime >> gpurun_out/r06soak_box_load.log
bash tools/r06_soaks_long.sh > gpurun_out/r06soak3.log 2>&1
uptime >> gpurun_out/r06soak_box_load.log
tail -q -n 1 gpurun_out/r06soak/*.log gpurun_out/r06soak2/*.log gpurun_out/r06soak3/*.log | cut -c1-140
