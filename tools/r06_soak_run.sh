#!/bin/bash
# Round 6: both soak sets in one lease (GPU chain vs CPU oracle chain).
cd $GRAFT_REPO_ROOT
uptime > gpurun_out/r06soak_box_load.log
bash tools/r06_soaks.sh > gpurun_out/r06soak.log 2>&1
uptime >> gpurun_out/r06soak_box_load.log
bash tools/r06_soaks_long.sh > gpurun_out/r06soak3.log 2>&1
uptime >> gpurun_out/r06soak_box_load.log
tail -n 1 gpurun_out/r06soak/*.log gpurun_out/r06soak3/*.log
grep -l "first diverging" gpurun_out/r06soak/*.log gpurun_out/r06soak3/*.log
