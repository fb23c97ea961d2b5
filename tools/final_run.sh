#!/bin/bash
# end of round 4, one lease: the proof (driver's bench command, GPU suite x5, thread-churn test x30) and then the profile set r04_c
bash tools/proof_run.sh r04_final 5 30
bash tools/collect_profiles.sh r04_c > gpurun_out/r04_c_collect.log 2>&1
tail -3 gpurun_out/r04_final/summary.txt
