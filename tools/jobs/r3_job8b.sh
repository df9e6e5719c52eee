#!/bin/bash
# round 3, GPU job 8b: the evidence committed under profiles/ -- bench line, rocprofv3 kernel stats of the same command, counters
set -e
R=$(pwd)
bash tools/pmc_traffic.sh > ${F3D_OUT:-$R/gpurun_out}/r3_pmc_traffic.log 2>&1 || { tail -20 ${F3D_OUT:-$R/gpurun_out}/r3_pmc_traffic.log; exit 1; }
tail -30 ${F3D_OUT:-$R/gpurun_out}/r3_pmc_traffic.log
cp ${F3D_OUT:-$R/gpurun_out}/traffic/pmc_traffic.json profiles/r03_pmc_traffic.json   # so that the bench line of this job carries it
bash tools/profile_round.sh > ${F3D_OUT:-$R/gpurun_out}/r3_profile_round.log 2>&1 || { tail -20 ${F3D_OUT:-$R/gpurun_out}/r3_profile_round.log; exit 1; }
tail -5 ${F3D_OUT:-$R/gpurun_out}/r3_profile_round.log
