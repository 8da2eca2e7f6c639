#!/bin/bash
# The round's closing measurement in one gpurun call: profiles of the bench workload, config 5 and the coupled sparse path
# (scripts/profile_round.sh), then every secondary configuration (scripts/bench_configs.py).  usage: scripts/gpu_final.sh TAG
TAG=${1:-r03}
mkdir -p gpurun_out
bash scripts/profile_round.sh $TAG config5 coupled > gpurun_out/${TAG}_profile.log 2>&1
echo "profile rc=$?"; tail -3 gpurun_out/${TAG}_profile.log
timeout -k 10 600 python scripts/bench_configs.py > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err
echo "configs rc=$?"; cut -c1-300 gpurun_out/${TAG}_configs.json
