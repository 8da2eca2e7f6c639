#!/bin/bash
# One gpurun call: GPU test suite, then (only if the suite was not killed / timed out) the bench line and the 2-rank rehearsal.
# usage: scripts/gpu_round.sh TAG [pytest args...]
TAG=${1:-r03}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${TAG}_tests.log
tail -5 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rb=$?
echo "bench rc=$rb"; cat gpurun_out/${TAG}_bench.json | cut -c1-600
if [ $rb -eq 124 ] || [ $rb -eq 137 ]; then exit $rb; fi
CEDARHIP_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_n2.json 2> gpurun_out/${TAG}_n2.err
echo "n2 rc=$?"; tail -c 1500 gpurun_out/${TAG}_n2.json; tail -5 gpurun_out/${TAG}_n2.err
exit $rc
