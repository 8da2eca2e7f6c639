#!/bin/bash
# One gpurun call for the compiled-device (config 5) path: its GPU tests, then the measurement.  usage: scripts/gpu_cfg5.sh TAG
TAG=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_va.py tests/test_gpu_stepper.py -m gpu -q -s -x > gpurun_out/${TAG}_cfg5_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${TAG}_cfg5_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/bench_configs.py config5 > gpurun_out/${TAG}_cfg5.json 2> gpurun_out/${TAG}_cfg5.err
echo "cfg5 rc=$?"; cat gpurun_out/${TAG}_cfg5.json | cut -c1-700
