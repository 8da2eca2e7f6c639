#!/bin/bash
# sparse-path round: tests that touch the sparse path, the coupled-array measurement, and a kernel-level profile of it
TAG=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_torn.py tests/test_gpu_parity.py tests/test_gpu_ac_noise.py -m gpu -q -s > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
export CEDARHIP_COUPLED_TILES=64,1024
timeout -k 10 600 python scripts/bench_configs.py coupled > gpurun_out/${TAG}_coupled.json 2> gpurun_out/${TAG}_coupled.err
rb=$?; if [ $rb -eq 124 ] || [ $rb -eq 137 ]; then exit $rb; fi
export CEDARHIP_COUPLED_TILES=1024 TMPDIR=/tmp
ROOT=$(pwd); cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_coupled_prof -o run -- python3 $ROOT/scripts/bench_configs.py coupled > $ROOT/gpurun_out/${TAG}_coupled_prof.log 2>&1
cut -c1-150 $ROOT/gpurun_out/${TAG}_coupled_prof/run_kernel_stats.csv | head -22
find $ROOT/gpurun_out/${TAG}_coupled_prof -name "*kernel_trace.csv" -size +256k -delete
