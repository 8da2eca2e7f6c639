#!/bin/bash
# probes + per-phase stamps of the headline kernel (diagnostic build)
TAG=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 120 scripts/_bin/lu_probe > gpurun_out/${TAG}_lu_probe.txt 2>&1
rc=$?; cat gpurun_out/${TAG}_lu_probe.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
CEDARHIP_LIB=libcedarhip_stamps.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skew > gpurun_out/${TAG}_stamps.json 2> gpurun_out/${TAG}_stamps.err
echo "stamps rc=$?"; grep pstamps gpurun_out/${TAG}_stamps.err | tail -1
