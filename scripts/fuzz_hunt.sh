#!/bin/bash
# narrow down a seed that hangs the extended fuzz: chunks of $3 seeds from $1 to $2, each under its own timeout
A=${1:-6000}; B=${2:-6250}; STEP=${3:-25}
mkdir -p gpurun_out
: > gpurun_out/fuzz_hunt.log
for ((s=A; s<B; s+=STEP)); do
  timeout -k 5 40 python scripts/extended_fuzz.py $s $STEP 30 > gpurun_out/fuzz_chunk.log 2>&1
  rc=$?
  echo "chunk $s..$((s+STEP-1)) rc=$rc $(tail -1 gpurun_out/fuzz_chunk.log | cut -c1-160)" | tee -a gpurun_out/fuzz_hunt.log
done
