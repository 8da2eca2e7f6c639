#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r02            the bench workload (1024-DFF array)
#   bash scripts/profile_round.sh r02 config5    + the BSIM-CMG x256 workload (tag r02_cfg5)
# per workload:
# 1. rocprofv3 --kernel-trace --stats                -> gpurun_out/<tag>_stats/
# 2. separate --pmc passes (never combined with other trace domains; MI355X_MICROARCH.md HBM section)
# 3. scripts/pmc_summary.py                          -> gpurun_out/<tag>_pmc_traffic.json, <tag>_bench_kernel_stats.csv
# Copy the summaries you want judged into profiles/.
set -e -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
profile() {   # tag, description, kernel name ("auto": the bench's stepper kernel), stats arguments..., "--", pmc arguments...
  local tag=$1 what=$2 kern=$3; shift 3
  local stats_args=() pmc_args=() seen=0
  for a in "$@"; do if [ "$a" = "--" ]; then seen=1; elif [ $seen = 0 ]; then stats_args+=("$a"); else pmc_args+=("$a"); fi; done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${tag}_stats" -o run -- python3 "${stats_args[@]}" > "$OUT/${tag}_stats.log" 2>&1 || echo "rocprofv3 (stats) ended with status $? (a crash in the tool's teardown at process exit still leaves the CSVs)"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${tag}_pmc_$name" -o run -- python3 "${pmc_args[@]}" > "$OUT/${tag}_pmc_$name.log" 2>&1 || echo "rocprofv3 (pmc $name) ended with status $?"
    echo "pmc pass $name done ($tag)"
  done
  (cd "$ROOT" && python3 scripts/pmc_summary.py "$OUT" "$tag" "$what" "$kern")
  # the raw per-dispatch CSVs (tens of MB for a launch-heavy workload) stay on the box: gpurun copies back at most 64 MiB
  find "$OUT" -path "$OUT/${tag}_pmc_*" -name "*.csv" -size +256k -delete 2>/dev/null || true
  find "$OUT/${tag}_stats" -name "*kernel_trace.csv" -size +256k -delete 2>/dev/null || true
}
profile "$TAG" "bench.py --steps 1 --warmup 0 (1024-DFF array transient); per-launch averages" auto \
  "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-skew -- "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-skew
# The plain-C client under the profiler: rocprofv3 runs have ended with SIGSEGV inside __cxa_finalize AFTER main returned 0 (round 2).
# CEDARHIP_DEMO_MAPS=1 makes the client print the faulting address, the instruction pointer and /proc/self/maps from a SIGSEGV
# handler, so that the frame can be assigned to a mapped object (-> gpurun_out/<tag>_c_demo_profiled.log).
if gcc -std=c99 -O2 -I "$ROOT/include" "$ROOT/examples/c_abi_demo.c" -L "$ROOT/cedarsim.jl_amd/lib" -lcedarhip -Wl,-rpath,"$ROOT/cedarsim.jl_amd/lib" -lm -o /tmp/c_abi_demo_prof; then
  export CEDARHIP_DEMO_MAPS=1
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c_demo_stats" -o run -- /tmp/c_abi_demo_prof > "$OUT/${TAG}_c_demo_profiled.log" 2>&1 || echo "profiled C client ended with status $? (see ${TAG}_c_demo_profiled.log)"
  timeout -k 10 120 /tmp/c_abi_demo_prof > "$OUT/${TAG}_c_demo_plain.log" 2>&1 || echo "plain C client ended with status $?"
  unset CEDARHIP_DEMO_MAPS
fi
bash "$ROOT/scripts/exit_probe.sh" "$TAG"
if [ "$2" = "coupled" ] || [ "$3" = "coupled" ]; then
  # the coupled 1024-DFF array on the general sparse path (subtree form), dominant kernel sp3_group_kernel
  export CEDARHIP_COUPLED_TILES=1024 CEDARHIP_COUPLED_FORMS=sparse
  profile "${TAG}_coupled" "scripts/bench_configs.py coupled (1024-DFF array behind 1-ohm rails, sparse path only, two transients); per-launch averages" "sp3_group_kernel" \
    "$ROOT/scripts/bench_configs.py" coupled -- "$ROOT/scripts/bench_configs.py" coupled
  unset CEDARHIP_COUPLED_TILES CEDARHIP_COUPLED_FORMS
fi
if [ "$2" = "config5" ]; then
  profile "${TAG}_cfg5" "scripts/bench_configs.py config5 (128 ASAP7 BSIM-CMG inverters, two transients); per-launch averages" "tran_persistent_kernel" \
    "$ROOT/scripts/bench_configs.py" config5 -- "$ROOT/scripts/bench_configs.py" config5
fi
