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
}
profile "$TAG" "bench.py --steps 1 --warmup 0 (1024-DFF array transient); per-launch averages" auto \
  "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-skew -- "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-skew
if [ "$2" = "config5" ]; then
  profile "${TAG}_cfg5" "scripts/bench_configs.py config5 (128 ASAP7 BSIM-CMG inverters, two transients); per-launch averages" "newton_block_kernel<16, true>" \
    "$ROOT/scripts/bench_configs.py" config5 -- "$ROOT/scripts/bench_configs.py" config5
fi
