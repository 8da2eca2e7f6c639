#!/bin/bash
# Counter passes for the config-5 persistent kernel: where do its wave cycles wait?  usage: scripts/gpu_cfg5_pmc.sh TAG
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $OUT/${TAG}_avail.txt 2>&1
grep -o "SQ_WAIT[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_ICACHE[A-Z_]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_BUSY[A-Z_]*\|SQC_[A-Z_]*" $OUT/${TAG}_avail.txt | sort -u | tr '\n' ' ' > $OUT/${TAG}_avail_sq.txt
n=0
for grp in "SQ_WAIT_IFETCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_c5pmc_$n" -o run -- python3 "$ROOT/scripts/bench_configs.py" config5 > "$OUT/${TAG}_c5pmc_$n.log" 2>&1 || echo "pass $n ended with status $?"
  echo "pass $n done"
done
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, json
out, tag = sys.argv[1], sys.argv[2]
acc = {}; cnt = {}
for f in glob.glob(os.path.join(out, tag + "_c5pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "tran_persistent_kernel" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
res = {k: acc[k] / cnt[k] for k in sorted(acc)}
json.dump(res, open(os.path.join(out, tag + "_c5pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
