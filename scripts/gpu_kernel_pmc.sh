#!/bin/bash
# Counter passes for ONE kernel of the coupled sparse workload.  usage: scripts/gpu_kernel_pmc.sh TAG KERNEL_SUBSTRING
TAG=${1:-r03}; KERN=${2:-sp3_top_kernel}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp CEDARHIP_COUPLED_TILES=1024 CEDARHIP_COUPLED_FORMS=sparse
cd /tmp
n=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_kpmc_$n" -o run -- python3 "$ROOT/scripts/bench_configs.py" coupled > "$OUT/${TAG}_kpmc_$n.log" 2>&1 || echo "pass $n ended with status $?"
done
cd $ROOT
python3 - "$OUT" "$TAG" "$KERN" <<'PY'
import csv, glob, os, sys, json
out, tag, kern = sys.argv[1:4]
acc = {}; cnt = {}
for f in glob.glob(os.path.join(out, tag + "_kpmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
res = {k: acc[k] / cnt[k] for k in sorted(acc)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, tag + "_kpmc.json"), "w"), indent=1)
PY
find $OUT -path "$OUT/${TAG}_kpmc_*" -name "*.csv" -size +256k -delete
