#!/bin/bash
# ONE counter pass over the coupled sparse workload, per-kernel averages of every kernel.  usage: scripts/gpu_kernel_pmc1.sh TAG "COUNTERS..."
TAG=${1:-r03}; GRP=${2:-"SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES"}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp CEDARHIP_COUPLED_TILES=1024 CEDARHIP_COUPLED_FORMS=sparse
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $GRP --output-format csv -d "$OUT/${TAG}_k1" -o run -- python3 "$ROOT/scripts/bench_configs.py" coupled > "$OUT/${TAG}_k1.log" 2>&1 || echo "pass ended with status $?"
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, json, re
out, tag = sys.argv[1:3]
acc = {}; cnt = {}
for f in glob.glob(os.path.join(out, tag + "_k1", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("chip::", "").replace("void ", "")
        k = (kn, r["Counter_Name"]); acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
res = {}
for (kn, c), v in acc.items(): res.setdefault(kn, {})[c] = round(v / cnt[(kn, c)], 1)
for kn in sorted(res): print(kn, res[kn])
json.dump(res, open(os.path.join(out, tag + "_k1.json"), "w"), indent=1)
PY
find $OUT/${TAG}_k1 -name "*.csv" -size +256k -delete
