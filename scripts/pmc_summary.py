"""Summarise the rocprofv3 outputs of scripts/profile_round.sh into small JSON/CSV files for profiles/."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
WORKLOAD = sys.argv[3] if len(sys.argv) > 3 else "bench.py --steps 1 --warmup 0 (1024-DFF array transient); per-launch averages"
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "auto"
if KERNEL == "auto":
    # the dominant kernel of the bench: the persistent transient kernel when the device-resident stepper ran, else the per-attempt kernel
    KERNEL = "tran_persistent_kernel"
    for f in glob.glob(os.path.join(out, tag + "_pmc_*", "**", "*counter_collection.csv"), recursive=True):
        if "tran_persistent_kernel" not in open(f).read():
            KERNEL = "newton_block_kernel"
        break


def counters(dirname):
    acc, n = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL not in r["Kernel_Name"]:
                continue
            k = r["Counter_Name"]
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    return {k: acc[k] / n[k] for k in acc}, (max(n.values()) if n else 0)


import hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedarsim_jl_amd import engine as _eng  # noqa: E402
res = {"kernel": KERNEL, "workload": WORKLOAD,
       "lib_sha256": hashlib.sha256(open(_eng.LIB_PATH, "rb").read()).hexdigest(), "ch_version": _eng.load_library().ch_version().decode(),
       "collection": "rocprofv3 --kernel-trace --pmc <group>, one group per pass (scripts/profile_round.sh)"}
allc, launches = {}, 0
for d in sorted(glob.glob(os.path.join(out, tag + "_pmc_*"))):
    if os.path.isdir(d):
        c, n = counters(d)
        allc.update(c)
        launches = max(launches, n)
res["launches_averaged"] = launches
res["counters_per_launch"] = allc
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    res["correction"] = "gfx950 FETCH_SIZE reports half of the fetched bytes: hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024"
    res["FETCH_SIZE_KB_per_launch"] = allc["FETCH_SIZE"]
    res["WRITE_SIZE_KB_per_launch"] = allc["WRITE_SIZE"]
    res["hbm_bytes_per_launch"] = (2 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024
if "SQ_WAVES" in allc and allc["SQ_WAVES"] > 0:
    w = allc["SQ_WAVES"]
    res["per_wave"] = {"wave_cycles": 4 * allc.get("SQ_WAVE_CYCLES", 0) / w, "valu_busy_cycles": 4 * allc.get("SQ_ACTIVE_INST_VALU", 0) / w,
                       "waitcnt_cycles": 4 * allc.get("SQ_WAIT_INST_ANY", 0) / w * 0 + 4 * allc.get("SQ_WAIT_ANY", 0) / w,
                       "issue_stall_cycles": 4 * allc.get("SQ_WAIT_INST_ANY", 0) / w, "valu_instructions": allc.get("SQ_INSTS_VALU", 0) / w,
                       "note": "SQ_* cycle counters count quad-cycles; x4 = shader cycles"}
f64 = [allc.get("SQ_INSTS_VALU_%s_F64" % k) for k in ("ADD", "MUL", "FMA", "TRANS")]
if all(v is not None for v in f64):
    add, mul, fma, trans = f64
    res["fp64_valu_instructions_per_launch"] = {"add": add, "mul": mul, "fma": fma, "trans": trans}
    res["fp64_flop_per_launch"] = 64.0 * (add + mul + 2 * fma + trans)
    res["fp64_flop_note"] = "wave-level instruction counts x 64 lanes (issue slots; 30 of 64 lanes are active during device evaluation)"
json.dump(res, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
for f in glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(out, tag + "_bench_kernel_stats.csv"), "w").write(open(f).read())
    print(open(f).read())
