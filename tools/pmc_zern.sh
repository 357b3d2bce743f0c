#!/bin/bash
# PMC pass over the Zernike kernel (config 4): VALU utilisation / stalls.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_zern
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/a" -- python3 "$ROOT/tools/bench_configs.py" --rays 1e8 --reps 1 --configs 4 > "$OUT/a.json" 2> "$OUT/a.err" || { tail -5 "$OUT/a.err"; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d "$OUT/b" -- python3 "$ROOT/tools/bench_configs.py" --rays 1e8 --reps 1 --configs 4 > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(out + "/" + sub + "/*/*_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "Zern" in r["Kernel_Name"] or "Flat" in r["Kernel_Name"]:
            k = "zern" if "Zern" in r["Kernel_Name"] else "flat"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(sub, k, {c: sum(v) / len(v) for c, v in d.items()})
PY
