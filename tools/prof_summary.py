#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<tag>_summary.{json,md}.

Per kernel (FULL-SIZE launches only: the bench also makes one 1e5-ray parity launch per kernel,
which would skew rocprof's own --stats average): launch count, average duration from the kernel
trace, and HBM bytes per launch from the PMC passes, corrected as MI355X_MICROARCH.md prescribes
for gfx950: FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x -> bytes =
FETCH_SIZE*1024*2; WRITE_SIZE (KiB) is exact for 16-B/lane streaming stores -> bytes =
WRITE_SIZE*1024.

usage: prof_summary.py <gpurun_out/prof_TAG> <tag> <profiles dir>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, tag, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
ALG = {"TransformOp": 144, "WolterPrimaryOp": 96, "WolterSecondaryOp": 96, "ReflectOp": 72, "FlatOp": 96,
       "RadgratOp": 64, "ConicOp": 96, "ZernOpSel": 96, "source_kernel": 80,
       "program_kernel": 120}  # program_kernel: the config-2 chain fused (6 columns read, 9 written)


def short(name):
    m = re.search(r"ray_kernel<pxf::(\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"pxf::(\w+)", name)
    return m.group(1) if m else name[:40]


def rows(pattern):
    f = glob.glob(os.path.join(src, pattern))
    if not f:
        return []
    f.sort(key=os.path.getmtime)          # gpurun merges into an existing directory: newest run wins
    with open(f[-1]) as fh:
        return list(csv.DictReader(fh))


trace = rows("trace/*/*_kernel_trace.csv")
for r in trace:  # the kernel-trace CSV names the launch geometry per axis
    r.setdefault("Grid_Size", r.get("Grid_Size_X", "0"))
    r.setdefault("Workgroup_Size", r.get("Workgroup_Size_X", "0"))
maxgrid = defaultdict(int)
for r in trace:
    maxgrid[short(r["Kernel_Name"])] = max(maxgrid[short(r["Kernel_Name"])], int(r["Grid_Size"]))
dur = defaultdict(list)
meta = {}
for r in trace:
    k = short(r["Kernel_Name"])
    if int(r["Grid_Size"]) == maxgrid[k]:
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = {"vgpr": int(r.get("VGPR_Count", 0) or 0), "sgpr": int(r.get("SGPR_Count", 0) or 0),
                   "lds": int(r.get("LDS_Block_Size", 0) or 0), "scratch": int(r.get("Scratch_Size", 0) or 0),
                   "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}
pmc = {}
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rr = rows(which + "/*/*_counter_collection.csv")
    mg = defaultdict(int)
    for r in rr:
        mg[short(r["Kernel_Name"])] = max(mg[short(r["Kernel_Name"])], int(r["Grid_Size"]))
    acc = defaultdict(list)
    for r in rr:
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == counter and int(r["Grid_Size"]) == mg[k]:
            acc[k].append(float(r["Counter_Value"]))
    pmc[counter] = {k: sum(v) / len(v) for k, v in acc.items()}

bench = {}
try:
    bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
except Exception:
    pass
N = bench.get("config", {}).get("rays_per_gpu")
summary = {"tag": tag, "rays_per_launch": N, "kernels": {}, "bench_line": bench}
lines = ["# rocprofv3 summary `%s`" % tag, "",
         "`rocprofv3 --kernel-trace --stats` (durations) and two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) of "
         "`bench.py`; full-size launches only (%s rays each). HBM bytes corrected per MI355X_MICROARCH.md "
         "(gfx950: FETCH_SIZE x2 for wide coalesced reads; WRITE_SIZE exact)." % N, "",
         "| kernel | launches | avg ms | alg. B/ray | alg. GB/s | HBM read GB | HBM write GB | HBM B/ray | traffic/alg | VGPR | LDS | scratch |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    avg_ms = sum(dur[k]) / len(dur[k]) / 1e6
    alg = ALG.get(k)
    rd = pmc["FETCH_SIZE"].get(k)
    wr = pmc["WRITE_SIZE"].get(k)
    rd_b = rd * 1024 * 2 if rd is not None else None
    wr_b = wr * 1024 if wr is not None else None
    ent = {"launches": len(dur[k]), "avg_ms": avg_ms, "alg_bytes_per_ray": alg, **meta[k],
           "FETCH_SIZE_KiB_raw": rd, "WRITE_SIZE_KiB_raw": wr, "hbm_read_bytes": rd_b, "hbm_write_bytes": wr_b}
    if N and alg:
        ent["alg_GBps"] = alg * N / (avg_ms * 1e-3) / 1e9
    if rd_b is not None and wr_b is not None:
        ent["hbm_bytes_per_launch"] = rd_b + wr_b
        if N:
            ent["hbm_bytes_per_ray"] = (rd_b + wr_b) / N
    summary["kernels"][k] = ent
    lines.append("| %s | %d | %.4f | %s | %s | %s | %s | %s | %s | %d | %d | %d |" % (
        k, len(dur[k]), avg_ms, alg, "%.0f" % ent["alg_GBps"] if "alg_GBps" in ent else "-",
        "%.3f" % (rd_b / 1e9) if rd_b is not None else "-", "%.3f" % (wr_b / 1e9) if wr_b is not None else "-",
        "%.1f" % ent["hbm_bytes_per_ray"] if "hbm_bytes_per_ray" in ent else "-",
        "%.3f" % (ent["hbm_bytes_per_ray"] / alg) if ("hbm_bytes_per_ray" in ent and alg) else "-",
        meta[k]["vgpr"], meta[k]["lds"], meta[k]["scratch"]))
os.makedirs(out_dir, exist_ok=True)
json.dump(summary, open(os.path.join(out_dir, tag + "_summary.json"), "w"), indent=1)
open(os.path.join(out_dir, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
# bench.py reads this to fill roofline.traffic for the dominant kernel
name_map = {"TransformOp": "transform", "WolterPrimaryOp": "wolterprimary", "WolterSecondaryOp": "woltersecondary",
            "ReflectOp": "reflect", "FlatOp": "flat"}
tj = {name_map[k]: {"rays": N, "hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "source": tag}
      for k, v in summary["kernels"].items() if k in name_map and "hbm_bytes_per_launch" in v}
if tj and len(sys.argv) > 4 and sys.argv[4] == "--traffic":
    json.dump(tj, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
