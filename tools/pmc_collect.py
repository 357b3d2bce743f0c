#!/usr/bin/env python3
"""Collect the judged profiling evidence on the GPU box (run inside gpurun) and condense it into
profiles/: per target one `rocprofv3 --kernel-trace --stats` pass (durations) and separate `--pmc`
passes (HBM bytes; instruction mix), exactly as MI355X_MICROARCH.md prescribes:

  * FETCH_SIZE and WRITE_SIZE in two passes (TCC has 4 slots; they need 3 and 2);
  * gfx950 correction: FETCH_SIZE under-reports wide coalesced reads by exactly 2x -> bytes =
    FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is exact for 16-B/lane streaming stores;
  * SQ counters in passes of at most 8; GRBM_GUI_ACTIVE rides along (GRBM has its own slots).

This process never touches the GPU: every workload is a CHILD started as `rocprofv3 ... -- python3 X`.

Outputs (tag = e.g. r02) go to gpurun_out/profiles/ (the only directory that comes back from the box) and
are then copied into profiles/ by hand:
  profiles/<tag>_<target>_kernels.{json,md}   every full-size kernel of the target: launches, avg ms,
                                              registers, HBM bytes/launch, VALU / fp64 instruction counts
  profiles/pmc_traffic.json                   HBM bytes per launch keyed "<config>:<launch name>" -> bench.py
  profiles/pmc_flops.json                     fp64 flops per ray keyed the same way              -> bench.py
Both carry `csrc_hash` (bench.csrc_hash(): sha256 over pyxfocus_amd/csrc sources) and the commit they
were captured at; bench.py ignores them when the kernel sources have changed since.

usage: pmc_collect.py --tag r02 --commit <sha> [--targets config2,config3,config4,ops] [--quick]
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PMC_PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sqA": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
            "SQ_INSTS_VALU_TRANS_F64", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    "sqB": ["SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU",
            "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
}

# launch name of tools/workloads.py -> (substring that identifies its kernel, algorithmic B/ray)
LAUNCH_KERNEL = {
    "transform": "TransformOp<false>", "transform(g)": "TransformOp<false>", "itransform(g)": "TransformOp<true>",
    "wolterprimary": "WolterPrimaryOp<false>", "woltersecondary": "WolterSecondaryOp", "reflect": "ReflectOp",
    "reflect(g)": "ReflectOp", "flat": "FlatOp<false>", "flat(g)": "FlatOp<false>", "flat(focal)": "FlatOp<false>",
    "radgrat": "RadgratOp<false>", "zernsurf": "ZernOpSel<7, false, false>",
}


def op_name(kernel):
    """`void pxf::ray_kernel<pxf::TransformOp<false>, 2, false, true>(...)` -> `TransformOp<false>`;
    other kernels: `pxf::program_kernel<...>` -> `program_kernel<...>` (argument list dropped)."""
    k = kernel
    m = re.search(r"ray_kernel<pxf::", k)
    if m:
        i, depth, j = m.end(), 0, m.end()
        while j < len(k):
            c = k[j]
            if c == "<":
                depth += 1
            elif c == ">":
                if depth == 0:
                    break
                depth -= 1
            elif c == "," and depth == 0:
                break
            j += 1
        return k[i:j].strip()
    k = re.sub(r"^void\s+", "", k)
    if k.startswith("_ZN3pxf19program_fast_kernel"):     # (an empty template pack the profiler's demangler gives up on)
        return "program_fast_kernel<interpreter>"
    k = k.split("(")[0]
    return k.replace("pxf::", "")[:70]


def newest(pattern):
    f = glob.glob(pattern, recursive=True)
    f.sort(key=os.path.getmtime)
    return f[-1] if f else None


def run(cmd, log):
    t0 = time.time()
    with open(log, "w") as fh:
        rc = subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")).returncode
    print("  [%5.1fs rc=%d] %s" % (time.time() - t0, rc, " ".join(cmd[:8]) + " ..."), flush=True)
    return rc


def collect(target, workload, out):
    """All passes of one target; returns {op name: {...}} for the full-size launches."""
    os.makedirs(out, exist_ok=True)
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out + "/trace", "--"] + workload,
             out + "/trace.log")
    if rc:
        raise RuntimeError("kernel-trace pass of %s failed: see %s/trace.log" % (target, out))
    for tag, counters in PMC_PASSES.items():
        rc = run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out + "/" + tag, "--"]
                 + workload, out + "/" + tag + ".log")
        if rc:
            print("  !! pmc pass %s of %s failed (kept going)" % (tag, target))
    ks = {}
    tr = newest(out + "/trace/**/*_kernel_trace.csv")
    rows = list(csv.DictReader(open(tr)))
    for r in rows:
        r["op"] = op_name(r["Kernel_Name"])
        r["grid"] = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
    maxgrid = defaultdict(int)
    for r in rows:
        maxgrid[r["op"]] = max(maxgrid[r["op"]], r["grid"])
    for r in rows:
        if r["grid"] != maxgrid[r["op"]]:
            continue                                   # parity launches on 1e5 rays etc.
        e = ks.setdefault(r["op"], {"kernel": r["Kernel_Name"][:160], "grid": r["grid"], "dur_ns": [],
                                    "wg": int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 0),
                                    "vgpr": int(r.get("VGPR_Count") or 0), "agpr": int(r.get("Accum_VGPR_Count") or 0),
                                    "sgpr": int(r.get("SGPR_Count") or 0), "lds": int(r.get("LDS_Block_Size") or 0),
                                    "scratch": int(r.get("Scratch_Size") or 0)})
        e["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for tag in PMC_PASSES:
        f = newest(out + "/" + tag + "/**/*_counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        mg = defaultdict(int)
        rr = list(csv.DictReader(open(f)))
        for r in rr:
            r["op"] = op_name(r["Kernel_Name"])
            mg[r["op"]] = max(mg[r["op"]], int(r["Grid_Size"]))
        for r in rr:
            if int(r["Grid_Size"]) == mg[r["op"]]:
                acc[r["op"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for op, d in acc.items():
            if op in ks:
                ks[op].setdefault("pmc", {}).update({c: sum(v) / len(v) for c, v in d.items()})
    import shutil
    for sub in ["trace"] + list(PMC_PASSES):          # raw CSVs are large (gpurun brings back <= 64 MiB): keep the logs only
        shutil.rmtree(os.path.join(out, sub), ignore_errors=True)
    for op, e in ks.items():
        e["launches"] = len(e["dur_ns"])
        e["avg_ms"] = sum(e["dur_ns"]) / len(e["dur_ns"]) / 1e6
        e["min_ms"] = min(e["dur_ns"]) / 1e6
        del e["dur_ns"]
        p = e.get("pmc", {})
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            e["hbm_read_bytes"] = p["FETCH_SIZE"] * 1024 * 2        # gfx950: x2 for wide coalesced reads
            e["hbm_write_bytes"] = p["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_INSTS_VALU_FMA_F64" in p:
            e["fp64_flops_per_launch"] = 64. * (p["SQ_INSTS_VALU_ADD_F64"] + p["SQ_INSTS_VALU_MUL_F64"] +
                                                2 * p["SQ_INSTS_VALU_FMA_F64"])
            e["fp64_trans_per_launch"] = 64. * p["SQ_INSTS_VALU_TRANS_F64"]
        if "SQ_WAVES" in p and p["SQ_WAVES"]:
            e["valu_insts_per_wave"] = p["SQ_INSTS_VALU"] / p["SQ_WAVES"]
        if "GRBM_GUI_ACTIVE" in p and "SQ_INSTS_VALU" in p:
            # fp64 VALU issue: 4 cycles per wave-instruction per SIMD; 1024 SIMDs; GRBM counts over 8 XCDs
            cyc = p["GRBM_GUI_ACTIVE"] / 8.
            e["gpu_cycles"] = cyc
            e["valu_issue_frac"] = (p["SQ_INSTS_VALU"] / 1024.) * 4. / cyc if cyc else None
    return ks


OUTDIR = os.path.join(ROOT, "gpurun_out", "profiles")       # only gpurun_out/ comes back from the box: copy into profiles/


def write_tables(tag, target, ks, rays, note, launch_bytes):
    os.makedirs(OUTDIR, exist_ok=True)
    base = os.path.join(OUTDIR, "%s_%s_kernels" % (tag, target))
    json.dump({"tag": tag, "target": target, "rays_per_launch": rays, "note": note, "kernels": ks}, open(base + ".json", "w"),
              indent=1)
    alg, per_op = {}, defaultdict(list)
    for lname, nb in launch_bytes:                  # a kernel launched several times per chain with different write
        per_op[LAUNCH_KERNEL.get(lname)].append(nb)  # sets (transform: pure translation 80 B, rotation 144 B): the
    for op in ks:                                   # trace averages over them, so does the algorithmic figure
        if per_op.get(op):
            alg[op] = sum(per_op[op]) / len(per_op[op])
    for op in ks:                                   # surface intersections: 6 columns in, 6 out
        if op.split("<")[0] in ("LLOp", "WSOp", "WolterSineOp", "ZernOpSel", "WolterPrimaryOp", "WolterSecondaryOp"):
            alg.setdefault(op, 96)
        if op == "program_mirror_kernel<12, true>":
            alg.setdefault(op, 120)                 # the L-L pair of tools/bench_ops.py: 6 columns in, 9 out
        if op == "program_mirror_kernel<0, true>":
            alg.setdefault(op, 144)                 # the W-S pair: 9 in (a failed W-S ray keeps its normal), 9 out
        if op.split("<")[0] in ("program_kernel", "program_zern_kernel", "program_fast_kernel", "program_chain_kernel") and target not in ("ops", "config5"):
            alg.setdefault(op, 120)                 # the configs' programs: 6 columns in, 9 out
        if op.split("<")[0] in ("program_seg_kernel", "program_fast_seg_kernel"):
            alg.setdefault(op, 160)                 # SPO pair: 9 in (a cone miss keeps position+normal), 9 out, 2 angle columns
    lines = ["# rocprofv3 summary `%s` / `%s`" % (tag, target), "", note, "",
             "Durations: `rocprofv3 --kernel-trace --stats`; HBM bytes: two `--pmc` passes (FETCH_SIZE x 1024 x 2 per "
             "MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE x 1024); instruction mix: `--pmc` SQ passes. "
             "Full-size launches only (%s rays).  fp64 flops = 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions; "
             "`VALU issue` = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8)." % rays, "",
             "| kernel | n | avg ms | alg B/ray | alg GB/s | of 8 TB/s | HBM B/ray (PMC) | traffic/alg | VALU/wave | fp64 flop/ray | "
             "TFLOP/s | VALU issue | VGPR | SGPR | LDS | scratch |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for op in sorted(ks, key=lambda o: -ks[o]["avg_ms"] * ks[o]["launches"]):
        e = ks[op]
        if e["avg_ms"] < 0.02:
            continue
        a = alg.get(op)
        gbs = a * rays / (e["avg_ms"] * 1e-3) / 1e9 if (a and rays) else None
        hb = e.get("hbm_bytes_per_launch")
        fl = e.get("fp64_flops_per_launch")
        lines.append("| `%s` | %d | %.3f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %d | %d | %d | %d |" % (
            op[:60], e["launches"], e["avg_ms"], ("%g" % a) if a else "-", "%.0f" % gbs if gbs else "-",
            "%.1f %%" % (gbs / 80.) if gbs else "-", "%.1f" % (hb / rays) if (hb and rays) else "-",
            "%.3f" % (hb / rays / a) if (hb and rays and a) else "-",
            "%.0f" % e["valu_insts_per_wave"] if "valu_insts_per_wave" in e else "-",
            "%.0f" % (fl / rays) if (fl and rays) else "-",
            "%.1f" % (fl / (e["avg_ms"] * 1e-3) / 1e12) if fl else "-",
            "%.0f %%" % (100 * e["valu_issue_frac"]) if e.get("valu_issue_frac") else "-",
            e["vgpr"], e["sgpr"], e["lds"], e["scratch"]))
    open(base + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--targets", default="config2,config3,config4,config5,ops")
    ap.add_argument("--quick", action="store_true", help="1e8-ray configs 3/4 (faster passes)")
    a = ap.parse_args()
    import bench
    from tools import workloads as wl
    py = ["python3"]
    b = os.path.join(ROOT, "bench.py")
    crays = int(1e8 if a.quick else 1e9)
    targets = {
        "config2": (py + [b, "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--configs", ""], int(1e8), wl.CONFIG2,
                    "BASELINE config 2 as bench.py runs it (1e8 rays, one launch per call, + the fused program)."),
        "config3": (py + [b, "--steps", "1", "--warmup", "0", "--rays", "1e6", "--no-cpu-baseline", "--no-fused", "--configs", "3",
                          "--config-rays", str(crays), "--config-steps", "2"], crays, wl.CONFIG3,
                    "BASELINE config 3 (two-bounce + radial grating), %g rays, per call and fused." % crays),
        "config4": (py + [b, "--steps", "1", "--warmup", "0", "--rays", "1e6", "--no-cpu-baseline", "--no-fused", "--configs", "4",
                          "--config-rays", str(crays), "--config-steps", "2"], crays, wl.CONFIG4,
                    "BASELINE config 4 (36-term Zernike surface), %g rays, per call and fused (program_zern)." % crays),
        "config5": (py + [b, "--steps", "1", "--warmup", "0", "--rays", "1e6", "--no-cpu-baseline", "--no-fused", "--configs", "5",
                          "--config-rays", str(crays), "--config-steps", "1"], crays, [],
                    "BASELINE config 5 share (Arcus petal: segmented source, segmented SPO-pair program exact + fast, grating "
                    "program, focusY), %g rays; the stacked-grating while loop runs on 1e8 rays." % crays),
        "ops": (py + [os.path.join(ROOT, "tools", "bench_ops.py"), "--rays", "1e8", "--reps", "3", "--only",
                      "woltersine,primaryLL,secondaryLL,wsPrimary,wsSecondary,wsPrimaryB,wsSecondaryB,zernsurf36,zernsurfN10,zernsurfN15,zernconic36,LL,W-S"],
                int(1e8), [], "Legendre-Legendre / Wolter-Schwarzschild / woltersine / Zernike kernels one by one "
                "(tools/bench_ops.py, 1e8 rays)."),
    }
    traffic = {"csrc_hash": bench.csrc_hash(), "ray_kernels_hash": bench.ray_kernels_hash(), "commit": a.commit,
               "source": "tools/pmc_collect.py --tag " + a.tag, "kernels": {}}
    flops = {"csrc_hash": bench.csrc_hash(), "ray_kernels_hash": bench.ray_kernels_hash(), "commit": a.commit,
             "source": "tools/pmc_collect.py --tag " + a.tag, "kernels": {}}
    for fn, blob in (("pmc_traffic.json", traffic), ("pmc_flops.json", flops)):      # keep entries of targets not re-run
        try:
            old = json.load(open(os.path.join(ROOT, "profiles", fn)))
            if old.get("csrc_hash") == blob["csrc_hash"]:
                blob["kernels"].update(old.get("kernels", {}))
        except Exception:
            pass
    for t in [t for t in a.targets.split(",") if t]:
        cmd, rays, launches, note = targets[t]
        print("== target %s" % t, flush=True)
        out = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (a.tag, t))
        ks = collect(t, cmd, out)
        shown = " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd)
        write_tables(a.tag, t, ks, rays, note + "  Commit %s, csrc %s.  Command: `rocprofv3 --kernel-trace --stats -- %s` "
                     "(and the same under `--pmc ...`)." % (a.commit, bench.csrc_hash(), shown), launches)
        shared = defaultdict(list)
        for lname, nb in launches:
            shared[LAUNCH_KERNEL.get(lname)].append(nb)
        for lname, nb in launches:
            op = LAUNCH_KERNEL.get(lname)
            e = ks.get(op)
            if not e:
                continue
            if "hbm_bytes_per_launch" in e:
                nbs = shared[op]
                ent = {"rays": rays, "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
                       "algorithmic_bytes_per_launch": sum(nbs) / len(nbs) * rays, "kernel": op}
                if len(set(nbs)) > 1:
                    ent["note"] = ("this kernel runs %d times per chain with different write sets (%s B/ray): both figures "
                                   "are averages over those launches" % (len(nbs), "/".join(str(b) for b in nbs)))
                traffic["kernels"]["%s:%s" % (t, lname)] = ent
            if "fp64_flops_per_launch" in e:
                flops["kernels"]["%s:%s" % (t, lname)] = {"rays": rays, "fp64_flops_per_ray": e["fp64_flops_per_launch"] / rays,
                                                         "valu_insts_per_wave": e.get("valu_insts_per_wave"), "kernel": op}
        json.dump(traffic, open(os.path.join(OUTDIR, "pmc_traffic.json"), "w"), indent=1)
        json.dump(flops, open(os.path.join(OUTDIR, "pmc_flops.json"), "w"), indent=1)


def retable(tag):
    """Rebuild profiles/<tag>_*_kernels.md from the committed JSONs (after a change to the table logic; no GPU)."""
    global OUTDIR
    from tools import workloads as wl
    OUTDIR = os.path.join(ROOT, "profiles")
    lb = {"config2": wl.CONFIG2, "config3": wl.CONFIG3, "config4": wl.CONFIG4}
    for f in sorted(glob.glob(os.path.join(OUTDIR, tag + "_*_kernels.json"))):
        d = json.load(open(f))
        ks = {}
        for op, e in d["kernels"].items():
            ks[op_name(e["kernel"]) if op.startswith("_ZN") else op] = e
        write_tables(tag, d["target"], ks, d["rays_per_launch"], d["note"], lb.get(d["target"], []))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--retable":
        retable(sys.argv[2])
    else:
        main()
