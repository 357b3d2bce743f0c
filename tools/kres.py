#!/usr/bin/env python3
"""Dev helper: compact per-kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:] 
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                      "-fno-fast-math", "-c", "-o", "/tmp/_kres.o", src,
                      "-Rpass-analysis=kernel-resource-usage"] + (extra or ["-ffp-contract=off"]),
                     capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    for k in ["SGPRs Spill", "VGPRs Spill", "VGPRs", "AGPRs", "ScratchSize", "Occupancy", "TotalSGPRs", "LDS Size"]:
        m = re.search(r"remark:\s+" + re.escape(k) + r"[^:]*: (\d+)", line)
        if m and cur is not None:
            cur.setdefault(k, m.group(1))
            break
for r in rows:
    if filt in r["name"]:
        print("%-78s V%-4s A%-3s S%-4s scr%-4s occ%-2s lds%-5s sspill%-3s vspill%s" % (
            r["name"][:78], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
            r.get("Occupancy"), r.get("LDS Size"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
