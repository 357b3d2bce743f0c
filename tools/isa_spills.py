#!/usr/bin/env python3
"""Per-kernel spill census of a `hipcc --save-temps` assembly file (*.s): SGPR spills to VGPR lanes
(v_writelane / v_readlane), scratch traffic, AGPR moves, and -- the pattern looked for in the Zernike
miscompile hunt -- a VGPR that holds SGPR-spill lanes being itself stored to scratch
while EXEC may be partial (no `s_or_saveexec_b64 ..., -1` bracket right before the store).
usage: isa_spills.py file.s [name-filter]"""
import re
import subprocess
import sys


def expand(tok):
    """a12 -> [12]; a[4:7] -> [4,5,6,7]"""
    m = re.fullmatch(r"a(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
    return list(range(int(m.group(1)), int(m.group(2)) + 1)) if m else []


def undefined_agprs(body):
    """AGPRs that some instruction READS but no instruction of the kernel ever WRITES (kernel-wide, so
    control flow cannot excuse it): the signature of the dropped half of a 64-bit VGPR->AGPR
    'Reload Reuse' copy found in the order-13 Zernike kernel."""
    written, read = set(), set()
    for ln in body:
        ln = ln.split(";")[0].strip()
        if not ln or ln.startswith(".") or ln.endswith(":"):
            continue
        op, _, rest = ln.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        ag = [(i, a) for i, t in enumerate(toks) for a in expand(t.split(" ")[0])]
        if not ag:
            continue
        store = "store" in op or op.startswith("ds_write") or op.startswith("ds_store")
        for i, a in ag:
            if i == 0 and not store:
                written.add(a)
            else:
                read.add(a)
    return sorted(read - written)


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    lines = open(path).read().split("\n")
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_Z\w+:", ln)]
    starts.append(len(lines))
    names = [lines[i].split(":")[0] for i in starts[:-1]]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.split("\n")
    print("%-88s %6s %6s %6s %6s %6s %s" % ("kernel", "wlane", "rlane", "sc_st", "sc_ld", "accvg", "AGPRs read but never written | lane-VGPRs stored to scratch under partial EXEC"))
    for k, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
        d = dem[k]
        if flt not in d:
            continue
        body = lines[a:b]
        wl = [ln for ln in body if "v_writelane_b32" in ln]
        lane_regs = set(re.findall(r"v_writelane_b32\s+(v\d+)", "\n".join(wl)))
        sus = []
        for i, ln in enumerate(body):
            m = re.search(r"scratch_store_dword\w*\s+off,\s*(v\d+)", ln) or re.search(r"buffer_store_dword\s+(v\d+),\s*off", ln)
            if m and m.group(1) in lane_regs:
                prev = "\n".join(body[max(0, i - 4):i])
                if "s_or_saveexec_b64" not in prev and "s_xor_saveexec_b64" not in prev:
                    sus.append("%s@+%d" % (m.group(1), i))
        txt = "\n".join(body)
        print("%-88s %6d %6d %6d %6d %6d %s" % (d[:88], len(wl), txt.count("v_readlane_b32"), txt.count("scratch_store"),
                                              txt.count("scratch_load"), txt.count("v_accvgpr"),
                                              ",".join("a%d" % a for a in undefined_agprs(body)[:8]) + " | " + ",".join(sus[:6])))


if __name__ == "__main__":
    main()
