#!/usr/bin/env python3
"""bench.py -- rays/sec through the two-bounce Wolter-I trace (BASELINE.json metric) on MI355X.

Workload (BASELINE config 2 / SURVEY 8d): per GPU 1e8 rays, annulus [primrad(z0), primrad(z0+L)],
r0=220, z0=8400, psi=1, L=100, fp64; one STEP = the reference call chain, one HIP launch per
PyXFocus API call (the drop-in contract):

    tran.transform(rays,0,0,-(z0+2L),0,0,0) -> surf.wolterprimary -> tran.reflect
        -> surf.woltersecondary -> tran.reflect -> surf.flat            (576 algorithmic B/ray)

Every step traces a FRESH bundle (the chain mutates rays in place, so re-tracing a traced bundle
would be different work): K+W bundles are generated on the device before the timed region
(8 GB each at 1e8 rays; 288 GB of HBM3E holds the default 12 with room to spare).  Inputs are
therefore resident in HBM when timing starts; nothing is generated, copied or reduced inside it.

N>1 (launched by torch.distributed.run, one rank per GPU): rays shard by global index, each rank
traces its own 1e8 rays (weak scaling), no collective in the data path.  After the timed region
the ranks all-reduce the focal-plane moments and all-gather the focal (x,y) over RCCL/xGMI -- the
one exchange the path has -- and report its time separately.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

R0, Z0, PSI, LMIR = 220., 8400., 1., 100.
BYTES = {"transform": 144, "wolterprimary": 96, "reflect": 72, "woltersecondary": 96, "flat": 96}
CHAIN = ["transform", "wolterprimary", "reflect", "woltersecondary", "reflect", "flat"]
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def chain_step(rays, tran, surf, ev=None):
    """One pass of the hot path.  `ev`: optional list of 7 torch.cuda.Event to bracket each launch."""
    def mark(i):
        if ev is not None:
            ev[i].record()
    mark(0)
    tran.transform(rays, 0, 0, -(Z0 + 2 * LMIR), 0, 0, 0)
    mark(1)
    surf.wolterprimary(rays, R0, Z0, PSI)
    mark(2)
    tran.reflect(rays)
    mark(3)
    surf.woltersecondary(rays, R0, Z0, PSI)
    mark(4)
    tran.reflect(rays)
    mark(5)
    surf.flat(rays)
    mark(6)


def cpu_baseline(gpu_chain, target_seconds=12.0):
    """Reference Fortran (oracle/_ref, OpenMP build) -- or our C port if _ref did not travel --
    timed on this box's host cores on a bounded sample of the same workload.  The first 1e5 rays of
    the sample also go through the GPU path (`gpu_chain(cols, fused)`), and the two results are
    compared: the checker's only other job in this file."""
    from oracle import api, cases
    # threads actually usable: the cgroup CPU quota (16 on a 1-GPU box), not the 256 visible cores;
    # must be in the environment BEFORE the OpenMP runtime initialises
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    kind, be = "port", None
    try:
        if api.ref_available(omp=True):
            be, kind = api.load_ref(omp=True), "reference"
    except OSError:
        be = None
    if be is None:
        be = api.load_port(omp=True)
    n = 10_000_000
    rng = np.random.default_rng(0)
    rin, rout = cases.primrad(Z0), cases.primrad(Z0 + LMIR)
    src = cases.annulus_rays(rng, n, rin, rout, 0.)
    c = {k: v.copy() for k, v in src.items()}
    cases.chain_config2(be, c)                      # warm-up (page faults, thread pool)
    reps, spent = 0, 0.0
    while spent < target_seconds and reps < 5000:
        c = {k: v.copy() for k, v in src.items()}
        t0 = time.perf_counter()
        cases.chain_config2(be, c)
        spent += time.perf_counter() - t0
        reps += 1
    # single-thread figure: the serial build of the same checker, 1e6-ray sample, ~3 s
    one = None
    try:
        be1 = api.load_ref(omp=False) if kind == "reference" else api.load_port(omp=False)
        n1, r1, s1 = 1_000_000, 0, 0.0
        while s1 < 3.0 and r1 < 1000:
            c1 = {k: v[:n1].copy() for k, v in src.items()}
            t0 = time.perf_counter()
            cases.chain_config2(be1, c1)
            s1 += time.perf_counter() - t0
            r1 += 1
        one = n1 * r1 / s1
    except Exception:
        one = None
    chk = 100_000
    want = {k: v[:chk] for k, v in c.items()}
    parity = {"rays": chk, "checker": kind}
    for tag, fused in (("", False), ("fused_", True)):
        got = gpu_chain({k: v[:chk].copy() for k, v in src.items()}, fused)
        err = np.sqrt(np.mean(sum((got[k] - want[k]) ** 2 for k in "xyz")))
        parity[tag + "rms_pos_err_mm"] = float(err)
        parity[tag + "bit_identical"] = bool(all(np.array_equal(got[k], want[k]) for k in cases.ALLCOLS))
    return {"value": n * reps / spent, "unit": "rays/s", "cores": cores, "kind": kind,
            "sample": "%d x %d rays, same config-2 chain, %s, %.1f s of CPU time" % (
                reps, n, "PyXFocus Fortran compiled with flang -O2 -fopenmp (oracle/_ref)" if kind == "reference"
                else "C restatement oracle/pxf_oracle.c -O2 -fopenmp", spent),
            "one_thread_rays_per_s": one, "parity_vs_gpu": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=float, default=1e8, help="rays per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--pool-gb", type=float, default=200.0, help="HBM budget for pre-generated bundles")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N>1 path with several ranks sharing one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(1, torch.cuda.device_count())   # (rehearsal: ranks may share a GPU under gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("PXF_BENCH_FORCE_DIST"):     # (the env knob rehearses the RCCL path on 1 GPU)
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    n_gpus = world

    import pyxfocus_amd.analyses as anal
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con

    N = int(args.rays)
    K, W = args.steps, args.warmup
    rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + LMIR, R0, Z0))
    bundle_gb = 80.0 * N / 1e9
    pool = max(1, min(K, int(args.pool_gb // max(bundle_gb, 1e-9))))

    def fresh(i):
        # global ray index: rank r owns [r*N, (r+1)*N) of step i's bundle -> independent of n_gpus
        return sources.annulus(rin, rout, N, rng="philox", seed=1000 + i, first_index=rank * N)

    # ---- warm-up -------------------------------------------------------------------------
    for i in range(W):
        b = fresh(-1 - i)
        chain_step(b, tran, surf)
        del b
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps -----------------------------------------------------
    bundles = [fresh(i) for i in range(pool)]
    regen = max(0, K - pool)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(K)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        if i >= pool:                       # only if K exceeds what HBM can hold pre-generated
            bundles[i % pool] = fresh(i)
        chain_step(bundles[i % pool], tran, surf, ev[i])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations from the HIP events recorded inside the timed region -----------
    per = {}
    for i in range(K):
        for j, name in enumerate(CHAIN):
            per.setdefault((j, name), []).append(ev[i][j].elapsed_time(ev[i][j + 1]))
    kern = []
    for (j, name), ms in sorted(per.items()):
        avg = float(np.mean(ms))
        kern.append({"launch": j, "kernel": name, "avg_ms": avg, "bytes_per_ray": BYTES[name],
                     "GBps": BYTES[name] * N / (avg * 1e-3) / 1e9})
    dom = max(kern, key=lambda k: k["avg_ms"])
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            ent = tj.get(dom["kernel"])
            if ent and int(ent.get("rays", 0)) == N:
                traffic = ent["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": dom["bytes_per_ray"] * N, "avg_launch_ms": dom["avg_ms"]}
    chain_gbps = 576.0 * N * K / (sum(k["avg_ms"] for k in kern) * K * 1e-3) / 1e9
    # measured device-to-device copy of the same nine columns ("also report against a
    # measured copy kernel") -- what a 9R+9W stream reaches on this GPU with no arithmetic at all
    try:
        srcb = bundles[(K - 1) % pool]
        dst = [torch.empty_like(srcb[1]) for _ in range(3)]
        ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(3):
            if rep == 1:
                ce[0].record()
            for j in range(9):
                dst[j % 3].copy_(srcb[1 + j])
        ce[1].record()
        torch.cuda.synchronize()
        copy_gbps = 2 * 144.0 * N / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
        del dst
        roofline["measured_copy_GBps"] = copy_gbps          # torch's own D2D copy kernel, same nine columns
        roofline["frac_of_measured_copy"] = dom["GBps"] / copy_gbps
        roofline["guide_copy_GBps"] = 6290.0                # MI355X_MICROARCH.md: float4 copy, "achievable"
        roofline["frac_of_guide_copy"] = dom["GBps"] / 6290.0
    except Exception:
        pass

    # ---- the same K steps again as ONE fused launch per step (reported separately, never mixed
    # into `value`: 120 algorithmic B/ray instead of 576; bit-identical results) --------------------
    last = bundles[(K - 1) % pool]
    del bundles                      # keep only the last traced bundle (focal-plane metrics below)
    torch.cuda.empty_cache()
    fused = None
    if not args.no_fused:
        import pyxfocus_amd as pxf
        fb = [fresh(10_000 + i) for i in range(pool)]
        with pxf.fused():
            chain_step(fb[0], tran, surf)               # warm-up (first bundle is re-generated below)
        fb[0] = fresh(10_000)
        fev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K)]
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        f0 = time.perf_counter()
        for i in range(K):
            if i >= pool:
                fb[i % pool] = fresh(10_000 + i)
            fev[2 * i].record()
            with pxf.fused():
                chain_step(fb[i % pool], tran, surf)
            fev[2 * i + 1].record()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        felapsed = time.perf_counter() - f0
        if dist:
            t = torch.tensor([felapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            felapsed = float(t.item())
        fms = float(np.mean([fev[2 * i].elapsed_time(fev[2 * i + 1]) for i in range(K)]))
        fused = {"value": N * n_gpus * K / felapsed, "unit": "rays/s", "ms_per_step": felapsed / K * 1e3,
                 "launches_per_step": 1, "algorithmic_bytes_per_ray": 120, "avg_launch_ms": fms,
                 "GBps": 120.0 * N / (fms * 1e-3) / 1e9, "frac_of_hbm_peak": 120.0 * N / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "note": "pyxfocus_amd.fused(): same calls recorded and run as one ray-program kernel"}
        del fb
        torch.cuda.empty_cache()

    # ---- focal plane: global metric via all-reduce, and the all-gather the north-star names ----
    rms = anal.rmsCentroid(last, group=True if dist else None)
    try:        # exact HPD of the whole (sharded) bundle: radix-select histograms all-reduced, no ray moves
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        hpd = anal.hpd(last, group=True if dist else None)
        hpd = {"mm": hpd, "ms": (time.perf_counter() - h0) * 1e3}
    except Exception as e:
        hpd = {"error": repr(e)[:200]}
    gather = None
    if dist and not args.no_gather:
        try:
            from pyxfocus_amd import dist as pdist
            torch.cuda.synchronize(); dist.barrier()
            g0 = time.perf_counter()
            gx, gy = pdist.all_gather_focal([last[1], last[2]])
            torch.cuda.synchronize(); dist.barrier()
            gather = {"ms": (time.perf_counter() - g0) * 1e3, "rays": int(gx.shape[0]),
                      "GB_received_per_rank": 16.0 * gx.shape[0] * (world - 1) / world / 1e9}
            del gx, gy
        except Exception as e:  # the trace result must survive a collective problem
            gather = {"error": repr(e)[:200]}

    out = None
    if rank == 0:
        value = N * n_gpus * K / elapsed
        out = {"metric": "rays/sec through two-bounce Wolter-I trace; RMS spot error vs Fortran ref",   # BASELINE.json
               "value": value, "unit": "rays/s", "rms_pos_err_vs_reference_mm": None,
               "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": "config2: %g rays/GPU annulus -> transform -> wolterprimary -> reflect -> "
                                      "woltersecondary -> reflect -> flat; r0=220 z0=8400 psi=1 L=100; one HIP "
                                      "launch per PyXFocus call" % N,
                          "rays_per_gpu": N, "launches_per_step": 6, "algorithmic_bytes_per_ray": 576,
                          "parallelism": "ray-sharded x%d, no data-path collective" % n_gpus,
                          "bundles_pregenerated": pool, "bundles_regenerated_in_timed_region": regen},
               "roofline": roofline,
               "chain_GBps": chain_gbps, "chain_frac_of_hbm_peak": chain_gbps / HBM_PEAK_GBS,
               "kernels": kern, "focal_rms_centroid_mm": rms, "focal_hpd": hpd, "focal_gather": gather,
               "fused": fused}
        if n_gpus == 1 and not args.no_cpu_baseline:
            def gpu_chain(cols, use_fused):
                import contextlib
                import pyxfocus_amd as pxf
                order = "opd x y z l m n ux uy uz".split()
                r = [torch.from_numpy(cols[k]).to(dev) for k in order]
                with (pxf.fused() if use_fused else contextlib.nullcontext()):
                    chain_step(r, tran, surf)
                return {k: v.cpu().numpy() for k, v in zip(order, r)}
            try:
                out["cpu_baseline"] = cpu_baseline(gpu_chain)
                # the metric's second half: RMS ray-position error of the GPU path vs the reference
                out["rms_pos_err_vs_reference_mm"] = out["cpu_baseline"]["parity_vs_gpu"]["rms_pos_err_mm"]
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
