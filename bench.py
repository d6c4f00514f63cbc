#!/usr/bin/env python3
"""Headline benchmark of the measurement-operator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2]           (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`; `--config`):
  C2 (default, the configuration the metric is quoted on)  exact Hessian apply  out = R^H W R x  (degrid + FFTs + grid
      with w-stacking), 1 band of 1e7 visibilities per GPU, 8192^2 image.  A step = one apply on every rank's band (the
      operator inside the per-band CG, which needs no communication); when N > 1 the timed region also holds the band sum
      of the result image -- ONE RCCL sum-to-root per --reduce-every applies (default: once per timed region).
  C1  the same apply at the reference's CPU-runnable size (1e5 visibilities, 1024^2).
  C5  the same apply at 1e8 visibilities, 16384^2 image, 64 ES-kernel w-planes.
  C3  per-band PCG solve on the C2 operator, one band per GPU: a step = one CG iteration (Hessian apply + the CG vector
      kernels, all in HBM: pfbhip_gridder_cg_dev); the timed region is ONE K-iteration solve followed by the RCCL
      sum-to-root of the solution-residual image (core/deconv.py:320-321).
  C4  SARA backward step: 4 bands of 4096^2 over the N GPUs, PSF-approximate Hessian (8192^2) + Psi / Psi^H
      (self, db1, db2, db3; 3 levels) + l21 dual update with its per-iteration band all-reduce; a step = one primal-dual
      iteration (pfbhip_primal_dual); the clock is the iteration loop alone (cubes resident in HBM).
Rank 0 prints ONE JSON line.

value        whole-job throughput: visibilities gridded + degridded per second (a Hessian apply touches every unmasked
             visibility twice), N * 2 * nactive / t_step, in Mvis/s; C4: PSF-Hessian applies per second.
roofline     the dominant device stage of the apply, timed live with HIP events on the handle's stream
             (pfbhip_gridder_profile): achieved = compulsory bytes of one launch / average launch duration.
             `actual_bytes_per_apply` / `actual_frac` sum the compulsory bytes of the PRUNED pipeline that runs (plus the
             plane clear) -- the honest HBM fraction; `apply_frac` keeps SURVEY section 8(d)'s unpruned accounting.
host_path    what a caller of the numpy-in / numpy-out surface sees (N = 1): hessian_slice, vis2dirty, dirty2vis per call.
cpu_baseline ducc0 when importable on this box (kind "ducc0"), else the oracle's CPU restatement (kind "port") of the
             same apply on this box's host cores.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
F64_PEAK_TFLOPS = 78.6  # f64 vector FMA: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (the chip holds ~2.0 GHz under this load)
# stage timer -> the kernel it brackets (name as rocprofv3 lists it); the record scatter / row-walk gather / transposing
# first-axis FFT replace k_grid_blk / k_degrid_mp / k_rowfft_plain + k_a2b / k_b2a where the plan admits them
KERNEL_OF = {"grid": "k_grid_rec", "degrid": "k_degrid_rw", "fft_rows": "k_rowfft_a2b / k_rowfft_b2a", "pad": "k_b2a",
             "crop": "k_a2b", "fft_crop": "k_fused_fft_crop", "pad_fft": "k_fused_pad_fft"}


def algorithmic_bytes(info, nx, ny, nrow, nactive):
    """Compulsory traffic: SURVEY.md section 8(d) (unpruned, per apply) and, per launch of every stage, of the PRUNED
    pipeline that runs.  Returns (survey bytes per apply, {stage: bytes per launch},
    {stage: launches per apply}, other bytes per apply)."""
    Sc, Sr = 16, 8
    G = info["nu"] * info["nv"] * Sc
    I = nx * ny * Sr
    P = info["nplanes"]
    b_vis = nactive * (2 * Sc + Sr + 2) + 2 * nrow * 24
    b_grid = P * (12 * G + 3 * I)
    occ = info["occ_rows"] / info["nu"]      # occupied fraction of the uv-plane rows (only those are cleared / transformed)
    B = ny * info["nu"] * Sc                 # cropped, transposed plane of the second axis
    ngroups = -(-P // 4)                     # scatter / gather / fused second-axis launches per direction (<= 4 planes each)
    ppl = P / ngroups                        # planes per such launch
    fused = bool(info.get("fft_mode", 0) & 2)
    tfft = bool(info.get("fft_mode", 0) & 8)
    rec = info.get("scatter_mode", 0) == 2
    wd = info.get("wmode", 0) == 2           # one plane, K kernel functions per axis: K values / coefficients per visibility
    K = info.get("nderiv", 0)
    nsl = max(int(info.get("scatter_launches", 1)), 1)
    vis_rec = 32 + (16 * K if wd else (16 * ppl if rec else 40))  # per visibility and pass: record + values / coordinates + value
    # cells of a plane the scatter / gather can touch (tiles with visibilities + halo): what the first-axis transforms and the
    # plane clear move of the occupied rows
    Au = info.get("used_cells", 0) * Sc or occ * G
    per_launch = {
        # used plane cells written (read-add-written by tile) once + the visibility records; one launch per tile colour
        "grid": (ppl * Au + nactive * vis_rec) / nsl,
        # used plane cells read once + records (+ the plane-weighted values it writes inside a Hessian apply)
        "degrid": ppl * Au + nactive * ((32 + 16 * K + 8 + 16 * K) if wd else (32 + 8 * ppl + (16 * ppl if rec else 16))),
    }
    launches = {"grid": ngroups * nsl, "degrid": ngroups}
    if fused:
        per_launch.update({
            # planes read once; image written (first group) or read + written; the last launch also reads the correction
            "fft_crop": ppl * occ * B + I * (2 - 1 / ngroups) + I / ngroups,
            # image and correction read once per launch (x * corr is formed in the load); occupied columns written per plane
            "pad_fft": 2 * I + ppl * occ * B,
        })
        launches.update({"fft_crop": ngroups, "pad_fft": ngroups})
        if tfft:  # first axis with the crop / pad + transpose folded in: occupied rows of A on one side, occupied columns of B
            # on the other; one launch per direction takes every plane of the pass
            per_launch["fft_rows"] = ppl * (Au + occ * B)
            launches["fft_rows"] = 2 * ngroups
        else:
            per_launch.update({"fft_rows": 2 * occ * G, "pad": occ * B + occ * G, "crop": occ * (ny / info["nv"]) * G + occ * B})
            launches.update({"fft_rows": 2 * P * 2, "pad": P, "crop": P})  # (two row spans per plane and direction)
            per_launch["fft_rows"] /= 2
    else:
        per_launch.update({"fft_rows": occ * G + B, "pad": (I + B + occ * B + occ * G) / 2,
                           "crop": (occ * (ny / info["nv"]) * G + B + (nx / info["nu"]) * B + 2 * I) / 2})
        launches.update({"fft_rows": 4 * P, "pad": 2 * P, "crop": 2 * P})
    other = P * Au  # clearing the used cells of the scatter's planes (on a side stream in single-pass plans)
    return b_vis + b_grid, per_launch, launches, other


def host_threads():
    nthr = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            nthr = min(nthr, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("PFB_CPU_THREADS", min(nthr, 16)))


def cpu_baseline(case, oracle_params, budget_s=60.0):
    """ducc0 on this box's host cores when it can be imported, else the oracle's CPU
    restatement of the same Hessian apply -- the whole apply when it fits the budget, else a bounded sample of planes."""
    c = case
    nthr = host_threads()
    wgt = np.ascontiguousarray(c["wgt"], dtype=np.float64)
    try:
        import ducc0  # noqa: F401
        from ducc0.wgridder.experimental import dirty2vis, vis2dirty

        kw = dict(uvw=c["uvw"], freq=c["freq"], pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0, epsilon=1e-7,
                  flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, nthreads=nthr, sigma_min=1.1,
                  sigma_max=3.0)
        ts = []
        for i in range(3):
            t0 = time.time()
            mv = dirty2vis(dirty=c["x"], mask=c["mask"], **kw)
            vis2dirty(vis=mv, wgt=wgt, mask=c["mask"], npix_x=c["nx"], npix_y=c["ny"], double_precision_accumulation=True, **kw)
            ts.append(time.time() - t0)
        t_apply = float(np.median(ts[1:]))
        nactive = int((c["mask"] != 0).sum())
        return {"value": 2 * nactive / t_apply / 1e6, "unit": "Mvis/s", "cores": nthr, "kind": "ducc0",
                "version": getattr(ducc0, "__version__", "?"), "sample": f"median of 2 whole applies after 1 warm-up, {nthr} threads",
                "sec_per_apply": t_apply}
    except ImportError:
        pass
    from oracle import _lib as olib
    from oracle import wgridder as owg

    olib.lib().pfbo_set_num_threads(nthr)
    owg.FFT_WORKERS = nthr
    t0 = time.time()
    plan = owg.Plan(c["uvw"], c["freq"], c["mask"], c["nx"], c["ny"], c["cell"], c["cell"], 0.0, 0.0, 1e-7, False, True, False, True,
                    False, params=oracle_params)
    t_plan = time.time() - t0
    P = plan.p.nplanes
    swgt = wgt.reshape(-1)
    acc_img = np.zeros((c["nx"], c["ny"]))
    sacc = np.zeros(plan.nrow * plan.nchan, dtype=np.complex128)
    dc = np.ascontiguousarray(c["x"])
    done, t = 0, 0.0
    for p in range(P):  # plane by plane until the whole apply or the budget is done
        t0 = time.time()
        plan.plane_round_trip(dc, swgt, p, acc_img, sacc)
        t += time.time() - t0
        done += 1
        if t > budget_s and done < P:
            break
    t_apply = t / done * P
    nactive = int(plan.active.sum())
    whole = done == P
    return {"value": 2 * nactive / t_apply / 1e6, "unit": "Mvis/s", "cores": int(olib.lib().pfbo_num_threads()), "kind": "port",
            "sample": (f"the whole apply ({P} w-planes" if whole else f"{done} of {P} w-planes of the same apply (scaled by {P}/{done}") +
                      f", all {nactive} vis, {c['nx']}^2 image, grid {plan.p.nu}x{plan.p.nv}); {t:.1f} s measured; ducc0 is not "
                      "importable on this box -- this is the repository's restatement, not the reference's speed",
            "sec_per_apply": t_apply, "plan_sec": t_plan}


def host_path(case, g, wsum, reps=3):
    """The callable surface as `pfb kclean / sara / fluxtractor` see it: numpy in, numpy out, plan cache keyed on content."""
    from pfb_imaging_amd import wgridder as wg
    from pfb_imaging_amd.operators.hessian import hessian_slice

    c = case
    out = {}
    hkw = dict(uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=None, cell=c["cell"], x0=0.0, y0=0.0,
               do_wgridding=True, epsilon=1e-7, eta=1e-3, wsum=wsum)
    xout = np.empty_like(c["x"])

    def timed(fn):
        fn()  # first call builds / binds (plan, weights): not timed
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    out["hessian_slice_ms"] = timed(lambda: hessian_slice(c["x"], **hkw))   # returns a new array (page-locked: D2H at the PCIe rate)
    # caller-owned pageable xout: the runtime stages the device-to-host copy (~12 GB/s on this host)
    out["hessian_slice_pageable_xout_ms"] = timed(lambda: hessian_slice(c["x"], xout=xout, **hkw))
    # the same arrays handed out read-only (the form Ray gives the band workers): hashed once, then recognised by address
    ro = {k: c[k].view() for k in ("uvw", "wgt", "mask", "freq")}
    for v in ro.values():
        v.flags.writeable = False
    frozen = dict(hkw, uvw=ro["uvw"], weight=ro["wgt"], vis_mask=ro["mask"], freq=ro["freq"])
    frozen_ok = all(not v.flags.writeable and v.base is not None and not v.base.flags.writeable for v in ro.values())
    if not frozen_ok:  # views of writeable arrays stay writeable through their base: freeze copies instead
        for k in ro:
            ro[k] = c[k].copy()
            ro[k].flags.writeable = False
        frozen = dict(hkw, uvw=ro["uvw"], weight=ro["wgt"], vis_mask=ro["mask"], freq=ro["freq"])
    out["hessian_slice_readonly_inputs_ms"] = timed(lambda: hessian_slice(c["x"], **frozen))
    x_host = c["x"]
    out["handle_hessian_ms"] = timed(lambda: g.hessian(x_host, eta=1e-3, wsum=wsum))
    # precision = "single" callers (float32 images and weights): half the PCIe bytes, sums still in double on the device
    x32, w32 = c["x"].astype(np.float32), c["wgt"].astype(np.float32)
    xout32 = np.empty_like(x32)
    out["hessian_slice_float32_ms"] = timed(lambda: hessian_slice(x32, **dict(hkw, weight=w32)))
    del x32, w32, xout32
    skw = dict(uvw=c["uvw"], freq=c["freq"], mask=c["mask"], pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0,
               epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1,
               sigma_max=2.6)
    vis = c.get("vis")
    if vis is None:
        rng = np.random.default_rng(0)
        vis = rng.standard_normal(c["mask"].shape) + 1j * rng.standard_normal(c["mask"].shape)
    out["vis2dirty_ms"] = timed(lambda: wg.vis2dirty(vis=vis, wgt=c["wgt"], npix_x=c["nx"], npix_y=c["ny"], **skw))
    out["dirty2vis_ms"] = timed(lambda: wg.dirty2vis(dirty=c["x"], **skw))
    from pfb_imaging_amd._lib import DeviceArray

    t0 = time.perf_counter()
    d = DeviceArray.from_host(c["x"])
    t1 = time.perf_counter()
    d.download()
    t2 = time.perf_counter()
    d.free()
    nb = c["x"].nbytes
    out["h2d_gbs"], out["d2h_gbs"] = nb / (t1 - t0) / 1e9, nb / (t2 - t1) / 1e9
    out["note"] = ("ms per call, numpy in / numpy out, median of %d after one untimed call; the image crosses PCIe once each way per "
                   "Hessian (%.0f MB), the stateless calls also hash their inputs (plan-cache key)" % (reps, nb / 1e6))
    wg.clear_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", help="C2 (headline), C1, C3, C4, C5 or 'nrow,nchan,npix[,zscale]'")
    ap.add_argument("--epsilon", type=float, default=1e-7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds of CPU work for the baseline sample")
    ap.add_argument("--force", default=None, help="developer knob: 'sigma,W' pins the kernel row")
    ap.add_argument("--verbosity", type=int, default=0)
    ap.add_argument("--reduce-every", type=int, default=0,
                    help="C2: band reduce (RCCL sum-to-root) every this many applies; 0 = once per timed region")
    ap.add_argument("--allow-no-rccl", action="store_true",
                    help="N > 1 only: if the RCCL communicator cannot be created, measure the per-band work without the band "
                         "exchange instead of failing (the line then says so; it is NOT a scaling measurement)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON of rank 0: native libraries (RCCL prints a version banner on fd 1 when
    # a communicator is created) write to stderr for the lifetime of the process, the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from pfb_imaging_amd import _lib
    from pfb_imaging_amd._lib import DeviceArray
    from pfb_imaging_amd.parallel import BandComm
    from pfb_imaging_amd.utils import synth
    from pfb_imaging_amd.wgridder import Gridder

    _lib.require_gpu()  # fail loudly: no CPU path
    # RCCL carries the band exchanges.  A communicator that fails to come up (all ranks agree on that over the TCP
    # rendezvous) costs the exchange, not the per-band measurement: it is then reported as skipped in config.parallelism.
    rccl_error = None
    try:
        comm = BandComm.from_env()
    except Exception as e:
        if int(os.environ.get("WORLD_SIZE", "1")) == 1:
            raise
        rccl_error = f"{type(e).__name__}: {e}"
        comm = BandComm.from_env(transport="host", set_device=False)
        _lib.check(_lib.lib().pfbhip_set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)))
    if comm.world_size > 1 and comm.min_over_ranks(0.0 if rccl_error else 1.0) == 0.0 and rccl_error is None:
        rccl_error = "RCCL communicator failed on another rank"
    use_rccl = comm.world_size > 1 and rccl_error is None
    if rccl_error is not None:
        if not args.allow_no_rccl:  # a multi-GPU line without its exchange step must not be mistaken for a scaling result
            print(f"[bench rank {comm.rank}] RCCL unavailable: {rccl_error}\n--gpus {args.gpus} measures the band exchange over "
                  "RCCL; pass --allow-no-rccl to time the per-band work alone", file=sys.stderr, flush=True)
            raise SystemExit(3)
        print(f"[bench rank {comm.rank}] RCCL unavailable, band exchange skipped: {rccl_error}", file=sys.stderr, flush=True)
        if comm.transport == "rccl":  # ours came up but a peer's did not: stay off it
            comm.transport = "host"
    rank, world = comm.rank, comm.world_size
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    if args.config == "C4":
        out = bench_c4(args, comm, use_rccl, rccl_error)
    else:
        out = bench_gridder(args, comm, use_rccl, rccl_error, synth, Gridder, DeviceArray, _lib)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    comm.barrier()
    comm.close()


def bench_gridder(args, comm, use_rccl, rccl_error, synth, Gridder, DeviceArray, _lib):
    rank, world = comm.rank, comm.world_size
    solve = args.config == "C3"
    cfg = "C2" if solve else args.config
    # ---- this rank's band -------------------------------------------------
    # C5 on N > 1 GPUs: ONE band split by |w| over the ranks (parallel.WShardedGridder: every rank stacks only the w-planes of
    # its range; the partial images are all-reduced inside every apply) -- strong scaling of the single-band stress config
    split = cfg == "C5" and world > 1
    if cfg in synth.CONFIGS:
        case = synth.make_config(cfg, band=0 if split else rank, with_vis=(world == 1 and cfg in ("C1", "C2")))
        nrow_c, nchan_c, npix_c, _ = synth.CONFIGS[cfg]
        what = "per-band PCG solve (on-device CG around the exact Hessian) + RCCL reduce of the image" if solve else \
            "exact Hessian apply (degrid+FFT+grid)"
        wl = f"{args.config}: 1 band/GPU, {nrow_c}x{nchan_c} vis, {npix_c}^2 image, {what}, double precision"
        size_txt = f"{npix_c}^2 grid, {nrow_c * nchan_c:.0e} vis/band".replace("e+0", "e")
    else:
        parts = args.config.split(",")
        nrow, nchan, npix = (int(v) for v in parts[:3])
        zscale = float(parts[3]) if len(parts) > 3 else 1e-3
        case = synth.make_case(nrow, nchan, npix, zscale=zscale, seed=rank, with_vis=False)
        wl = f"custom {nrow}x{nchan} vis, {npix}^2 image, antenna z-scale {zscale}"
        size_txt = f"{npix}^2 grid, {nrow * nchan:.0e} vis/band".replace("e+0", "e")
    nx, ny = case["nx"], case["ny"]
    t0 = time.time()
    gkw = dict(npix_x=nx, npix_y=ny, pixsize_x=case["cell"], pixsize_y=case["cell"], center_x=0.0, center_y=0.0, epsilon=args.epsilon,
               flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1, sigma_max=3.0,
               verbosity=args.verbosity,
               force=None if args.force is None else (float(args.force.split(",")[0]), int(args.force.split(",")[1])))
    planes_per_rank = None
    if split:
        from pfb_imaging_amd.parallel import WShardedGridder

        wsg = WShardedGridder(comm, case["uvw"], case["freq"], case["mask"], **gkw)
        wsg.set_weights(case["wgt"])
        g = wsg.local
        planes_per_rank = wsg.planes_per_rank()
        wl = wl.replace("1 band/GPU", f"ONE band split by |w| over {world} GPUs")
    else:
        g = Gridder(case["uvw"], case["freq"], case["mask"], **gkw)
        g.set_weights(case["wgt"])
    t_plan = time.time() - t0
    info = g.info
    wsum = float(case["wgt"][case["mask"] != 0].sum())
    x_dev = DeviceArray.from_host(case["x"])
    out_dev = DeviceArray((nx, ny), np.float64)
    red_dev = DeviceArray((nx, ny), np.float64) if (world > 1 and rank == 0) else None
    eta = 1e-3  # Tikhonov term of the wsum-normalised operator (C3); the plain apply uses 0

    if solve:
        # One timed region = ONE solve of exactly K CG iterations per band (tol = 0, minit = maxit = K; everything in HBM)
        # + the band reduce of the result.  Warm-up: W-iteration solves.
        def run(k):
            g.cg_dev(x_dev, out_dev, eta=eta, wsum=wsum, tol=0.0, maxit=k, minit=k)
            if use_rccl:
                comm.reduce_sum_dev(out_dev, red_dev, root=0)

        if args.warmup > 0:
            run(args.warmup)
        reduce_every = args.steps
    else:
        reduce_every = args.reduce_every if args.reduce_every > 0 else max(args.steps, 1)
        nstep = [0]

        def step():
            g.hessian_dev(x_dev, out_dev, eta=0.0, wsum=wsum)
            nstep[0] += 1
            if split:  # the partial images of the |w| ranges are summed inside EVERY apply (the operator's own exchange step)
                if use_rccl:
                    comm.allreduce_sum_dev(out_dev, out_dev)
            elif use_rccl and nstep[0] % reduce_every == 0:
                comm.reduce_sum_dev(out_dev, red_dev, root=0)

        for _ in range(args.warmup):
            step()
        if use_rccl:  # warm the communicator (first-call setup is not part of a step)
            if split:
                comm.allreduce_sum_dev(out_dev, out_dev)
            else:
                comm.reduce_sum_dev(out_dev, red_dev, root=0)
        nstep[0] = 0
    comm.barrier()
    _lib.check(_lib.lib().pfbhip_synchronize())
    # Stage timers (HIP events on the handle's stream) run inside the timed region -- except on small plans (C1), whose applies
    # are replayed from a captured hipGraph unless a timer wants events between the kernels: those are timed clean and
    # profiled in a second pass of the same steps.
    two_pass = (not solve) and g.nactive < 2_000_000
    g.profile(not two_pass)
    g.profile_get(reset=True)
    t0 = time.perf_counter()
    if solve:
        run(args.steps)
    else:
        for _ in range(args.steps):
            step()
    _lib.check(_lib.lib().pfbhip_synchronize())
    comm.barrier()
    elapsed = time.perf_counter() - t0
    if two_pass:
        g.profile(True)
        g.profile_get(reset=True)
        for _ in range(args.steps):
            step()
        _lib.check(_lib.lib().pfbhip_synchronize())
    stages = g.profile_get(reset=True)
    g.profile(False)
    elapsed = comm.max_over_ranks(elapsed)
    total_active = comm.sum_over_ranks(g.nactive)
    ms_per_step = elapsed / args.steps * 1e3
    out = None
    if rank == 0:
        b_apply, per_launch, launches, other_bytes = algorithmic_bytes(info, nx, ny, case["uvw"].shape[0], g.nactive)
        dom = max((s for s in stages if s in per_launch and stages[s][1]), key=lambda s: stages[s][0])
        dom_ms, dom_calls = stages[dom]
        avg_ms = dom_ms / max(dom_calls, 1)
        achieved = per_launch[dom] / (avg_ms * 1e-3) / 1e9
        stage_ms = {s: round(v[0] / args.steps, 3) for s, v in stages.items()}
        stage_launches = {s: v[1] / args.steps for s, v in stages.items()}
        actual = sum(per_launch[s] * stages[s][1] / args.steps for s in per_launch if stages[s][1]) + other_bytes
        apply_s = elapsed / args.steps
        pmc, pmc_file = {}, None
        import glob

        # the newest tracked profile: by the collection time written into the file (mtimes do not survive a checkout)
        best = None
        for fn in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
            try:
                cand = json.load(open(fn))
            except Exception:
                continue
            stamp = (float(cand.get("_collected_unix", 0.0)), os.path.basename(fn))
            if best is None or stamp > best[0]:
                best = (stamp, fn, cand)
        if best is not None:
            pmc = best[2]
            pmc_file = os.path.relpath(best[1], ROOT) + " (separate rocprofv3 --pmc passes of this command)"
        names = dict(KERNEL_OF)
        if info["wmode"] == 2:
            names.update({"grid": "k_grid_wd", "degrid": "k_degrid_wd"})
            if os.environ.get("PFBHIP_PAD_PERSIST", "1") != "0" and info["fft_mode"] & 8:
                names["pad_fft"] = "k_fused_pad_fft_p"   # the single plane's persistent pad kernel (csrc/rowfft.hip)
        elif info["scatter_mode"] != 2:
            names["grid"] = "k_grid_blk" if info["scatter_mode"] == 1 else "k_grid_mp"
        if not info["fft_mode"] & 8:
            names["fft_rows"] = "k_rowfft_plain"
        scatter_txt = {2: (f"record-driven register footprint ({info['W'] + info.get('scatter_block', 4) - 1}-cell frame on {info.get('scatter_block', 4)} x {info.get('scatter_block', 4)}-cell blocks), "
                           f"{info['nderiv']} kernel functions per axis (k_grid_wd), " if info["wmode"] == 2 else
                           "record-driven register footprint (k_grid_rec), ") + f"{info['scatter_launches']} launch(es) per pass",
                       1: f"register footprint (k_grid_blk), {info['scatter_launches']} launch(es) per pass",
                       0: "diagonal walk (k_grid_mp)"}[info["scatter_mode"]]
        metric = f"Mvis/s gridded+degridded in exact Hessian applies ({size_txt})" if not solve else \
            f"Mvis/s gridded+degridded inside per-band PCG solves ({size_txt}, one band per GPU, RCCL reduce of the image)"
        out = {
            "metric": metric,
            "value": total_active * 2 / apply_s / 1e6,   # (split: the ranks' active visibilities add up to the one band's)
            "unit": "Mvis/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "hessian_applies_per_s": (1 if split else world) / apply_s,
            "rccl_ranks": world if use_rccl else 0,
            "config": {
                "workload": wl, "bands": 1 if split else world,
                **({"w_planes_per_rank": planes_per_rank, "split": "contiguous |w| ranges (parallel.partition_rows_by_w), one "
                    "all-reduce of the image per apply"} if split else {}),
                "w_scheme": {0: "ES-kernel planes", 1: "polynomial planes",
                             2: f"one plane, the w-term in {info['nderiv']} differentiated kernel functions per axis"}[info["wmode"]],
                "vis_per_band": int(case["uvw"].shape[0] * case["freq"].size), "active_vis_per_band": int(g.nactive),
                "image": [nx, ny], "epsilon": args.epsilon, "grid": [info["nu"], info["nv"]], "occupied_rows": info["occ_rows"],
                "w_planes": info["nplanes"], "kernel_support": info["W"], "scatter": scatter_txt,
                "gather": "row walk, DPP-broadcast FMAs (k_degrid_wd)" if info["wmode"] == 2 else "row walk, DPP-broadcast FMAs (k_degrid_rw)" if (info["scatter_mode"] == 2 or info["nplanes"] <= 4 and info["wmode"] == 1) else "diagonal walk (k_degrid_mp)",
                "plane_transform": ("own row FFT" if info["fft_mode"] & 1 else "rocFFT rows") +
                (" with the transposes folded in" if info["fft_mode"] & 8 else "") +
                (" + fused second axis" if info["fft_mode"] & 2 else
                 (" + own second axis (unfused)" if info["fft_mode"] & 4 else " + rocFFT second axis")),
                "sigma": info["sigma"],
                "parallelism": (f"one band, |w| ranges x{world} + 1 RCCL all-reduce of the image per apply" if split and use_rccl else
                                f"one band, |w| ranges x{world} (image all-reduce skipped: {rccl_error})" if split else
                                f"band-per-gpu x{world}") + ("" if split else (
                    (f" + 1 RCCL sum-to-root of the image per {'solve' if solve else str(reduce_every) + ' applies'}") if use_rccl else
                    (f" (band reduce skipped: {rccl_error})" if rccl_error else ""))),
                "plan_seconds": round(t_plan, 2),
            },
            "roofline": {
                "bound": "hbm", "kernel": names.get(dom, dom), "stage": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes per launch of this stage from the PMC counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
                # this same command (tools/final_profile.sh -> tools/pmc_traffic.py, gfx950 FETCH_SIZE x 2 correction), read from
                # the newest tracked profile of the default workload; null for other workloads or when no profile is tracked
                # (only a profile taken on the SAME kernel: the file names the kernel of every stage it measured)
                "traffic": pmc.get(dom) if (cfg == "C2" and not solve and pmc.get("_kernels", {}).get(dom) == names.get(dom)) else None,
                "traffic_profile": pmc_file if pmc.get("_kernels", {}).get(dom) == names.get(dom) else None,
                "alg_bytes_per_launch": per_launch[dom], "avg_launch_ms": avg_ms, "launches": dom_calls,
                "actual_bytes_per_apply": actual,
                "actual_frac": actual / apply_s / 1e9 / HBM_PEAK_GBS,
                "actual_note": "sum over stages of (compulsory bytes per launch of the pruned pipeline x launches) + the plane clear; "
                               + ("per CG iteration, the CG vector kernels (~10 image passes) not counted" if solve else "per apply"),
                "apply_alg_bytes": b_apply,
                "apply_frac": b_apply / apply_s / 1e9 / HBM_PEAK_GBS,
                "apply_frac_note": "SURVEY.md section 8(d) UNPRUNED accounting (every plane row and the full second axis); the "
                                   "pipeline that runs moves actual_bytes_per_apply",
                "stage_kernels": {s_: names.get(s_, s_) for s_ in per_launch if stages[s_][1]},
                "stage_ms_per_step": stage_ms,
                "stage_launches_per_step": stage_launches,
                "stage_achieved_gbs": {s: round(per_launch[s] / (stages[s][0] / max(stages[s][1], 1) * 1e-3) / 1e9, 1)
                                       for s in per_launch if stages[s][1]},
            },
        }
        if dom == "grid" and info["scatter_mode"] in (1, 2):
            # Register-footprint scatters: no LDS atomic per tap; what a visibility costs is VALU
            # issue -- per held cell one product and 2 FMAs per plane, NR (- 1 for the record kernel at W = 16) cells per
            # lane, plus the kernel evaluation -- at 4 cycles per wave64 f64 instruction on each of the 1024 SIMDs.
            ngroups = -(-info["nplanes"] // 4)
            kp = info["nplanes"] / ngroups
            nr = -(-(info["W"] + 3) // 3)
            if info["scatter_mode"] == 2 and (info["W"] + 3) % 3 == 1:
                nr -= 1
            if info["wmode"] == 2:
                # K complex FMAs per held cell, the K column sums S_r (K (K + 1) real operations + the binomial multiples), and the
                # kernel polynomials: a 16 x 16-cell frame on 4 x 16 lanes (W + block edge - 1 <= 16: 4 cells per lane) with the K
                # Horner chains of degree 12, 10, 8 run for two visibilities at a time (K <= 3); else 3 x 20 lanes and two rounds per
                # visibility (degree 12, and 8 for the fourth / sixth derivatives)
                K = info["nderiv"]
                frame = info["W"] + info.get("scatter_block", 4) - 1
                if frame <= 16:
                    nr = -(-frame // 4)
                horner = (sum(12 - 2 * k for k in range(K)) + 1) / 2 if (frame <= 16 and K <= 3) else 14 + (9 if K > 2 else 0)
                f64_ops = nr * 2 * K + (K * (K + 1) + K - 2) + horner
            else:
                f64_ops = nr * (1 + 2 * kp) + 14
            nsl = max(int(info["scatter_launches"]), 1)
            clock_ghz = 2.0   # what the chip holds under f64 load (tools/ubench.cpp: clock64 / wall = 1.96 GHz), not the 2.4 GHz peak
            floor_ms = g.nactive * f64_ops * 4 / (256 * 4) / (clock_ghz * 1e9) * 1e3 / nsl   # per launch
            # The stage is bound by f64 VALU issue, not by HBM.  Headline figures of the object (count what is
            # USEFUL, not what is issued): achieved = the footprint's multiply-adds only -- W^2 cells x (K kernel functions, or the
            # planes a visibility touches) x (re, im) x 2 flop per visibility; kernel evaluation, the column sums and the zero cells of
            # the lane frame are not counted -- / the launch time, against the f64 peak (vector and matrix: the same
            # 78.6 TFLOP/s on this part; the kernel issues v_fma_f64).  What the kernel must ISSUE, padding included, is in
            # `limiter`; the HBM figures the contract defines are in `hbm`.
            r = out["roofline"]
            r["hbm"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"]}
            terms = info["nderiv"] if info["wmode"] == 2 else (min(info["W"], info["nplanes"]) if info["wmode"] == 0 else info["nplanes"])
            useful = g.nactive * (info["W"] ** 2) * terms * 4.0 / (nsl * ngroups)   # per launch: the apply's total over its scatter launches
            issued = g.nactive / nsl * f64_ops * 128.0
            r.update({"bound": "valu_f64", "achieved": useful / (avg_ms * 1e-3) / 1e12, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": useful / (avg_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                      "flops_note": f"useful flops: {info['W']}^2 cells x {terms} terms x 4 per visibility (footprint multiply-adds only)"})
            out["roofline"]["limiter"] = {
                "bound": "valu_f64", "f64_wave_instr_per_vis": f64_ops, "cycles_per_instr": 4, "simds": 1024,
                "clock_ghz": clock_ghz, "launches_per_pass": nsl, "floor_ms": floor_ms, "frac": floor_ms / avg_ms,
                "issued_tflops": issued / (avg_ms * 1e-3) / 1e12, "issued_frac_of_peak": issued / (avg_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                "note": "f64 FMA/MUL wave instructions the kernel must issue per visibility (padding lanes, kernel evaluation and column "
                        "sums included) at 4 cycles each on 1024 SIMDs at the clock held under f64 load; scalar / LDS / wait "
                        "instructions cost issue slots too (tools/stamp_scatter.py: in-kernel phase stamps)",
            }
        if world == 1 and not args.no_host_path and cfg in ("C1", "C2") and not solve:
            try:
                out["host_path"] = host_path(case, g, wsum)
            except Exception as e:
                out["host_path"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(case, g.oracle_params(), args.cpu_budget)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mvis/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
    g.close()
    return out


def bench_c4(args, comm, use_rccl, rccl_error):
    """BASELINE config C4: the SARA backward step on 4 bands of 4096^2 (PSF 8192^2), bands sharded b % N."""
    from pfb_imaging_amd import prox
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.operators.hessian import HessPSF, HessTreeRay
    from pfb_imaging_amd.operators.psi import PsiNocopyt
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad

    rank, world = comm.rank, comm.world_size
    nband, n, npsf = 4, 4096, 8192
    bases, nlevel = ("self", "db1", "db2", "db3"), 3
    rng = np.random.default_rng(4)  # same data on every rank
    ky = np.fft.fftfreq(npsf)[:, None] ** 2
    kx = np.fft.rfftfreq(npsf)[None, :] ** 2
    psfhat = np.stack([np.exp(-(ky + kx) * (3.0e4 + 1.0e4 * b)) for b in range(nband)])
    eta = np.full(nband, 0.05)
    model = np.abs(rng.standard_normal((nband, n, n))) * (rng.random((nband, n, n)) > 0.99)
    xtilde = model + 0.1 * rng.standard_normal(model.shape)
    psi = PsiNocopyt(nband, n, n, bases, nlevel, 1)
    reg = L21(psi, bases, nu=np.sqrt(len(bases)))
    if world == 1:
        hess = HessPSF(n, n, psfhat, beam=None, eta=eta)
    else:
        pool = BandWorkerPool(nband, comm=comm if use_rccl else None)
        parts = [[dict(psfhat=psfhat[b][None], beam=np.ones((1, n, n)), wsum=np.ones(1))] for b in range(nband)]
        hess = HessTreeRay(parts, n, n, npsf, npsf, etas=eta, wsums=np.ones(nband), workers=pool)

    def run(k):
        pd = PrimalDual(tol=0.0, maxit=k, verbosity=0, gamma=1.0, primal_prox=prox.positivity_prox(1))
        pd.setup(reg, 1.0 + eta.max())
        pd.set_grad(PsfGrad(hess, xtilde, 1.0))
        if pd._device_path() is None:
            raise SystemExit("C4: the device-resident primal-dual loop is not available for this layout")
        pd.solve(model.copy(), 0.01)
        return pd.last

    if args.warmup > 0:
        run(args.warmup)
    comm.barrier()
    t0 = time.perf_counter()
    last = run(args.steps)
    comm.barrier()
    wall = comm.max_over_ranks(time.perf_counter() - t0)
    loop_s = comm.max_over_ranks(float(last.get("loop_ms", 0.0)) * 1e-3) or wall
    out = None
    if rank == 0:
        per = loop_s / args.steps
        # ---- roofline of the dominant stage (stage clocks of pfbhip_primal_dual: HIP events on the loop's stream) ----
        I = n * n * 8
        nyo2 = npsf // 2 + 1
        nwav = sum(1 for b in bases if b != "self")
        levels = sum(0.25**l for l in range(nlevel))
        # compulsory bytes per bracketed launch:
        #  psf_hessian  one PSF-approximate Hessian apply of one band, pruned pipeline: x, beam read + T2 written | T2, |psfhat|
        #               read + T1 written | T1, beam, x read + out written  = 6 I + 4 nx nyo2 16 + nxp nyo2 8
        #  psi_analysis Psi^H of every local band: identity 2 I; per wavelet basis and level two passes of ~2 I / 4^l read + written
        #  psi_synthesis one band: the mirror image (+ the image written once)
        #  dual_update  vp, v read, v, vext written over the coefficient cubes (+ the weights once)
        #  primal_step  x, xp, xout read, x written
        cube = len(bases) * psi.nxmax * psi.nymax * 8
        nloc = len([b for b in range(nband) if b % world == rank])
        stage_bytes = {
            "psf_hessian": 6 * I + 4 * n * nyo2 * 16 + npsf * nyo2 * 8,
            "psi_analysis": nloc * (2 * I * (len(bases) - nwav) + nwav * 4 * I * levels + cube),
            "psi_synthesis": 2 * I * (len(bases) - nwav) + nwav * 4 * I * levels + I,
            "dual_update": nloc * 4 * cube + cube,
            "primal_step": nloc * 4 * I,
        }
        stages = last.get("stages", {})
        timed = {k: v for k, v in stages.items() if v[1] > 0}
        roof = None
        if timed:
            dom = max(timed, key=lambda k: timed[k][0])
            avg_ms = timed[dom][0] / timed[dom][1]
            ach = stage_bytes[dom] / (avg_ms * 1e-3) / 1e9
            Xr, Xc = npsf * npsf * 8, npsf * nyo2 * 16
            roof = {"bound": "hbm", "stage": dom,
                    "kernel": {"psf_hessian": "k_psf_rows_fwd + k_psf_cols + k_psf_rows_inv", "psi_analysis": "k_dwt_rows / k_dwt_cols",
                               "psi_synthesis": "k_idwt_cols / k_idwt_rows", "dual_update": "k_l21_fused", "primal_step": "k_pd_step"}[dom],
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "alg_bytes_per_launch": stage_bytes[dom], "avg_launch_ms": avg_ms, "launches": timed[dom][1],
                    "survey_bytes_per_launch": (6 * I + 3 * Xr + 6.5 * Xc) if dom == "psf_hessian" else None,
                    "survey_note": "SURVEY.md section 8(d) UNPRUNED B_psf = 6 I + 3 Xr + 6.5 Xc (full padded r2c / c2r); the pipeline "
                                   "that runs never stores outside the nx x (nyp/2+1) corner: alg_bytes_per_launch",
                    "stage_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in stages.items()},
                    "stage_launches_per_step": {k: v[1] / args.steps for k, v in stages.items()},
                    "stage_achieved_gbs": {k: round(stage_bytes[k] / (v[0] / v[1] * 1e-3) / 1e9, 1) for k, v in timed.items()},
                    "iteration_bytes": sum(stage_bytes[k] * v[1] / args.steps for k, v in timed.items()),
                    "iteration_frac": sum(stage_bytes[k] * v[1] / args.steps for k, v in timed.items()) / per / 1e9 / HBM_PEAK_GBS}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline_c4(n, npsf, bases, nlevel, psfhat[0], eta[0], xtilde[0], model[0])
            except Exception as e:
                cpu = {"value": None, "unit": "Hessian-applies/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        out = {
            "metric": "PSF-approximate Hessian applies/s inside the SARA primal-dual step (4 bands, 4096^2, Psi: self+db1+db2+db3, 3 levels)",
            "value": nband / per, "unit": "Hessian-applies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "rccl_ranks": world if use_rccl else 0,
            "config": {"workload": "C4: SARA backward step, 4 bands x 4096^2, PSF 8192^2, a step = one primal-dual iteration "
                                   "(Psi^H, l21 dual update, Psi, PSF Hessian, primal step, positivity)",
                       "bands": nband, "image": [n, n], "psf": [npsf, npsf], "bases": list(bases), "nlevels": nlevel,
                       "clock": "iteration loop of pfbhip_primal_dual (cubes resident in HBM); wall time of the call incl. "
                                f"uploads / downloads: {wall / args.steps * 1e3:.2f} ms per iteration",
                       "parallelism": f"bands b % {world}" + (" + 1 RCCL all-reduce of the band sum and 1 of the norms per iteration"
                                                              if use_rccl else (f" (RCCL skipped: {rccl_error})" if rccl_error else ""))},
        }
        if roof is not None:
            out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
    return out


def cpu_baseline_c4(n, npsf, bases, nlevel, psfhat0, eta0, xtilde0, model0):
    """The oracle's restatement (oracle/psi.py + oracle/fftconv.py: numpy, one thread) of ONE primal-dual iteration on ONE of
    the four bands -- a bounded sample; the bands are independent except for the l21 band sum, so the four-band iteration is
    four times this."""
    from oracle import fftconv
    from oracle import psi as opsi

    o = opsi.Psi(1, n, n, bases, nlevel)
    x = model0[None].copy()
    v = np.zeros((1, o.nbasis, o.nxmax, o.nymax))
    vp = np.zeros_like(v)
    w = np.ones(v.shape[1:])
    xout = np.zeros_like(x)
    t0 = time.perf_counter()
    o.dot(x, v)
    opsi.dual_update(vp, v, 0.01, 0.5, w)
    vext = 2.0 * v - vp
    o.hdot(vext, xout)
    xout -= fftconv.hess_psf_dot(xtilde0[None] - x, psfhat0[None], npsf, beam=None, eta=eta0)
    xn = x - 0.3 * xout
    opsi.positivity(xn)
    float(np.sqrt(((xn - x) ** 2).sum() / max((xn**2).sum(), 1e-12)))
    t = time.perf_counter() - t0
    return {"value": 1.0 / t, "unit": "Hessian-applies/s", "cores": 1, "kind": "port",
            "sample": f"one primal-dual iteration of one band (of 4): Psi^H, dual update, Psi, PSF Hessian, primal step, positivity; "
                      f"{t:.1f} s measured; numpy restatement on one thread -- the repository's oracle, not the reference's numba / ducc0 speed",
            "sec_per_band_iteration": t}


if __name__ == "__main__":
    main()
