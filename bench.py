#!/usr/bin/env python3
"""Headline benchmark: exact Hessian apply (degrid + grid with w-stacking and 2-D FFTs) on
BASELINE.json's configs[1]: 1 band, 1e7 synthetic visibilities, 8192^2 image, one band per GPU.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one Hessian apply  out = R^H W R x  on every rank's band (inputs resident in HBM:
tile-sorted visibilities, weights, image): the operator inside the per-band CG, which needs no
communication.  When N > 1 the timed region also contains the band sum of the result images -- the
reference's one exchange per major cycle (core/deconv.py:320-321) -- as ONE RCCL sum-to-root per
--reduce-every applies (default: once per K-apply solve).  Rank 0 prints ONE JSON line.

value      = whole-job visibility throughput: N * 2 * nactive / t_step (a Hessian apply touches
             every unmasked visibility twice: degrid + grid), in Mvis/s.
roofline   = the dominant device stage of the apply, timed live with HIP events on the handle's
             stream (pfbhip_gridder_profile): achieved = algorithmic bytes per launch / average
             launch duration (SURVEY.md section 8(d) accounting, restated in DESIGN.md).
cpu_baseline = the oracle's CPU restatement (kind "port"; ducc0 itself is not installable) timed
             on this box's host cores on a bounded sample (a few w-planes of the same workload).
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
# stage timer -> the kernel it brackets (name as rocprofv3 lists it)
KERNEL_OF = {"grid": "k_grid_mp", "degrid": "k_degrid_mp", "fft_rows": "k_rowfft_plain", "pad": "k_b2a", "crop": "k_a2b",
             "fft_crop": "k_fused_fft_crop", "pad_fft": "k_fused_pad_fft"}


def algorithmic_bytes(info, nx, ny, nrow, nactive):
    """compulsory traffic of one Hessian apply and of one launch per stage."""
    Sc, Sr = 16, 8
    G = info["nu"] * info["nv"] * Sc
    I = nx * ny * Sr
    P = info["nplanes"]
    b_vis = nactive * (2 * Sc + Sr + 2) + 2 * nrow * 24
    b_grid = P * (12 * G + 3 * I)
    # Per-launch compulsory bytes of the PRUNED pipeline actually run: only the
    # occupied rows of the uv-plane A (nu,nv) are cleared / scattered / transformed, the second axis runs
    # on the cropped, transposed plane B (ny,nu), of which only the occupied columns are read / written.
    # A launch = what the stage timers count (include/pfbhip.h, PFBHIP_NSTAGES).
    occ = info["occ_rows"] / info["nu"]
    B = ny * info["nu"] * Sc
    ngroups = -(-P // 4)          # scatter / gather / fused second-axis launches per direction (<= 4 planes each)
    ppl = P / ngroups             # planes per such launch
    fused = bool(info.get("fft_mode", 0) & 2)
    per_launch = {
        "grid": ppl * (occ * G + nactive * (Sc + 24)),    # per plane: occupied plane rows written once + records read
        "degrid": ppl * (occ * G + nactive * (Sc + 24)),  # per plane: occupied plane rows read once + records read
    }
    if fused:
        per_launch.update({
            "fft_rows": 2 * occ * G,                                  # first axis, one plane: read + write of the occupied rows
            "pad": occ * B + occ * G,                                 # B -> A transpose with zero padding, one plane
            "crop": occ * (ny / info["nv"]) * G + occ * B,            # A -> B transpose with crop, one plane
            # planes read once; image written (first group) or read+written; the last launch also reads the correction
            # image (the finalize is folded into it)
            "fft_crop": ppl * occ * B + I * (2 - 1 / ngroups) + I / ngroups,
            # image and correction image read once per launch (x * corr is formed in the load); occupied columns written per plane
            "pad_fft": 2 * I + ppl * occ * B,
        })
    else:
        per_launch.update({
            "fft_rows": occ * G + B,                                  # one row-FFT pass: 1 read + 1 write (average of the A and B passes)
            "pad": (I + B + occ * B + occ * G) / 2,                   # pad+screen (image -> B) and transpose+pad (B -> occupied A)
            "crop": (occ * (ny / info["nv"]) * G + B + (nx / info["nu"]) * B + 2 * I) / 2,  # A -> B, B -> image RMW
        })
    return b_vis + b_grid, per_launch


def cpu_baseline(case, ginfo, oracle_params, nplanes_sample=2):
    """Oracle (CPU port) timing of `nplanes_sample` w-planes of the same Hessian apply."""
    from oracle import _lib as olib
    from oracle import wgridder as owg

    # threads = this process's CPU share (cgroup quota / affinity), not the host's core count
    nthr = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            nthr = min(nthr, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    nthr = int(os.environ.get("PFB_CPU_THREADS", min(nthr, 16)))
    olib.lib().pfbo_set_num_threads(nthr)
    owg.FFT_WORKERS = nthr
    c = case
    t0 = time.time()
    plan = owg.Plan(c["uvw"], c["freq"], c["mask"], c["nx"], c["ny"], c["cell"], c["cell"], 0.0, 0.0, 1e-7, False,
                    True, False, True, False, params=oracle_params)
    t_plan = time.time() - t0
    P = plan.p.nplanes
    planes = sorted(set(np.linspace(0, P - 1, nplanes_sample).round().astype(int).tolist()))
    swgt = np.ascontiguousarray(c["wgt"], dtype=np.float64).reshape(-1)
    acc_img = np.zeros((c["nx"], c["ny"]))
    sacc = np.zeros(plan.nrow * plan.nchan, dtype=np.complex128)
    dc = np.ascontiguousarray(c["x"])
    t0 = time.time()
    for p in planes:
        plan.plane_round_trip(dc, swgt, p, acc_img, sacc)
    t = time.time() - t0
    t_apply = t / len(planes) * P
    nactive = int(plan.active.sum())
    return {
        "value": 2 * nactive / t_apply / 1e6,
        "unit": "Mvis/s",
        "cores": int(olib.lib().pfbo_num_threads()),
        "kind": "port",
        "sample": f"{len(planes)} of {P} w-planes of the same apply (all {nactive} vis, {c['nx']}^2 image, "
                  f"grid {plan.p.nu}x{plan.p.nv}); {t:.1f} s measured, scaled by {P}/{len(planes)}",
        "sec_per_apply_est": t_apply,
        "plan_sec": t_plan,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", help="C2 (headline), C1, C4 or 'nrow,nchan,npix'")
    ap.add_argument("--epsilon", type=float, default=1e-7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-planes", type=int, default=2)
    ap.add_argument("--force", default=None, help="developer knob: 'sigma,W' pins the kernel row")
    ap.add_argument("--verbosity", type=int, default=0)
    ap.add_argument("--reduce-every", type=int, default=0,
                    help="band reduce (RCCL sum-to-root) every this many applies; 0 = once per timed region")
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
    # RCCL carries the band reduce.  The timed Hessian applies need no communication, so a communicator that fails to
    # come up (all ranks agree on that over the gloo rendezvous) costs the reduce, not the measurement: it is then
    # reported as skipped in config.parallelism.
    rccl_error = None
    try:
        comm = BandComm.from_env()
    except Exception as e:
        if int(os.environ.get("WORLD_SIZE", "1")) == 1:
            raise
        rccl_error = f"{type(e).__name__}: {e}"
        comm = BandComm.from_env(transport="gloo", set_device=False)
        _lib.check(_lib.lib().pfbhip_set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)))
    if comm.world_size > 1 and comm.min_over_ranks(0.0 if rccl_error else 1.0) == 0.0 and rccl_error is None:
        rccl_error = "RCCL communicator failed on another rank"
    use_reduce = comm.world_size > 1 and rccl_error is None
    if rccl_error is not None:
        print(f"[bench rank {comm.rank}] RCCL unavailable, band reduce skipped: {rccl_error}", file=sys.stderr, flush=True)
        if comm.transport == "rccl":  # ours came up but a peer's did not: stay off it
            comm.transport = "gloo"
    rank, world = comm.rank, comm.world_size
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # ---- this rank's band -------------------------------------------------
    if args.config in synth.CONFIGS:
        case = synth.make_config(args.config, band=rank)
        wl = f"{args.config}: 1 band/GPU, {synth.CONFIGS[args.config][0]}x{synth.CONFIGS[args.config][1]} vis, " \
             f"{synth.CONFIGS[args.config][2]}^2 image, exact Hessian apply (degrid+FFT+grid), double precision"
    else:
        parts = args.config.split(",")
        nrow, nchan, npix = (int(v) for v in parts[:3])
        zscale = float(parts[3]) if len(parts) > 3 else 1e-3
        case = synth.make_case(nrow, nchan, npix, zscale=zscale, seed=rank)
        wl = f"custom {nrow}x{nchan} vis, {npix}^2 image, antenna z-scale {zscale}"
    nx, ny = case["nx"], case["ny"]
    t0 = time.time()
    g = Gridder(case["uvw"], case["freq"], case["mask"], npix_x=nx, npix_y=ny, pixsize_x=case["cell"],
                pixsize_y=case["cell"], center_x=0.0, center_y=0.0, epsilon=args.epsilon, flip_u=False, flip_v=True,
                flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1, sigma_max=3.0, verbosity=args.verbosity,
                force=None if args.force is None else (float(args.force.split(",")[0]), int(args.force.split(",")[1])))
    g.set_weights(case["wgt"])
    t_plan = time.time() - t0
    info = g.info
    wsum = float(case["wgt"][case["mask"] != 0].sum())
    x_dev = DeviceArray.from_host(case["x"])
    out_dev = DeviceArray((nx, ny), np.float64)
    red_dev = DeviceArray((nx, ny), np.float64) if (world > 1 and rank == 0) else None

    # A step = one exact Hessian apply of this rank's band: the inner operator of the per-band CG, which needs no
    # communication ("CG itself needs no communication").  The band sum of the result image -- the one
    # exchange of the reference's major cycle (core/deconv.py:320-321) -- is ONE RCCL sum-to-root every
    # --reduce-every applies (default: once per timed region, i.e. once per K-apply solve), inside the timed region.
    reduce_every = args.reduce_every if args.reduce_every > 0 else max(args.steps, 1)
    nstep = [0]

    def step():
        g.hessian_dev(x_dev, out_dev, eta=0.0, wsum=wsum)
        nstep[0] += 1
        if use_reduce and nstep[0] % reduce_every == 0:
            comm.reduce_sum_dev(out_dev, red_dev, root=0)

    for _ in range(args.warmup):
        step()
    if use_reduce:  # warm the communicator (first-call setup is not part of a step)
        comm.reduce_sum_dev(out_dev, red_dev, root=0)
    nstep[0] = 0
    comm.barrier()
    _lib.check(_lib.lib().pfbhip_synchronize())
    g.profile(True)
    g.profile_get(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(_lib.lib().pfbhip_synchronize())
    comm.barrier()
    elapsed = time.perf_counter() - t0
    stages = g.profile_get(reset=True)
    g.profile(False)
    elapsed = comm.max_over_ranks(elapsed)
    total_active = comm.sum_over_ranks(g.nactive)
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        b_apply, per_launch = algorithmic_bytes(info, nx, ny, case["uvw"].shape[0], g.nactive)
        # the register-footprint scatter runs one launch per tile colour: each moves a quarter of the pass's bytes
        nsl = max(int(info["scatter_launches"]), 1)
        per_launch["grid"] /= nsl
        dom = max((s for s in stages if s in per_launch and stages[s][1]), key=lambda s: stages[s][0])
        dom_ms, dom_calls = stages[dom]
        avg_ms = dom_ms / max(dom_calls, 1)
        achieved = per_launch[dom] / (avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # the PMC passes were collected on the headline configuration only
        if os.path.exists(tfile) and args.config == "C2" and args.epsilon == 1e-7 and args.force is None:
            try:
                traffic = json.load(open(tfile)).get(dom)
            except Exception:
                traffic = None
        stage_ms = {s: round(v[0] / args.steps, 3) for s, v in stages.items()}
        stage_launches = {s: v[1] / args.steps for s, v in stages.items()}
        out = {
            "metric": "Mvis/s gridded+degridded in exact Hessian applies (8192^2 grid, 1e7 vis/band)",
            "value": total_active * 2 / (elapsed / args.steps) / 1e6,
            "unit": "Mvis/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "hessian_applies_per_s": world / (elapsed / args.steps),
            "config": {
                "workload": wl, "bands": world, "w_scheme": "polynomial planes" if info["wmode"] == 1 else "ES-kernel planes", "vis_per_band": int(case["uvw"].shape[0] * case["freq"].size),
                "active_vis_per_band": int(g.nactive), "image": [nx, ny], "epsilon": args.epsilon,
                "grid": [info["nu"], info["nv"]], "occupied_rows": info["occ_rows"], "w_planes": info["nplanes"],
                "kernel_support": info["W"], "scatter": (f"register footprint (k_grid_blk), {info['scatter_launches']} launch(es) per pass" if info["scatter_mode"] == 1 else "diagonal walk (k_grid_mp)"), "plane_transform": ("own row FFT" if info["fft_mode"] & 1 else "rocFFT rows") +
                (" + fused second axis" if info["fft_mode"] & 2 else
                 (" + own second axis (unfused)" if info["fft_mode"] & 4 else " + rocFFT second axis")),
                "sigma": info["sigma"], "parallelism": f"band-per-gpu x{world}" + (f" + 1 RCCL sum-to-root of the image per {reduce_every} applies" if use_reduce else
                                                                   (f" (band reduce skipped: {rccl_error})" if rccl_error else "")),
                "plan_seconds": round(t_plan, 2),
            },
            "roofline": {
                "bound": "hbm", "kernel": KERNEL_OF.get(dom, dom), "stage": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "alg_bytes_per_launch": per_launch[dom], "avg_launch_ms": avg_ms, "launches": dom_calls,
                "apply_alg_bytes": b_apply,
                "apply_frac": b_apply / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "stage_ms_per_step": stage_ms,
                "stage_launches_per_step": stage_launches,
                "stage_achieved_gbs": {s: round(per_launch[s] / (stages[s][0] / max(stages[s][1], 1) * 1e-3) / 1e9, 1)
                                       for s in per_launch if stages[s][1]},
            },
        }
        if dom == "grid" and info["scatter_mode"] == 1:
            # Register-footprint scatter: no LDS atomic per tap; what a visibility costs is f64
            # VALU work -- per held cell one product and 2 FMAs per plane, NR cells per lane, plus the 13-step Horner
            # chain of the kernel values -- at 4 cycles per wave64 f64 instruction on each of the 1024 SIMDs.
            out["roofline"]["kernel"] = "k_grid_blk"
            ngroups = -(-info["nplanes"] // 4)
            kp = info["nplanes"] / ngroups
            nr = -(-(info["W"] + 3) // 3)
            f64_ops = nr * (1 + 2 * kp) + 13
            floor_ms = g.nactive * f64_ops * 4 / (256 * 4) / 2.4e9 * 1e3 / nsl   # per launch
            out["roofline"]["limiter"] = {
                "bound": "valu_f64", "f64_wave_instr_per_vis": f64_ops, "cycles_per_instr": 4, "simds": 1024,
                "clock_ghz": 2.4, "launches_per_pass": nsl, "floor_ms": floor_ms, "frac": floor_ms / avg_ms,
                "note": "counted f64 FMA/MUL only; rocprofv3 SQ counters (profiles/) give 107 VALU instructions per "
                        "visibility and ~50 % VALU-busy, the rest is LDS flush + tile load/store phases",
            }
        elif dom == "grid":
            # The diagonal-walk scatter is bound by the LDS f64-atomic pipe, not by HBM:
            # 2*16*16 ds_add_f64 lane-operations per visibility and plane, ~12 cycles per wave-instruction.
            ngroups = -(-info["nplanes"] // 4)
            winstr = g.nactive * (info["nplanes"] / ngroups) * 512 / 64          # wave-instructions per launch
            floor_ms = winstr / 256 * 12 / 2.4e9 * 1e3                           # 256 CUs, 2.4 GHz
            out["roofline"]["limiter"] = {
                "bound": "lds_atomic", "wave_instr_per_launch": winstr, "cycles_per_instr": 12, "cus": 256,
                "clock_ghz": 2.4, "floor_ms": floor_ms, "frac": floor_ms / avg_ms,
            }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(case, info, g.oracle_params(), args.cpu_planes)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mvis/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    comm.barrier()
    g.close()
    comm.close()


if __name__ == "__main__":
    main()
