#!/usr/bin/env python3
"""Headline benchmark: Mray/s and frames/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one frame of BASELINE.json configs[1]: generated Cornell box, 1920x1080, 4 spp, fp32, GI on, SVGF on,
TLAS rebuilt per frame as the reference does.  The same run also measures configs[2] (packed fp16, under "fp16") and, at
N = 1, configs[3] (the 262 k-triangle atrium, 1080p, 8 spp, fp16, every BLAS rebuilt inside each frame, under "c4" with
its own roofline and CPU baseline sample).  Inputs (geometry, BVH, materials) are resident in HBM before the timed region
starts.  With N GPUs the frame is cut into N row strips (+ SVGF halo) and completed by one RCCL all-gather per frame
(lprt_amd.dist); total work is fixed, so scaling is "strong".  --gpus N > 1 refuses to run unless torch.distributed
really has N ranks on the nccl (= RCCL) backend: a rehearsal (--rehearsal with --backend gloo / --single-device) is
labelled as one and carries no value.

`roofline` carries two fractions: `frac` = ALGORITHMIC bytes per launch (visit counters x the reference's record sizes,
SURVEY.md 8(d)) / launch time / 8 TB/s -- a work rate in the reference's units, which counts cache-resident bytes at
HBM price -- and `frac_counter` = fabric bytes per launch from the committed rocprofv3 PMC pass of the same workload
(profiles/pmc_traffic.json) / launch time / 8 TB/s, the measured share of the HBM roof.

Rank 0 prints ONE JSON line.
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

import lprt_amd  # noqa: E402
from lprt_amd import abi, scenes, dist as lp_dist  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3e12 achievable)

# ---- algorithmic bytes, reference layout sizes fp32 | fp16 ------------------------------
S_NODE, S_AABB, S_OBJ, S_TRI, S_FALLBACK = (40, 28), (24, 12), (136, 72), (52, 28), (48, 48)
RAY_IO = {  # bytes per ray: command in + record out
    "primary": (76, 44),                 # 7 G-buffer surfaces written
    "trace_di1": (56 + 8 + 12, 36 + 8 + 6),  # TraceLightCommand + skip in, di_intensity out
    "trace_di2": (56 + 8 + 12, 36 + 8 + 6),
    "trace_gi": (28 + 8 + 80, 16 + 8 + 48),  # TraceGICommand + skip in, PixelShaderInput out
}
PIXEL_BYTES = {  # compulsory bytes per pixel per launch of the streaming kernels (both SVGF lobes per launch)
    "shade1": (460, 236), "shade2": (324, 206), "compose": (72, 36), "temporal_map": (100, 80), "rng": (20, 20),
    "svgf_gradient": (40, 20), "svgf_temporal": (152, 81), "svgf_moments": (168, 90), "svgf_wavelet": (120, 64),
    "finalize": (128, 70), "backup": (0, 0),
}


FUSED_TAIL = {"trace_gi": "shade2", "trace_di2": "compose"}  # k_trace_gi shades the hit; the last k_trace_di composes


def trace_bytes(counters, stage, prec):
    i = 0 if prec == abi.F32 else 1
    c = counters
    return (c["tlas_node_visits"] * S_NODE[i] + c["blas_node_visits"] * S_NODE[i] + c["aabbw_tests"] * S_AABB[i] +
            c["object_visits"] * S_OBJ[i] + c["tri_tests"] * S_TRI[i] + c["fallback_tests"] * S_FALLBACK[i] +
            c["rays"] * RAY_IO[stage][i])


# DESIGN.md 4.4 "Negative radiance in long half-precision sequences": reference behaviour, reproduced by the oracle
HALF_OUTPUT_NOTE = ("half precision, reference behaviour reproduced pixel for pixel by the oracle (DESIGN.md 4.4, tests/test_gpu_long_sequences.py): "
                    "the reprojection truncates history positions toward zero, border pixels extrapolate their own history, after ~15 frames "
                    "their filtered radiance is negative and pow(negative, 1/2.2) in the output pass is NaN; LPRT_OPT_FLOOR_REPROJECTION=1 removes it")


def dist_info():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def measure(prec, sc, W, H, spp, steps, warmup, rank, world, device, stage_timing=True, backend="nccl", rebuild_blas=False, reuse_direct=False,
            trace_halo=False, defer_gather=False, sustain_s=2.0):
    """Times `steps` frames; returns a dict (identical on every rank for the reduced fields)."""
    win_b, win_e = lp_dist.window_rows(H, world, rank)
    r = lprt_amd.Renderer(W, H, prec, device=device, row_begin=win_b, row_end=win_e)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    frame = lp_dist.ShardedFrame(W, H, rank, world, torch.device("cuda", device), backend=backend, trace_halo=trace_halo, defer_gather=defer_gather)
    frame.attach(r)  # output rows -> this rank's slot of the gathered image; N > 1: the spp loop runs on the owned rows only
    r.load_scene(sc)
    # BVH build rate, reported apart: BLAS of every primitive, three timed rebuilds after the first
    r.enable_stage_timing(True)
    r.reset_stage_times()
    for _ in range(3):
        r.build_blas()
    r.synchronize()
    build_ms = r.stage_times()["build_blas"][0] / 3
    r.enable_stage_timing(False)
    fs = abi.default_settings(spp=spp, gi_on=True, svgf=True)
    fs.rebuild_blas = int(rebuild_blas)
    fs.reuse_direct_light = int(reuse_direct)

    last = {}

    prepared = r.prepare_scene(sc)  # the C arrays of the (static) RDScene, built once

    def step():
        r.set_scene(prepared=prepared)   # per-frame RDScene hand-over + TLAS rebuild (rt/rtrt/cuda.hpp:662-710)
        # N > 1: the halo rows' sample means and temporal history come from the ranks that own them, one point-to-point
        # batch per frame between the two render phases (dist.py); r enqueues on torch's current stream
        last["hist_bytes"] = frame.render(r, fs, stream_ordered=True)
        img = frame.all_gather()  # deferred mode: the previous frame's image (this frame's gather runs under the next frame)
        if img is not None:
            last["image"] = img

    # rays of one whole frame (full image, rank 0) and per-kernel work counters of THIS rank's launches: one
    # instrumented, untimed frame each
    frame_rays, stage_counters = None, {}
    if rank == 0:
        r.enable_counters(True)
        r.reset_counters()
        r.set_scene(sc)
        r.render(fs)
        r.synchronize()
        for st in ("primary", "trace_di1", "trace_gi", "trace_di2"):
            stage_counters[st] = r.stage_counters(st)
        frame_rays = r.counters()["rays"]
        r.enable_counters(False)
        r.reset_history()
        if world > 1:  # the strip + halo is not the frame: count the rays of the whole image once
            full = lprt_amd.Renderer(W, H, prec, device=device)
            full.load_scene(sc)
            full.enable_counters(True)
            full.reset_counters()
            full.render(fs)
            full.synchronize()
            frame_rays = full.counters()["rays"]
            full.close()
    cdev = f"cuda:{device}" if backend == "nccl" else "cpu"
    if world > 1:
        t = torch.tensor([frame_rays or 0], dtype=torch.int64, device=cdev)
        torch.distributed.broadcast(t, 0)
        frame_rays = int(t.item())

    for _ in range(warmup):
        step()
    # Per-stage breakdown from CALIBRATION frames (untimed, every stage bracketed by HIP events: two event records per
    # launch cost ~0.2 ms per frame in total).  The timed region then instruments only the dominant kernel family, whose
    # live launch duration the roofline needs.
    calib, stage, dominant = {}, {}, None
    if stage_timing:
        r.enable_stage_timing(True)
        r.reset_stage_times()
        n_calib = 3
        for _ in range(n_calib):
            step()
        r.synchronize()
        calib = {k: (v[0] / n_calib, v[1] // n_calib) for k, v in r.stage_times().items() if v[1]}
        cands = {k: v for k, v in calib.items() if k not in ("build_blas", "build_tlas")}
        dominant = max(cands, key=lambda k: cands[k][0]) if cands else None
        r.set_stage_timing_mask([dominant] if dominant else [])
        r.reset_stage_times()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if frame.defer:
        last["image"] = frame.flush()  # inside the timed region: the last frame is only done when it is complete everywhere
    barrier_sync(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # A timed region of K steps can be a few tens of milliseconds -- too short for anything outside this process (the driver's
    # SMI sampler, a wall clock around the run) to see the GPU busy.  When it is under a second the same step is repeated for
    # ~sustain_s more seconds and reported NEXT TO the contract's figure (same work per step; every rank runs the same count).
    sustained = None
    if dt < 1.0 and sustain_s > 0:
        n_sus = max(steps, int(np.ceil(sustain_s / (dt / steps))))
        barrier_sync(world)
        t1 = time.perf_counter()
        for _ in range(n_sus):
            step()
        if frame.defer:
            frame.flush()
        barrier_sync(world)
        dts = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=cdev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": 1e3 * dts / n_sus}
    # what each rank saw of the transport: the communicator it is part of and the bytes it moved per frame
    comm = {"rank": rank, "world_size_observed": torch.distributed.get_world_size() if world > 1 else 1,
            "backend": torch.distributed.get_backend() if world > 1 else None, "device": torch.cuda.get_device_name(device),
            "rows_owned": [frame.begin, frame.end], "rows_rendered": [win_b, win_e],
            "p2p_bytes_per_frame": int(last.get("hist_bytes") or 0),
            "all_gather_bytes_contributed_per_frame": int(frame.bytes_gathered // max(frame.gathers, 1)),
            "all_gathers": int(frame.gathers), "all_gather_deferred": bool(frame.defer)}
    comms = [comm]
    if world > 1:
        comms = [None] * world
        torch.distributed.all_gather_object(comms, comm)
    if stage_timing:
        stage = {k: v for k, v in r.stage_times().items() if v[1]}   # the dominant stage only, over the timed region
        r.enable_stage_timing(False)
    nonfinite_px = int((~torch.isfinite(last["image"]).all(dim=-1)).sum().item())
    out_ok = nonfinite_px == 0 and bool((last["image"][..., 3] == 1).all().item())
    res = dict(dt=dt, ms_per_step=1e3 * dt / steps, frames_per_s=steps / dt, frame_rays=frame_rays,
               mrays_per_s=frame_rays * steps / dt / 1e6, stage=stage, calib=calib, stage_counters=stage_counters,
               finite=out_ok, nonfinite_px=nonfinite_px, window_rows=[win_b, win_e], build_blas_ms=build_ms, sustained=sustained, ranks=comms,
               sample_rows=(win_e - win_b) if frame.trace_halo else (frame.end - frame.begin))
    r.close()
    return res


def workload_key(scene_name, W, H, spp, rebuild_blas):
    """What a set of PMC counters belongs to: counters of one workload are never attached to another."""
    return f"{scene_name} {W}x{H} {spp}spp rebuild_blas={int(bool(rebuild_blas))}"


def pmc_traffic(kernel, prec, wkey):
    """Fabric bytes per launch of `kernel` in workload `wkey` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes over the same frames, FETCH_SIZE doubled as the gfx950 guide prescribes:
    tools/collect_profiles.sh, tools/summarize_profiles.py, tools/merge_pmc.py).  Counters cannot be read from inside the
    timed process, so this is the last collected value for the SAME workload, or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path)).get("workloads", {}).get(wkey, {})
        e = d.get(f"{kernel}[{'f32' if prec == abi.F32 else 'f16'}]")
        if e and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"], f"profiles/pmc_traffic.json [{wkey}] (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    except (OSError, ValueError):
        pass
    return None, f"no PMC pass committed for [{wkey}]"


SAMPLE_STAGES = ("rng", "shade1", "shade2", "compose", "trace_di1", "trace_gi", "trace_di2")  # run on the sample rows only


def roofline_of(res, prec, width, spp, world, wkey):
    """Dominant kernel family (largest HIP-event time over the calibration frames) against the HBM roof; its launch
    duration is the live average over the timed region, where it is the only instrumented stage."""
    i = 0 if prec == abi.F32 else 1
    best = None
    for name, (ms, launches) in res["stage"].items():
        if launches == 0 or name in ("build_blas", "build_tlas"):
            continue
        if best is None or ms > best[1]:
            best = (name, ms, launches)
    if best is None:
        return None
    name, ms, launches = best
    avg_s = ms / launches / 1e3
    npix_window = width * (res["sample_rows"] if name in SAMPLE_STAGES else res["window_rows"][1] - res["window_rows"][0])
    streams = "shade2" in res["calib"]  # GI / round-1 shadow rays ran as ray streams: their tails are kernels of their own
    if name in RAY_IO:
        per_frame = trace_bytes(res["stage_counters"][name], name, prec)
        launches_per_frame = max(1, res["calib"][name][1])  # per frame: one per batch of samples (LPRT_OPT_SAMPLE_BATCH), not one per sample
        bytes_per_launch = per_frame / launches_per_frame
        basis = "visit counters x reference record sizes (SURVEY.md 8(d))"
        if name in FUSED_TAIL and not streams:  # the kernel also runs a streaming stage of the reference
            # shade #2 only has work where a GI ray was cast; the composition touches every pixel
            units = res["stage_counters"][name]["rays"] / launches_per_frame if name == "trace_gi" else npix_window * spp / launches_per_frame
            bytes_per_launch += PIXEL_BYTES[FUSED_TAIL[name]][i] * units
            basis += f" + {FUSED_TAIL[name]} compulsory bytes (SURVEY.md 8(a))"
    else:
        per_frame_launches = max(1, res["calib"][name][1]) if name in res["calib"] else 1
        samples_per_launch = spp / per_frame_launches if name in SAMPLE_STAGES else 1
        bytes_per_launch = PIXEL_BYTES[name][i] * npix_window * samples_per_launch
        basis = "compulsory bytes/pixel x pixels (SURVEY.md 8(a))"
    achieved = bytes_per_launch / avg_s
    kernel = (abi.STREAM_STAGE_KERNELS if streams else {}).get(name) or abi.STAGE_KERNELS.get(name, name)
    # the committed PMC pass is of the N = 1 command: a strip's launches are not comparable
    traffic, traffic_src = pmc_traffic(kernel, prec, wkey) if world == 1 else (None, "collected for N=1 only")
    out = {"bound": "hbm", "kernel": kernel, "stage": name, "achieved": achieved / 1e9,
           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
           "frac_counter": None, "algorithmic_over_traffic": None,
           "avg_launch_us": avg_s * 1e6, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
           "basis": basis,
           "note": "frac prices every byte a ray touches at the reference's record sizes, cache-resident or not; frac_counter is the "
                   "measured share of the HBM roof (fabric bytes / launch time / peak). Traversal is bound by dependent fetches and "
                   "instruction issue, not by either"}
    if traffic:
        out["frac_counter"] = traffic / avg_s / HBM_PEAK
        out["algorithmic_over_traffic"] = bytes_per_launch / traffic
        if out["frac_counter"] < 0.3:
            # the fabric moves under 30 % of what the roof allows during this kernel: HBM is the roof `frac` is priced against
            # (the contract's figure), not what bounds the kernel
            out["bound"] = "latency/issue"
            out["roof"] = "hbm"
    return out


def cpu_baseline(sc, spp, prec=abi.F32, W=1920, H=1080, budget_s=12.0):
    """The CPU oracle (a port of the reference pipeline; the reference has no CPU path) on a bounded sample of the
    same workload: whole frames of W x H until ~budget_s of CPU work have been spent (at most 16 frames).  Mray/s does not
    depend on the frame size, so the large scene is sampled at a reduced one."""
    from oracle.binding import Oracle
    cores = min(len(os.sched_getaffinity(0)), 32)  # the GPU box grants a CPU share; more threads than that only thrash
    with Oracle(W, H, prec, threads=cores) as o:
        o.load_scene(sc)
        fs = abi.default_settings(spp=spp, gi_on=True, svgf=True)
        o.reset_counters()
        frames, t0 = 0, time.perf_counter()
        while frames < 16 and (frames == 0 or time.perf_counter() - t0 < budget_s):
            o.set_scene(sc)
            o.render_frame(fs)
            frames += 1
        dt = time.perf_counter() - t0
        rays = o.counters()["rays"]
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"{sc.name} {W}x{H}, {spp} spp, {'fp32' if prec == abi.F32 else 'fp16'}, gi+svgf on, {frames} frame(s), "
                      f"{dt:.1f} s of OpenMP oracle (BVH built once, outside the sample)",
            "frames_per_s": frames / dt}


def workload_text(sc, W, H, spp, prec, rebuild_blas, reuse_direct):
    return (f"{sc.name} {W}x{H} {spp}spp {'fp32' if prec == abi.F32 else 'fp16'} gi=on svgf=on traced-primary tlas-per-frame "
            f"rebuild_blas={int(bool(rebuild_blas))}" + (" reuse_direct_light=1" if reuse_direct else ""))


def baseline_config(sc, W, H, spp, prec, rebuild_blas, reuse_direct):
    """Which BASELINE.json config the arguments of this run are, if any (a label is only attached to the workload it names)."""
    if reuse_direct or (W, H) != (1920, 1080):
        return None
    if sc.name == "cornell" and spp == 4 and not rebuild_blas:
        return "configs[1]" if prec == abi.F32 else "configs[2]"
    if sc.name == "sponza_like" and spp == 8 and prec == abi.F16 and rebuild_blas:
        return "configs[3]"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--scene", default="cornell", help="generated scene name or a .gltf/.glb path")
    ap.add_argument("--no-fp16", action="store_true")
    ap.add_argument("--rebuild-blas", action="store_true", help="rebuild every BLAS inside each frame (BASELINE configs[3])")
    ap.add_argument("--reuse-direct-light", action="store_true",
                    help="cast the sample-invariant shadow rays of round 0 once per frame (same image, more frames/s, fewer executed rays)")
    ap.add_argument("--trace-halo", action="store_true",
                    help="N > 1: every rank also traces its 68 halo rows instead of receiving their sample means from the neighbours")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL over xGMI) | gloo (needs --rehearsal: gathers through host memory)")
    ap.add_argument("--single-device", action="store_true", help="needs --rehearsal: every rank renders on GPU 0")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 without N GPUs on RCCL (code-path rehearsal): the JSON line says so and carries value = null")
    ap.add_argument("--defer-gather", action="store_true",
                    help="N > 1: start each frame's all-gather asynchronously and let it run under the next frame (dist.ShardedFrame(defer_gather=True))")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="when the K timed steps take under a second, repeat the step for this long and report sustained_ms_per_step beside it (0: off)")
    ap.add_argument("--no-c4", action="store_true", help="skip the configs[3] leg (atrium, 8 spp, fp16, BLAS rebuilt per frame)")
    args = ap.parse_args()

    rank, world, local = dist_info()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(no JSON line is printed for a run that is not what it says)")
    if (args.backend != "nccl" or args.single_device) and not args.rehearsal:
        raise SystemExit("--backend other than nccl / --single-device are rehearsals of the N > 1 code path, not measurements: pass --rehearsal")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(args.backend)
        if torch.distributed.get_world_size() != args.gpus or (torch.distributed.get_backend() != "nccl" and not args.rehearsal):
            raise SystemExit(f"process group has {torch.distributed.get_world_size()} ranks on {torch.distributed.get_backend()}, "
                             f"not {args.gpus} on nccl")
        if not args.rehearsal and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} visible GPUs, found {torch.cuda.device_count()}")

    if args.scene in scenes.SCENES:
        sc = scenes.SCENES[args.scene]()
    elif os.path.exists(args.scene):  # a .gltf / .glb file, e.g. the real Sponza when the operator supplies it
        from lprt_amd import gltf
        sc = gltf.load_gltf(args.scene)
    else:
        raise SystemExit(f"unknown scene {args.scene!r}: one of {sorted(scenes.SCENES)} or a path to a .gltf/.glb file")
    W, H, spp = args.width, args.height, args.spp
    extra = dict(defer_gather=args.defer_gather, sustain_s=args.sustain_seconds)
    res32 = measure(abi.F32, sc, W, H, spp, args.steps, args.warmup, rank, world, local, not args.no_stage_timing, args.backend, args.rebuild_blas,
                    args.reuse_direct_light, args.trace_halo, **extra)
    res16 = None if args.no_fp16 else measure(abi.F16, sc, W, H, spp, args.steps, args.warmup, rank, world, local,
                                               not args.no_stage_timing, args.backend, args.rebuild_blas, args.reuse_direct_light, args.trace_halo, **extra)
    # configs[3] rides along at N = 1 in the default run: the only workload in which traversal of a large scene is the cost
    res_c4, sc_c4 = None, None
    if world == 1 and not args.no_c4 and args.scene == "cornell" and (W, H, spp) == (1920, 1080, 4) and not args.rebuild_blas:
        sc_c4 = scenes.SCENES["sponza_like"]()
        res_c4 = measure(abi.F16, sc_c4, 1920, 1080, 8, args.steps, args.warmup, rank, world, local, not args.no_stage_timing, args.backend, True,
                         False, False, sustain_s=args.sustain_seconds)
    if rank == 0:
        wkey = workload_key(sc.name, W, H, spp, args.rebuild_blas)
        bc32, bc16 = (baseline_config(sc, W, H, spp, p_, args.rebuild_blas, args.reuse_direct_light) for p_ in (abi.F32, abi.F16))
        out = {
            "metric": "Mray/s at 1920x1080, 4 spp (frames/s alongside), fp32 headline + fp16, Cornell-box glTF-equivalent scene",
            "value": res32["mrays_per_s"], "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res32["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(sc, W, H, spp, abi.F32, args.rebuild_blas, args.reuse_direct_light),
                       "baseline_config": bc32, "triangles": sc.n_tris, "objects": len(sc.objects), "rays_per_frame": res32["frame_rays"],
                       "parallelism": (f"row-strips x{world} + halo {lp_dist.HALO_ROWS} ({'traced locally' if args.trace_halo else 'samples from the owners'})"
                                       " + all-gather") if world > 1 else "single GPU"},
            "frames_per_s": res32["frames_per_s"],
            "sustained_ms_per_step": res32["sustained"]["ms_per_step"] if res32["sustained"] else None, "sustained": res32["sustained"],
            "ranks": res32["ranks"],
            "mrays_per_s_trace_kernels_only": (res32["frame_rays"] / 1e3 / sum(res32["calib"][k][0] for k in ("primary", "trace_di1", "trace_gi", "trace_di2")
                                                                                  if k in res32["calib"])) if res32["calib"] else None,
            "build_blas_ms": res32["build_blas_ms"], "build_mtris_per_s": sc.n_tris / 1e3 / res32["build_blas_ms"],
            "output_finite": res32["finite"], "output_nonfinite_px": res32["nonfinite_px"],
            "stage_ms_per_frame": {k: v[0] for k, v in res32["calib"].items()},
            "stage_ms_note": "3 untimed calibration frames with every stage bracketed by HIP events; the timed region "
                             "instruments the dominant stage only (roofline.avg_launch_us)",
        }
        if bc32 is None:
            out["metric"] = f"Mray/s (frames/s alongside) of: {out['config']['workload']} -- NOT a BASELINE.json config"
        if args.rehearsal:
            out["rehearsal"] = f"backend={args.backend} single_device={args.single_device}: a rehearsal of the N > 1 code path, not a measurement"
            out["value_rehearsal"] = out["value"]
            out["value"] = None
        if res16 is not None:
            out["fp16"] = {"value": res16["mrays_per_s"], "unit": "Mray/s", "ms_per_step": res16["ms_per_step"],
                           "frames_per_s": res16["frames_per_s"], "rays_per_frame": res16["frame_rays"], "dtype": "f16",
                           "sustained_ms_per_step": res16["sustained"]["ms_per_step"] if res16["sustained"] else None, "sustained": res16["sustained"],
                           "workload": workload_text(sc, W, H, spp, abi.F16, args.rebuild_blas, args.reuse_direct_light), "baseline_config": bc16,
                           "speedup_vs_f32": res16["mrays_per_s"] / res32["mrays_per_s"],
                           "output_finite": res16["finite"], "output_nonfinite_px": res16["nonfinite_px"], "output_note": HALF_OUTPUT_NOTE,
                           "stage_ms_per_frame": {k: v[0] for k, v in res16["calib"].items()},
                           "roofline": roofline_of(res16, abi.F16, W, spp, world, wkey) if res16["stage"] else None}
        out["roofline"] = roofline_of(res32, abi.F32, W, spp, world, wkey) if res32["stage"] else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, spp)
        else:
            out["cpu_baseline"] = None
        if res_c4 is not None:
            k4 = workload_key(sc_c4.name, 1920, 1080, 8, True)
            out["c4"] = {"workload": workload_text(sc_c4, 1920, 1080, 8, abi.F16, True, False), "baseline_config": "configs[3]",
                         "value": res_c4["mrays_per_s"], "unit": "Mray/s", "dtype": "f16", "ms_per_step": res_c4["ms_per_step"],
                         "frames_per_s": res_c4["frames_per_s"], "rays_per_frame": res_c4["frame_rays"], "triangles": sc_c4.n_tris,
                         "sustained_ms_per_step": res_c4["sustained"]["ms_per_step"] if res_c4["sustained"] else None, "sustained": res_c4["sustained"],
                         "build_blas_ms": res_c4["build_blas_ms"], "build_mtris_per_s": sc_c4.n_tris / 1e3 / res_c4["build_blas_ms"],
                         "output_finite": res_c4["finite"], "output_nonfinite_px": res_c4["nonfinite_px"], "output_note": HALF_OUTPUT_NOTE, "stage_ms_per_frame": {k: v[0] for k, v in res_c4["calib"].items()},
                         "roofline": roofline_of(res_c4, abi.F16, 1920, 8, world, k4) if res_c4["stage"] else None,
                         "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(sc_c4, 8, abi.F16, 480, 270, budget_s=10.0)}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
