#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s on Scene 1 (random-spheres cover),
1200x675, 500 spp nominal = 484 effective (camera.cuh:51-53), depth 20, one
MI355X per rank, image rows partitioned across ranks.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full frame through the hot path: mega_kernel over this rank's
row blocks with RNG states resident in HBM (they persist from frame to frame
exactly as in the reference's update() loop, mort.cu:93-120), followed, for
N > 1, by the RCCL gather of the packed uchar4 rows to rank 0 and the
de-interleave into the full framebuffer.  Seeding (setup_rng) and scene upload
are outside the timed region, as in the reference (mort.cu:691-725).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MODE_NAMES = {"mega": "megakernel", "wave": "wavefront kernels", "throughput": "megakernel over (pixel, stratum row) streams [non-parity]"}
SCENE_NAMES = {1: "random-spheres cover", 2: "two checker spheres", 3: "earth", 4: "two Perlin spheres", 5: "quads",
               6: "Cornell box, emissive light + MIS", 7: "Cornell smoke", 8: "book-2 final scene", 9: "book-2 final scene, low settings",
               10: "out-of-order spheres"}


def _oracle_run(scene, width, spp, depth, aspect, threads):
    from mort_amd import host
    from tests import oracle_lib as O
    world, cam = host.build_scene(scene, width=width, spp=spp, depth=depth, aspect=aspect)
    states = O.seed_states(69420, cam.image_width, cam.image_height)
    t0 = time.perf_counter()
    r = O.render(world, cam, states=states, nthreads=threads, want_accum=False, want_segments=False)
    dt = time.perf_counter() - t0
    eff = host.effective_spp(cam)
    samples = cam.image_width * cam.image_height * eff
    return {"msamples_per_s": samples / dt / 1e6, "seconds": dt, "segments": r["segments"], "samples": samples,
            "geometry": f"{cam.image_width}x{cam.image_height}", "eff_spp": eff, "depth": cam.bounce_limit, "threads": threads}


def cpu_baseline(args, threads):
    """Oracle (CPU restatement, oracle/) timed on this box's host cores: a bounded sample of the
    benchmarked workload (full geometry, reduced spp) on all threads and on one thread, and BASELINE config 1
    (Scene 1 200x112, 4 spp) in full on one thread."""
    # final_scene's oracle scans 3 400 primitives per ray (the reference has no BVH there): sample it on a smaller image
    cw = args.cpu_width or (args.width if args.scene not in (8, 9) else min(args.width, 160))
    a = _oracle_run(args.scene, cw, args.cpu_spp, args.depth, args.aspect, threads)
    one = _oracle_run(args.scene, cw if args.scene not in (8, 9) else 48, 1, args.depth, args.aspect, 1)
    c1 = _oracle_run(1, 200, 4, None, None, 1)
    return {
        "value": a["msamples_per_s"], "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"Scene {args.scene} {a['geometry']} at {args.cpu_spp} spp ({a['eff_spp']} effective), depth {a['depth']}: "
                  f"{a['samples']} samples, {a['segments']} segments in {a['seconds']:.2f} s on {threads} host threads "
                  f"(C oracle, oracle/mort_oracle.c)",
        "seconds": a["seconds"], "segments": a["segments"],
        "one_thread": {"value": one["msamples_per_s"], "unit": "Msamples/s", "cores": 1,
                       "sample": f"{one['geometry']} at 1 spp: {one['samples']} samples in {one['seconds']:.2f} s"},
        "config1_full": {"value": c1["msamples_per_s"], "unit": "Msamples/s", "cores": 1,
                         "sample": f"BASELINE config 1 in full: Scene 1 {c1['geometry']}, 4 spp, depth {c1['depth']}: "
                                   f"{c1['samples']} samples, {c1['segments']} segments in {c1['seconds']:.3f} s"},
    }


def host_loop(args, threads):
    """The product's own `--mode host` (mort_hip_render_host: the kernel body as a host loop, north_star's CPU figure) on the
    same bounded sample as cpu_baseline, all threads and one thread.  Reported beside cpu_baseline, never instead of it."""
    from mort_amd import host, hip
    cw = args.cpu_width or (args.width if args.scene not in (8, 9) else min(args.width, 160))
    world, cam = host.build_scene(args.scene, width=cw, spp=args.cpu_spp, depth=args.depth, aspect=args.aspect)
    tree = args.scene not in (1, 10)
    out = {}
    for name, t in (("all_threads", threads), ("one_thread", 1)):
        if t == 1:
            world, cam = host.build_scene(args.scene, width=cw if args.scene not in (8, 9) else 48, spp=1, depth=args.depth, aspect=args.aspect)
        r = hip.render_host(world, cam, nthreads=t, tree=tree, want_accum=False, want_segments=False)["stats"]
        out[name] = {"value": r["eff_samples"] / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": t, "seconds": r["seconds"],
                     "sample": f"{cam.image_width}x{cam.image_height} at {cam.samples_per_pixel} spp, {r['segments']} segments ({r['kernel_name']})"}
    return out


def _norm_kernel(name):
    """'void mega_bvh_kernel<768 false false false>(FastArgs)' and 'mega_bvh_kernel<768, false, false, false>' -> the same key."""
    return name.replace("void ", "").split("(")[0].replace(",", "").replace(" ", "")


def profile_figures(tag, kernel_name, valu_cpi):
    """Counter figures of the profiled configuration, computed from the rocprofv3 summaries committed under
    profiles/ (rocprofv3 cannot run inside this process): HBM traffic per launch, VALU issue fraction and VALU
    lane occupancy of the dominant kernel.  `kernel_name` is the launch's own name (template arguments included), so the
    one-sample cost probe and the non-parity launch of the same template never match.  `valu_cpi`: cycles one SIMD needs per
    wave64 VALU instruction, measured on this box by mort_hip_calib_valu (the guide's figure is 2)."""
    import csv
    out = {"traffic": None, "config": None}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm.json")))
        out["traffic"] = pm["traffic_bytes_per_launch"]
        out["config"] = pm["config"]
        out["traffic_source"] = f"profiles/{tag}_pmc_hbm.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
    except Exception:
        pass
    try:
        c = {}
        want = _norm_kernel(kernel_name)
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_pmc_sq_summary.csv"))):
            if _norm_kernel(r["kernel"]) == want or (kernel_name.startswith("wf_") and "wf_" in r["kernel"]):
                c[r["counter"]] = c.get(r["counter"], 0.0) + float(r["sum_over_dispatches"] if kernel_name.startswith("wf_") else (r.get("max_dispatch") or r["per_dispatch"]))
        # 1024 SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader engines.  Issue fraction = VALU instructions x (cycles a SIMD
        # needs per instruction) / SIMD-cycles of the launch.  No clamp: a value above 1 would mean the model is wrong.
        simd_cycles = 1024.0 * c["SQ_BUSY_CYCLES"] / 32.0
        out["valu_issue_frac"] = c["SQ_INSTS_VALU"] * valu_cpi / simd_cycles
        out["valu_cycles_per_inst_per_simd"] = valu_cpi
        out["valu_lane_occupancy"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0)
        out["valu_lane_throughput_frac"] = out["valu_issue_frac"] * out["valu_lane_occupancy"]
        out["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        if "SQ_INSTS_SALU" in c and "SQ_INSTS_BRANCH" in c:
            tot = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"] + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_FLAT", 0) + c.get("SQ_INSTS_SMEM", 0)
            out["scalar_and_branch_share_of_insts"] = (c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"]) / tot
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            out["wave_wait_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        out["valu_source"] = (f"profiles/{tag}_pmc_sq_summary.csv: issue = SQ_INSTS_VALU x {valu_cpi:.2f} cycles (measured, mort_hip_calib_valu) / "
                              "(1024 SIMDs x SQ_BUSY_CYCLES / 32); lane occupancy = SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64)")
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--scene", type=int, default=1, help="reference scene id 1..10 (default 1 = the headline workload)")
    ap.add_argument("--depth", type=int, default=None, help="bounce limit override")
    ap.add_argument("--aspect", type=float, default=None, help="aspect ratio override")
    ap.add_argument("--profile-tag", default="r3/headline", help="profiles/<tag>_pmc_*.{json,csv}: counter figures quoted when the run is the profiled configuration")
    ap.add_argument("--rows-per-block", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=4, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all online cores")
    ap.add_argument("--cpu-width", type=int, default=0, help="image width of the CPU-baseline sample (0 = the benchmarked width; 160 for scenes 8/9)")
    ap.add_argument("--mode", choices=["mega", "wave", "throughput"], default="mega",
                    help="mega: the headline megakernel; wave: the wavefront (HBM-streaming) form of the same path; throughput: the labelled "
                         "NON-PARITY mode (one stream per (pixel, stratum row) -- other random numbers than the reference's; never the headline)")
    ap.add_argument("--no-calib", action="store_true", help="skip the roofline calibration kernels (profiling runs: keeps the kernel list to the render's)")
    ap.add_argument("--no-throughput-line", action="store_true", help="skip the extra non-parity figure (profiling runs: keeps the kernel list to the headline's)")
    args = ap.parse_args()

    import torch
    from mort_amd import host, hip, partition

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the render path)")
    # MORT_BENCH_REHEARSAL=1: every rank on GPU 0 and the gather through gloo (host staging) -- a rehearsal of the N-rank control flow
    # on a one-GPU box, where RCCL refuses ranks that share a device; its timings mean nothing and the line says so
    rehearsal = os.environ.get("MORT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    # ---- scene, upload, seed (outside the timed region) ----
    world, cam = host.build_scene(args.scene, width=args.width, spp=args.spp, depth=args.depth, aspect=args.aspect)
    W, H = cam.image_width, cam.image_height
    eff = host.effective_spp(cam)
    ctx = hip.Context(local_rank)
    ctx.set_partition(rank, world_size, args.rows_per_block)
    ctx.upload_world(world)
    ctx.rng_seed(69420, W, H)
    lr = ctx.local_rows(H)
    fg = partition.FrameGather(H, W, 4, torch.uint8, rank, world_size, args.rows_per_block, dev)
    tile = fg.tile  # packed owned rows (padded so every rank's tile has the same shape)
    fg_host = partition.FrameGather(H, W, 4, torch.uint8, rank, world_size, args.rows_per_block, torch.device("cpu")) if rehearsal else None

    # a non-default torch stream: its handle is what the C ABI launches on, and torch.cuda.Event /
    # torch.distributed both follow torch's *current* stream, so everything below runs under it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ev_pairs = []
    mode = {"wave": hip.MODE_WAVE, "throughput": hip.MODE_THROUGHPUT}.get(args.mode, hip.MODE_MEGA)

    def step(record, mode=mode):
        e0 = e1 = None
        if record:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        # the kernel is launched on torch's current stream, so these events bracket it
        ctx.render_device(cam, tile.data_ptr(), 0, stream.cuda_stream, mode=mode, sync=False)
        if record:
            e1.record(stream)
            ev_pairs.append((e0, e1))
        if world_size > 1 and rehearsal:
            fg_host.tile.copy_(tile)  # blocking D2H on the current stream, then gloo
            fg_host.gather(dist)
        elif world_size > 1:
            fg.gather(dist)  # RCCL gather of the packed uchar4 rows + de-interleave on rank 0
        # N == 1: `tile` already is the full framebuffer (rank 0 owns every row)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in ev_pairs]
    # the labelled non-parity mode, timed the same way beside the headline (worlds of the two LDS state-machine kernels); never `value`
    elapsed_tp = None
    if args.mode == "mega" and args.scene in (1, 10, 8, 9) and not args.no_throughput_line:
        step(False, hip.MODE_THROUGHPUT)  # seeds the sub-streams
        step(False, hip.MODE_THROUGHPUT)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(False, hip.MODE_THROUGHPUT)
        sync_all()
        elapsed_tp = time.perf_counter() - t1
    t = torch.tensor([elapsed, elapsed_tp or 0.0], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t[0].item())
    elapsed_tp = float(t[1].item()) if elapsed_tp is not None else None

    # one more frame through the blocking entry to read the library's own counters/HIP-event time
    st = ctx.render_device(cam, tile.data_ptr(), 0, stream.cuda_stream, mode=mode, sync=True)

    rehearsal_ok = None
    if rehearsal and world_size > 1:  # the composed frame must be the single-rank frame
        step(False, mode)
        sync_all()
        if rank == 0:
            composed = fg_host.frame.clone()
            with hip.Context(0) as solo:
                solo.upload_world(world)
                solo.rng_seed(69420, W, H)
                for _ in range(args.warmup + args.steps + 2 + (0 if elapsed_tp is None else 0)):
                    ref = solo.render(cam, mode=mode, want_accum=False)["rgba"]
            rehearsal_ok = bool((composed.numpy() == ref).all())

    if rank == 0:
        samples_per_step = W * H * eff
        value = samples_per_step * args.steps / elapsed_max / 1e6
        avg_kernel_s = (sum(kernel_ms) / max(len(kernel_ms), 1)) * 1e-3
        pixels_launch = W * lr
        algo_bytes = 100 * pixels_launch  # 48 B state in + 48 B out + 4 B uchar4 per pixel; 0 B per segment
        if args.mode == "throughput":  # per (pixel, stratum row): 48 B state in + 48 B out + 12 B partial sum; the resolve kernel's 12 B in and 4 B per pixel out
            algo_bytes = pixels_launch * (cam.sqrt_spp * (96 + 12 + 12) + 4)
        if args.mode == "wave":
            algo_bytes = int(st["algorithmic_hbm_bytes"])  # + 240 B per segment of front / hit / pixel / stack records (wave_bvh.h)
        achieved = algo_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        kernel_name = ("wf_trav + wf_shade (per front)" if args.mode == "wave" else st["kernel_name"])
        # roofline calibration on THIS box, outside the timed region (include/mort_hip.h mort_hip_calib_*): the HBM rate a float4
        # copy reaches, and the cycles a SIMD needs per wave64 VALU instruction at 1, 3 and 8 resident waves
        calib = None
        if world_size == 1 and not args.no_calib:
            cv = {w: ctx.calib_valu(w, 0)["cycles_per_valu_per_simd"] for w in (1, 3, 4, 8)}
            calib = {"hbm_copy_GBs": ctx.calib_hbm_copy(1 << 30, 3), "valu_cycles_per_inst_per_simd": {str(w): v for w, v in cv.items()},
                     "note": "mort_hip_calib_valu (independent v_fma_f32, waves per SIMD -> cycles per instruction per SIMD) and "
                             "mort_hip_calib_hbm_copy (float4 copy of 1 GiB per buffer, read + write)"}
        valu_cpi = min(calib["valu_cycles_per_inst_per_simd"].values()) if calib else 2.0  # MI355X_MICROARCH.md: 2 cycles (SIMD-32)
        # counter figures from the PMC passes committed under profiles/, quoted only when this run is the profiled configuration
        pf = profile_figures(args.profile_tag, "wf_" if args.mode == "wave" else kernel_name, valu_cpi)
        cfg = pf.get("config") or {}
        same = (args.scene, W, H, args.spp, cam.bounce_limit, world_size, args.mode) == (
            cfg.get("scene"), cfg.get("width"), cfg.get("height", H if args.aspect is None else None), cfg.get("spp"),
            cfg.get("depth", cam.bounce_limit if args.depth is None else None), cfg.get("gpus"), cfg.get("mode", "mega"))
        if not same:
            pf = {"traffic": None}
        traffic = pf.get("traffic")
        out = {
            "metric": f"Msamples/sec (width x height x spp/s), Scene {args.scene} {W}x{H}" +
                      (" -- NON-PARITY throughput mode (own RNG streams; not the reference's image)" if args.mode == "throughput" else ""),
            "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Scene {args.scene} ({SCENE_NAMES.get(args.scene, '?')}) {W}x{H}, {args.spp} spp nominal = {eff} effective, "
                                   f"depth {cam.bounce_limit}, {MODE_NAMES[args.mode]}, seed 69420, host LCG scene",
                       "mode": args.mode, "partition": f"rows/{args.rows_per_block} interleaved over {world_size} rank(s)",
                       "nominal_msamples_per_s": W * H * args.spp * args.steps / elapsed_max / 1e6},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "peak_measured": calib["hbm_copy_GBs"] if calib else None,
                         "kernel": kernel_name,
                         "avg_kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "traffic_source": pf.get("traffic_source"),
                         "note": ("megakernel keeps scene, RNG state and bounce stack on chip: 0 B/segment by construction, "
                                  "so the HBM fraction is tiny; what binds it is instruction latency at the occupancy its registers allow -- "
                                  "valu_issue_frac x valu_lane_occupancy is the share of the VALU lane throughput it uses (DESIGN.md 4.7)") if args.mode == "mega" else
                                 ("NON-PARITY mode: one stream per (pixel, stratum row), so a pixel's sqrt_spp rows are independent work items; results are "
                                  "this mode's own (tests/test_gpu_throughput.py), not the reference's -- reported beside the headline, never as it") if args.mode == "throughput" else
                                 ("wavefront form: 100 B per pixel + 240 B per segment of front / hit / pixel / stack records; bound by "
                                  "front granularity (one segment of every live pixel per launch pair), not by HBM (DESIGN.md 4)"),
                         "valu_issue_frac": pf.get("valu_issue_frac"), "valu_lane_occupancy": pf.get("valu_lane_occupancy"),
                         "valu_lane_throughput_frac": pf.get("valu_lane_throughput_frac"),
                         "scalar_and_branch_share_of_insts": pf.get("scalar_and_branch_share_of_insts"), "wave_wait_share": pf.get("wave_wait_share"),
                         "valu_insts_per_launch": pf.get("valu_insts_per_launch"), "valu_source": pf.get("valu_source")},
            "calibration": calib,
            "kernel": {"segments_per_frame": st["segments"], "segments_per_s": st["segments"] / st["seconds"],
                       "hip_event_seconds": st["seconds"], "vgprs": st["kernel_vgprs"], "lds_bytes": st["kernel_lds_bytes"]},
        }
        if rehearsal:
            out["rehearsal"] = {"note": "MORT_BENCH_REHEARSAL=1: all ranks on GPU 0, gather through gloo -- control-flow rehearsal, timings meaningless",
                                "composed_frame_equals_single_rank": rehearsal_ok}
        if elapsed_tp:
            out["nonparity_throughput_mode"] = {
                "value": samples_per_step * args.steps / elapsed_tp / 1e6, "unit": "Msamples/s", "ms_per_step": elapsed_tp / args.steps * 1e3,
                "note": "MORT_MODE_THROUGHPUT: one XORWOW stream per (pixel, stratum row) instead of the reference's one per pixel -- same estimator and "
                        "per-sample arithmetic, other random numbers, so NOT the reference's image; an extra figure, not the headline (DESIGN.md 4.8)"}
        if world_size == 1 and args.cpu_spp > 0:
            threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)  # 16 = one GPU's CPU share
            out["cpu_baseline"] = cpu_baseline(args, threads)
            out["host_loop"] = host_loop(args, threads)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
