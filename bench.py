#!/usr/bin/env python3
"""bench.py -- headline benchmark: wavefront path tracing throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mrays/s (+ samples/s) at 1920x1080.  Workload at N=1: BASELINE configs[2], the 81,920-triangle displaced
icosphere over the reference's floor at 1920x1080, 256 spp -- the largest single-GPU configuration the metric is quoted at
(configs[1], the 34-triangle Cornell box, is rendered too and reported in `also`; --config picks any other).
One "step" = one complete render of that frame: every wavefront iteration (compact+shade -> generate+extend+logic) until both
stage-2 queues are empty, with the scene and all buffers already resident in HBM.  3 path pools of 750,000 slots are in flight
per GPU (--pools, --paths); the reference's own shape, one pool of NUM_PATHS = 1,000,000, is timed as well (`reference_pool_shape`).
N > 1: one process per GPU (torchrun), each GPU renders the full spp with its own RNG streams (slot_base = (rank * pools + pool)
* NUM_PATHS) and its own ticket range, i.e. the node accumulates N*spp ("weak": per-GPU work fixed); the per-GPU accumulation
buffers are combined by one RCCL sum-reduce to rank 0 inside the timed region of every step.

Prints ONE JSON line on rank 0.  `roofline` describes the extend kernel from an EXCLUSIVE pass (one pool, so its launches do not
overlap anything) that runs after the timed region; `cpu_baseline` is the oracle on the host cores (N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

# before torch starts the HIP runtime: the hosts' default at 4K is four path pools, which need more than HIP's four hardware queues
# (path-tracer_amd/__init__.py sets the same default, but bench.py initialises the GPU through torch before it imports the package)
# The GPU boxes export HIP's own default of 4; the bench raises it to 8 (never lowers a larger value) and says so in the line.
try:
    _hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
except ValueError:
    _hwq = 0
if _hwq < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
# N > 1: RCCL shares device memory between the ranks' processes; this pool's host driver only supports dmabuf IPC (the boxes export
# this already; kept here so that an environment built by a launcher cannot lose it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)

CONFIGS = {
    # name: (scene factory name, kwargs, width, height, spp, BASELINE.json configs index)
    "sphere": ("bunny_scale_scene", {}, 1920, 1080, 256, 2),    # the metric's configuration at N=1 (default)
    "cornell": ("cornell_box", {}, 1920, 1080, 64, 1),
    "floor": ("floor_scene", {}, 512, 512, 16, 0),
    "grid": ("grid_scene", {}, 3840, 2160, 32, 3),              # 32 spp per GPU
    "instanced": ("instanced_scene", {}, 3840, 2160, 32, 4),
    # configs[4] as BASELINE.json words it: mixed diffuse / specular / glossy materials and the 8-bounce cut (extensions, SURVEY 8f.1;
    # the scene description carries them in desc["settings"])
    "instanced_ext": ("instanced_scene_ext", {}, 3840, 2160, 32, 4),
}
# The 4K configs are also rendered with the tolerance-mode traversal north_star allows ("hit indices bit-exact, except for counted
# grazing/edge rays within 1e-6"): both modes in one line, the differing rays counted, the frames compared at equal spp.
BOTH_MODES = {"grid", "instanced", "instanced_ext"}


def build_scene(pta, dev, name):
    factory, kwargs, w, h, spp, idx = CONFIGS[name]
    desc = getattr(pta.scenes, factory)(**kwargs)
    t0 = time.time()
    scene = pta.scenes.populate(pta.init_scene(dev), desc)
    scene.build()
    return desc, scene, time.time() - t0


def timed_renders(pt, n, time_extend=False):
    """n complete renders back to back; returns (seconds, stats of the last one, summed rays, summed extend ms, timed launches)."""
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = launches = 0
    ext_ms = 0.0
    for _ in range(n):
        pt.reset()
        rs = pt.run(time_extend=time_extend)
        rays += rs.total_rays; ext_ms += rs.extend_ms; launches += rs.extend_launches
    torch.cuda.synchronize()
    return time.perf_counter() - t0, rs, rays, ext_ms, launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fast", action="store_true", help="fast mode: 4-wide BVH traversal (pt_set_traversal_mode; outside the bit-exact contract, reported separately)")
    ap.add_argument("--check-period", type=int, default=-1, help="tolerance mode: one extend workgroup in N re-walks its rays exactly (library default 64; 1 = all, 0 = none)")
    ap.add_argument("--config", default="sphere", choices=sorted(CONFIGS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--paths", type=int, default=0,
                    help="path slots per pool (the reference has one pool of 1,000,000, path-tracer.js:4).  0 = host.default_pool_shape: 3 pools of a "
                         "quarter of the frame's pixels each, between 750,000 (1080p) and 2,000,000 (4K) -- the measured optima "
                         "(profiles/r01_pool_size_sweep_v3.json, profiles/r03_pool_shape_and_weight_sweeps.txt)")
    ap.add_argument("--pools", type=int, default=0,
                    help="independent path pools per GPU, --paths slots each, driven together on their own streams (pt_render_run_many)")
    ap.add_argument("--roofline-paths", type=int, default=2_000_000, help="slots of the single pool of the exclusive roofline pass")
    ap.add_argument("--no-extras", action="store_true", help="skip the exclusive roofline pass, the reference-shaped pool and the configs[1] render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--strong", action="store_true", help="N > 1: split the config's spp N ways (fixed total work, 'scaling': 'strong') instead of giving every GPU the full spp")
    ap.add_argument("--lib-comm", action="store_true", help="N > 1: sum the accumulation buffers with the library's own RCCL reduce (pt_comm_create_rank / pt_comm_reduce_image) "
                                                             "instead of torch.distributed's")
    ap.add_argument("--png", default="", help="write the tonemapped frame of the last step here (rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    # PT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then share GPUs and the
    # reduce goes through host memory); the real thing is nccl = RCCL, one rank per GPU.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pta = importlib.import_module("path-tracer_amd")
    _, _, w, h, spp, cfg_index = CONFIGS[args.config]
    w, h, spp = args.width or w, args.height or h, args.spp or spp
    d_pools, d_paths = pta.host.default_pool_shape(w, h)
    args.pools, args.paths = args.pools or d_pools, args.paths or d_paths

    stream = torch.cuda.current_stream().cuda_stream
    dev = pta.request_device(gpu, stream=stream)
    if args.fast:
        dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
        if args.check_period >= 0:
            dev.call("pt_set_traversal_check", args.check_period)
    desc, scene, build_s = build_scene(pta, dev, args.config)
    info = scene.info()

    # per-GPU share: full spp each, distinct ticket range and RNG streams
    image = torch.zeros(h * w * 4, dtype=torch.float32, device="cuda")
    if args.strong and (spp % world or (spp // world) % 8):
        raise SystemExit(f"--strong: {spp} spp do not split into {world} shares of whole pixel tickets (8 samples each, logic.js:6)")
    share = pta.multigpu.strong_share if args.strong else pta.multigpu.weak_share
    settings = share(rank, world, w, h, spp, args.paths, pools=args.pools)
    spp_total = spp if args.strong else spp * world
    settings["imageDev"] = image.data_ptr()
    settings.update(desc.get("settings", {}))       # a scene's extension materials and bounce cut
    pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": settings})

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib_comm = None
    if world > 1 and args.lib_comm:                            # the library's communicator: rank 0 makes the id, torch.distributed only carries it
        ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(pta.multigpu.LibComm.unique_id().raw), dtype=torch.uint8))
        dist.broadcast(ident, src=0)
        import ctypes
        lib_comm = pta.multigpu.LibComm([dev], id128=ctypes.create_string_buffer(bytes(ident.cpu().numpy().tobytes()), 128), rank=rank, world=world)
    reduce_s, render_s = [], []

    def one_step():
        pt.reset()
        t_a = time.perf_counter()
        rs = pt.run()                                          # returns when this GPU's render has finished
        t_b = time.perf_counter()
        if lib_comm is not None:
            lib_comm.reduce_image(0)                           # ncclReduce on the context's stream, waited for
        else:
            pta.multigpu.reduce_image(image, dst=0)            # RCCL over xGMI: W*H*4 floats (no-op at N=1)
        if world > 1:
            # The reduce runs on RCCL's own stream; the next render's reset zeroes this same buffer from the pools' streams.
            # Make the order explicit instead of leaning on NULL-stream semantics: the host waits for the reduce.
            torch.cuda.synchronize()
        render_s.append(t_b - t_a); reduce_s.append(time.perf_counter() - t_b)
        return rs

    for _ in range(args.warmup):
        one_step()
    del reduce_s[:], render_s[:]
    barrier()
    t0 = time.perf_counter()
    rays = samples = wasted = checked = mismatched = flipped = beyond = lost = 0
    max_rel = 0.0
    for _ in range(args.steps):
        rs = one_step()
        rays += rs.total_rays; samples += rs.total_samples; wasted += rs.wasted_rays
        checked += rs.checked_rays; mismatched += rs.mismatched_rays; flipped += rs.flipped_rays; beyond += rs.beyond_tolerance_rays; lost += rs.lost_hit_rays; max_rel = max(max_rel, rs.max_rel_dist_diff)
        steps_per_render = rs.steps
        if rs.stack_overflows:
            raise SystemExit(f"traversal guard hit on {rs.stack_overflows} rays")
    barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(rays), float(samples), float(wasted)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    all_rays, all_samples, all_wasted = float(tot[1]), float(tot[2]), float(tot[3])

    ranks_report = None
    if world > 1:
        # what makes an N-GPU record verifiable: the world size the communicator reports, and per rank the physical device, its own rays,
        # render time and reduce time (the reduce includes waiting for the slowest rank)
        import ctypes
        bus = ctypes.create_string_buffer(64)
        dev.call("pt_device_pci_bus_id", bus, 64)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": gpu, "pci_bus_id": bus.value.decode(), "device_name": torch.cuda.get_device_name(gpu),
                "rays": int(rays), "render_ms_mean": round(1e3 * sum(render_s) / max(len(render_s), 1), 3), "reduce_ms_mean": round(1e3 * sum(reduce_s) / max(len(reduce_s), 1), 3),
                "Mrays_per_s_of_its_own_renders": round(rays / max(sum(render_s), 1e-9) / 1e6, 3), "slot_base": settings["slotBase"],
                "tickets": [settings["ticketBegin"], settings["ticketEnd"]]}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_report = {"world_size_seen_by_the_communicator": dist.get_world_size(), "backend": dist.get_backend(),
                        "reduce": "pt_comm_reduce_image (library, RCCL)" if lib_comm is not None else f"torch.distributed.reduce ({dist.get_backend()}; nccl = RCCL)", "ranks": gathered,
                        "distinct_devices": len({g["pci_bus_id"] for g in gathered})}
    if rank == 0:
        img = image.cpu().numpy().reshape(h, w, 4)
        assert np.all(img[..., 3] == spp_total), f"every pixel must have received exactly {spp_total} samples (image .w = spp x world)"
        if args.png:
            disp = pta.init_display({"device": dev, "canvas": args.png, "image": {"buffer": image.data_ptr(), "width": w, "height": h}})
            disp.draw()
        mode = ", FAST MODE: 4-wide BVH traversal, not bit-exact" if args.fast else ""
        ext = desc.get("settings", {})
        rules = (f"extension materials ({', '.join(sorted({m['type'] for m in ext['materials']}))}), bounce cut maxNb = {ext.get('maxNb')}"
                 if ext.get("materials") else "reference bounce rules")
        out = {
            "metric": "Mrays/s", "value": round(all_rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{cfg_index}] {args.config}: {info.n_tris} tris / {info.n_objects} instances, {w}x{h}, {spp // world if args.strong else spp} spp per GPU, "
                                   f"{args.pools} pool(s) of {args.paths} path slots (reference: 1 pool of 1000000), {rules}{mode}",
                       "spp_total": spp_total, "parallelism": f"spp-split x{world} ({'strong: the spp are divided' if args.strong else 'weak: every GPU renders the full spp'}), "
                                                                 "RCCL sum-reduce of the accumulation buffer"},
            "samples_per_s": round(all_samples / elapsed, 1),
            "rays_per_render_per_gpu": rays // args.steps, "wavefront_iterations_per_render": int(steps_per_render),
            # slots that ticket exhaustion has just retired are still pushed to the camera queue and trace one last ray
            # (logic.js:177-190); they are part of `value`'s ray count, as they are of the reference's ray-trace queue
            "wasted_rays_per_render_per_gpu": wasted // args.steps, "useful_Mrays_per_s": round((all_rays - all_wasted) / elapsed / 1e6, 3),
            "scene_build_s": round(build_s, 3),
            "hip_hardware_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),     # GPU_MAX_HW_QUEUES as this process set it (see the top of the file)
        }
        if ranks_report:
            out["multi_gpu"] = ranks_report
            out["multi_gpu"]["image_check"] = f"every pixel of the reduced image holds exactly {spp_total} samples: asserted"
        if args.fast:
            out["tolerance_check"] = {"rays_rewalked_exactly": checked, "different_hit": mismatched, "hit_vs_miss": flipped, "beyond_1e-6": beyond, "of_those_reference_hit_lost": lost, "max_rel_hit_dist_diff": max_rel,
                                      "share_rewalked": round(checked / max(rays, 1), 5)}
        if not args.no_extras:
            out["roofline"] = roofline_block(pta, dev, scene, desc, args, w, h, spp, pt, elapsed / args.steps * 1e3)
            out["reference_pool_shape"] = reference_shape_pass(pta, dev, scene, desc, w, h, spp)
            if args.config != "cornell" and not args.fast:
                out["also"] = {"cornell": other_config_pass(pta, gpu, "cornell", args)}
                # the headline workload moved from configs[1] (round 1) to configs[2] (round 2 on): both figures side by side, with the earlier rounds' driver records
                out["configs1_cornell_Mrays_per_s"] = out["also"]["cornell"]["Mrays_per_s"]
                out["previous_rounds"] = {"BENCH_r01 (headline = configs[1] cornell)": 12904.9, "BENCH_r02 (headline = configs[2] sphere)": 6182.0,
                                          "note": "compare value with r02's and configs1_cornell_Mrays_per_s with r01's"}
        if world == 1 and args.config in BOTH_MODES and not args.no_extras:
            # last of the GPU passes, with the main tracer's pools gone: their streams would otherwise keep hardware queues the other mode's pools need
            pt.destroy(); pt = None
            out["other_mode"] = other_mode_pass(pta, gpu, args, w, h, spp, img, args.fast)
            out["tolerance_mode_full_check"] = full_check_pass(pta, gpu, args, w, h, spp)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pta, desc, w, h, args.cpu_spp, args.paths)
        print(json.dumps(out), flush=True)
    if lib_comm is not None:
        lib_comm.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if pt is not None:
        pt.destroy()             # pool contexts borrow dev's scene: they go first
    dev.destroy()


def single_pool(pta, dev, scene, desc, w, h, spp, paths):
    """One more context on dev's GPU that borrows the built scene and holds ONE pool of `paths` slots with its own image."""
    d2 = pta.host.Device(sharing=dev)
    pt = pta.init_path_tracer({"device": d2, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": dict({"samples": spp, "numPaths": paths}, **desc.get("settings", {}))})
    return d2, pt


def pmc_file(args, pools, paths):
    f = os.path.join(ROOT, "profiles", f"roofline_{args.config}_{pools}x{paths}{'_fast' if args.fast else ''}.json")
    return f, (json.load(open(f)) if os.path.exists(f) else None)


def exclusive_pass(pta, dev, scene, desc, args, w, h, spp):
    """The extend kernel by itself: ONE pool (nothing overlaps its launches), one render that counts the algorithmic bytes (counting
    variant, exact per-ray visit counts), then one render with HIP events on the launch stream around EVERY extend launch -- the
    contract's per-launch figure, which a kernel trace of the same command must reproduce (rocprof_avg_launch_us).  The counter-side
    ceilings come from the committed PMC passes of this pool shape (profiles/roofline_<config>_1x<paths>.json, tools/pmc_passes.sh +
    tools/pmc_roofline.py): counters cannot be read inside a timed run."""
    paths = args.roofline_paths
    d2, pt = single_pool(pta, dev, scene, desc, w, h, spp, paths)
    try:
        rs_c = pt.run(count_bytes=True)
        sec, rs, rays, ext_ms, timed = timed_renders(pt, 1, time_extend="all")
    finally:
        d2.destroy()
    launches = int(rs.steps)
    avg_s = ext_ms / 1e3 / max(timed, 1)
    alg_bytes = rs_c.traversal_bytes / max(launches, 1)
    alg_gbs = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    f, pmc = pmc_file(args, 1, paths)
    res = {"shape": f"1 pool of {paths} slots: launches never overlap; HIP events on the launch stream around every extend launch",
           "hbm_algorithmic": {"achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                               "what": "SURVEY 8d: algorithmic bytes per launch / average launch duration"},
           "avg_launch_us": round(avg_s * 1e6, 3), "launches_per_render": launches, "launches_timed": int(timed),
           "kernel_time_per_render_ms": round(avg_s * launches * 1e3, 3), "render_ms": round(sec * 1e3, 3),
           "Mrays_per_s": round(rays / sec / 1e6, 3), "algorithmic_bytes_per_launch": round(alg_bytes, 1)}
    if pmc:
        t_gbs = pmc["hbm_bytes_per_launch"] / (pmc["rocprof_avg_launch_us"] * 1e-6) / 1e9
        res["hbm_traffic"] = {"achieved": round(t_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(t_gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": pmc["hbm_bytes_per_launch"],
                              "what": "bytes that crossed the L2's memory side (TCC_EA0_RDREQ by size + WRITE_SIZE) / the profiled launch time"}
        if "valu_cycles_per_inst" in pmc and avg_s > 0:
            busy = pmc["valu_insts_per_launch"] * pmc["valu_cycles_per_inst"] / (1024 * avg_s * pmc["gpu_clock_ghz"] * 1e9)
            res["valu_issue_MODELLED"] = {"frac": round(busy, 4), "counter_active_share": round(pmc["valu_busy"], 4),
                                          "what": f"MODELLED, not measured in the run: {pmc['valu_insts_per_launch'] / 1e6:.1f} M VALU instructions per launch (SQ_INSTS_VALU, committed counter pass) x "
                                                  f"{pmc['valu_cycles_per_inst']} cycles, the mean issue cost of the traversal loop's instruction mix from a microbenchmark table (tools/valu_mix.py, "
                                                  "profiles/r01_ubench_issue_rates_gfx950.txt) / (1024 SIMDs x launch time x clock), idle tail included.  counter_active_share = "
                                                  "SQ_ACTIVE_INST_VALU * 4 / SIMD cycles is the raw counter; it overstates (it passes 1 when instructions are added, profiles/r02_valu_sensitivity_sphere.txt)"}
        res.update({"rocprof_avg_launch_us": pmc["rocprof_avg_launch_us"], "occupancy": pmc["occupancy"], "l1_hit": pmc["l1_hit"], "l2_hit": pmc["l2_hit"],
                    "wait_on_data_share_of_wave_time": pmc["wait_on_data_share_of_wave_time"],
                    "sources": [os.path.relpath(f, ROOT)] + [f"profiles/{s}" for s in pmc["sources"]]})
    else:
        res["sources"] = []
        res["note"] = f"no committed PMC passes for this pool shape ({os.path.relpath(f, ROOT)} absent): only the algorithmic figure is known"
    return res


def roofline_block(pta, dev, scene, desc, args, w, h, spp, pt, ms_per_step):
    """`roofline` for the SHAPE THAT IS TIMED: SURVEY 8d's algorithmic bytes against the HBM peak.  With several pools in flight the
    extend launches of different pools overlap, so a launch's own duration (events) includes the time it shares the chip: the figure
    that describes the chip is the render's -- all algorithmic bytes of the render / the render's wall time -- and that is `frac`.  The
    literal per-launch figure is printed for the timed shape (events around one launch in sixteen of a further render) and for an
    exclusive one-pool pass, and the cross-check kernel time <= wall time x overlap depth is printed, not implied."""
    pools = getattr(pt, "pools", [pt])
    pt.reset()
    rs_c = pt.run(count_bytes=True)                      # counting variant: exact visit counts; deterministic render => the timed renders' bytes
    n_launches = sum(int(r.steps) for r in getattr(pt, "last_pool_stats", [rs_c]))
    pt.reset()
    import torch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rs_t = pt.run(time_extend=True)                      # events around one extend launch in sixteen, per pool
    torch.cuda.synchronize(); wall_ev = time.perf_counter() - t0
    avg_ov_s = rs_t.extend_ms / 1e3 / max(rs_t.extend_launches, 1)
    wall_s = ms_per_step / 1e3
    alg = float(rs_c.traversal_bytes)
    gbs = alg / wall_s / 1e9
    f, pmc = pmc_file(args, len(pools), args.paths)
    kernel_s = avg_ov_s * n_launches
    res = {"kernel": "k_extend_queue", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
           "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
           "shape": f"the timed shape: {len(pools)} pool(s) of {args.paths} slots, launches of different pools overlap",
           "what": "achieved = SURVEY 8d algorithmic bytes of ALL extend launches of one render (44 + 64 n_inner + 48 n_leaf + 128 (n_enter + n_leave) per ray, exact "
                   "counts) / ms_per_step: what HBM would have to deliver if no cache held a node.  traffic = HBM bytes per launch from the committed counter passes of this shape",
           "algorithmic_bytes_per_render": alg, "extend_launches_per_render": n_launches, "algorithmic_bytes_per_launch": round(alg / max(n_launches, 1), 1),
           "per_launch_in_the_timed_shape": {"avg_launch_us": round(avg_ov_s * 1e6, 3), "launches_timed": int(rs_t.extend_launches),
                                             "hbm_algorithmic_frac": round(alg / max(n_launches, 1) / avg_ov_s / 1e9 / HBM_PEAK_GBS, 4) if avg_ov_s > 0 else None,
                                             "what": "the contract's literal formula; a launch shares the chip with the other pools' launches for most of its duration"},
           "kernel_time_per_render_ms": round(kernel_s * 1e3, 3), "ms_per_step": round(ms_per_step, 3), "overlap_depth": round(kernel_s / wall_s, 3),
           "cross_check": f"sum of extend launch durations {kernel_s * 1e3:.1f} ms = ms_per_step {ms_per_step:.1f} x overlap depth {kernel_s / wall_s:.2f} "
                          f"(<= {len(pools)} pools); the render with events took {wall_ev * 1e3:.1f} ms"}
    if pmc:
        t_gbs = pmc["hbm_bytes_per_launch"] * n_launches / wall_s / 1e9
        res["hbm_traffic"] = {"achieved": round(t_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(t_gbs / HBM_PEAK_GBS, 4),
                              "what": "counter bytes per launch (committed passes of this shape) x launches per render / ms_per_step"}
        res.update({"rocprof_avg_launch_us": pmc["rocprof_avg_launch_us"], "occupancy": pmc["occupancy"], "l1_hit": pmc["l1_hit"], "l2_hit": pmc["l2_hit"],
                    "sources": [os.path.relpath(f, ROOT)] + [f"profiles/{s}" for s in pmc["sources"]]})
    else:
        res["sources"] = []
    res["exclusive"] = exclusive_pass(pta, dev, scene, desc, args, w, h, spp)
    ex = res["exclusive"]
    physical = {k: ex[k]["frac"] for k in ("hbm_traffic", "valu_issue_MODELLED") if k in ex}
    if physical:
        top = max(physical, key=physical.get)
        res["closest_physical_ceiling"] = {"name": top, "frac": physical[top],
                                           "what": "of the ceilings other than the contract's algorithmic one, the one the exclusive launch is closest to; hbm_algorithmic prices every node "
                                                   "record as if it came from HBM and passes 1 whenever the caches serve the nodes"}
    return res


def reference_shape_pass(pta, dev, scene, desc, w, h, spp):
    """The same frame with the reference's own pool: one pool of NUM_PATHS = 1,000,000 slots (path-tracer.js:4)."""
    d2, pt = single_pool(pta, dev, scene, desc, w, h, spp, 1_000_000)
    try:
        pt.run()
        sec, rs, rays, _, _ = timed_renders(pt, 1)
    finally:
        d2.destroy()
    return {"pools": 1, "paths": 1_000_000, "Mrays_per_s": round(rays / sec / 1e6, 3), "ms_per_render": round(sec * 1e3, 3),
            "wavefront_iterations_per_render": int(rs.steps)}


def other_config_pass(pta, gpu, name, args):
    """BASELINE configs[1] (the Cornell box) with the same pools, on a context of its own: 1 warm-up + 3 timed renders."""
    _, _, w, h, spp, idx = CONFIGS[name]
    dev = pta.request_device(gpu)
    try:
        desc, scene, _ = build_scene(pta, dev, name)
        info = scene.info()
        pools, paths = pta.host.default_pool_shape(w, h)
        pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                                   "settings": {"samples": spp, "numPaths": paths, "pools": pools}})
        pt.run()
        sec, rs, rays, _, _ = timed_renders(pt, 3)
        pt.destroy()
    finally:
        dev.destroy()
    return {"workload": f"BASELINE configs[{idx}] {name}: {info.n_tris} tris / {info.n_objects} instances, {w}x{h}, {spp} spp, {pools} pool(s) of {paths} path slots",
            "Mrays_per_s": round(rays / sec / 1e6, 3), "ms_per_render": round(sec / 3 * 1e3, 3), "renders": 3}


def other_mode_pass(pta, gpu, args, w, h, spp, main_image, main_is_tolerance):
    """The same frame in the OTHER traversal mode (reference-exact <-> tolerance mode), on a context of its own with the same pools:
    1 warm-up + args.steps timed renders, then the two finished frames compared at equal spp.  The tolerance-mode render of the pair also
    runs once more with EVERY ray re-walked exactly (check period 1) for the full count of differing rays."""
    dev = pta.request_device(gpu)
    tol = not main_is_tolerance
    try:
        if tol:
            dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
            if args.check_period >= 0:
                dev.call("pt_set_traversal_check", args.check_period)
        desc, scene, _ = build_scene(pta, dev, args.config)
        settings = dict({"samples": spp, "numPaths": args.paths, "pools": args.pools}, **desc.get("settings", {}))
        pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": settings})
        pt.run()
        sec, rs, rays, _, _ = timed_renders(pt, args.steps)
        img = pt.read_image().astype(np.float64)
        res = {"mode": "tolerance (4-wide collapse, pt_set_traversal_mode PT_TRAVERSAL_WIDE)" if tol else "reference-exact",
               "Mrays_per_s": round(rays / sec / 1e6, 3), "ms_per_render": round(sec / args.steps * 1e3, 3), "renders": args.steps,
               "rays_per_render": int(rs.total_rays)}
        if tol:
            res["sampled_check_in_the_timed_renders"] = check_block(rs)
        pt.destroy()
    finally:
        dev.destroy()
    ref = main_image.astype(np.float64)
    a, b = img[..., :3] / np.maximum(img[..., 3:], 1.0), ref[..., :3] / np.maximum(ref[..., 3:], 1.0)
    rmse, mean = float(np.sqrt(((a - b) ** 2).mean())), float(b.mean())
    differing = int((np.abs(a - b).max(axis=-1) > 1e-5 * np.maximum(np.abs(b).max(axis=-1), 1e-3)).sum())
    res["frame_vs_the_other_mode_at_equal_spp"] = {"rmse": rmse, "rmse_over_mean_radiance": rmse / mean, "pixels_differing_by_more_than_1e-5_relative": differing,
                                                  "pixels": int(w * h), "sample_counts_equal": bool(np.array_equal(img[..., 3], ref[..., 3])),
                                                  "note": "a ray with another hit index forks its path; from then on its slot finishes at other steps, pixel tickets go to other slots "
                                                          "(other random streams), and the rest of the frame is another sample of the same estimator: the two frames agree like two "
                                                          "renders with different seeds.  At 8 spp (one ticket per pixel) only the forked pixels differ: tests/test_gpu_full_size.py"}
    return res


def full_check_pass(pta, gpu, args, w, h, spp):
    """One tolerance-mode render of the frame in which EVERY ray is walked a second time exactly (check period 1): the full count of
    rays whose closest hit differs from the reference walk's, and the largest relative hit-distance difference."""
    dev = pta.request_device(gpu)
    try:
        dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
        dev.call("pt_set_traversal_check", 1)
        desc, scene, _ = build_scene(pta, dev, args.config)
        settings = dict({"samples": spp, "numPaths": args.paths, "pools": args.pools}, **desc.get("settings", {}))
        pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": settings})
        rs = pt.run()
        pt.destroy()
    finally:
        dev.destroy()
    return check_block(rs)


def check_block(rs):
    return {"rays": int(rs.total_rays), "rays_rewalked_exactly": int(rs.checked_rays), "different_hit": int(rs.mismatched_rays), "hit_vs_miss": int(rs.flipped_rays),
            "hit_dist_apart_by_more_than_1e-6": int(rs.beyond_tolerance_rays), "of_those_a_reference_hit_was_lost": int(rs.lost_hit_rays),
            "max_rel_hit_dist_diff": float(rs.max_rel_dist_diff)}


def cpu_baseline(pta, desc, w, h, spp, paths):
    """The CPU oracle (a port: strict-IEEE C restatement of the reference kernels, OpenMP over path slots) timed on
    this box's host cores on a bounded sample of the same workload: same scene, camera, resolution and slot count,
    spp reduced.  The reference's own WGSL needs a WebGPU adapter, which this image does not have."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    # every host core this process may use: the affinity mask, and the thread cap the box sets for its one-GPU share (OMP_NUM_THREADS = 16
    # there; os.cpu_count() shows the whole machine, many times that, most of it other tenants')
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(usable, cap) if cap > 0 else usable
    s = orc.OrcScene(desc)
    r = orc.OrcRender(s, w, h, desc["camera"], spp, paths, ext=desc.get("settings"))
    t0 = time.perf_counter()
    steps, rays, _ = r.run(threads=threads)
    dt = time.perf_counter() - t0
    webgpu = "UNAVAILABLE (no WebGPU adapter)"
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "cores_note": f"threads used = min(cores in this process's affinity mask: {usable}, OMP_NUM_THREADS: {cap or 'unset'}); os.cpu_count() = {os.cpu_count()}",
            "sample": f"same scene and camera at {w}x{h}, {spp} spp (bounded sample), {paths} slots: {rays} rays in {dt:.1f} s",
            "reference_wgsl_on_software_adapter": webgpu}


if __name__ == "__main__":
    main()
