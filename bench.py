#!/usr/bin/env python3
"""bench.py -- headline benchmark: wavefront path tracing throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mrays/s at 1920x1080.  Workload at N=1: configs[1], the Cornell box at 1920x1080,
64 spp, 3 path pools of 750,000 slots in flight (--pools, --paths; the reference has one pool of NUM_PATHS = 1,000,000).
One "step" = one complete render of that frame: every wavefront iteration (compact+generate+shade -> extend+logic) until both
stage-2 queues are empty, with the scene and all buffers already resident in HBM.  N > 1: one process per GPU (torchrun),
each GPU renders the full 64 spp with its own RNG streams (slot_base = (rank * pools + pool) * NUM_PATHS) and its own ticket
range, i.e. the node accumulates N*64 spp ("weak": per-GPU work fixed); the per-GPU accumulation buffers are
combined by one RCCL sum-reduce to rank 0 inside the timed region of every step.

Prints ONE JSON line on rank 0 with `roofline` (extend kernel, HBM bound) and `cpu_baseline` (oracle, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)

CONFIGS = {
    # name: (scene factory name, kwargs, width, height, spp)
    "cornell": ("cornell_box", {}, 1920, 1080, 64),          # BASELINE configs[1]  (the metric's configuration)
    "floor": ("floor_scene", {}, 512, 512, 16),              # BASELINE configs[0]
    "sphere": ("bunny_scale_scene", {}, 1920, 1080, 256),    # BASELINE configs[2]
    "grid": ("grid_scene", {}, 3840, 2160, 32),              # BASELINE configs[3]: 32 spp per GPU
    "instanced": ("instanced_scene", {}, 3840, 2160, 32),    # BASELINE configs[4]
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fast", action="store_true", help="fast mode: 4-wide BVH traversal (pt_set_traversal_mode; outside the bit-exact contract, reported separately)")
    ap.add_argument("--no-extend-timing", action="store_true", help="leave the HIP events around the extend launches out (roofline fields become 0)")
    ap.add_argument("--config", default="cornell", choices=sorted(CONFIGS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--paths", type=int, default=750_000,
                    help="path slots per pool (the reference has one pool of 1,000,000, path-tracer.js:4); 3 pools of 750,000 is the measured optimum "
                         "on MI355X, one pool of 2,000,000 the best single pool (profiles/r01_pool_size_sweep_v3.json)")
    ap.add_argument("--pools", type=int, default=3,
                    help="independent path pools per GPU, --paths slots each, driven together on their own streams (pt_render_run_many)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=64, help="spp of the bounded CPU-baseline sample")
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
    factory, kwargs, w, h, spp = CONFIGS[args.config]
    w, h, spp = args.width or w, args.height or h, args.spp or spp
    desc = getattr(pta.scenes, factory)(**kwargs)

    stream = torch.cuda.current_stream().cuda_stream
    dev = pta.request_device(gpu, stream=stream)
    if args.fast:
        dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
    t0 = time.time()
    scene = pta.scenes.populate(pta.init_scene(dev), desc)
    scene.build()
    build_s = time.time() - t0
    info = scene.info()

    # per-GPU share: full spp each, distinct ticket range and RNG streams
    image = torch.zeros(h * w * 4, dtype=torch.float32, device="cuda")
    settings = pta.multigpu.weak_share(rank, world, w, h, spp, args.paths, pools=args.pools)
    settings["imageDev"] = image.data_ptr()
    pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": settings})

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(timed, count):
        pt.reset()
        rs = pt.run(time_extend=timed, count_bytes=count)
        pta.multigpu.reduce_image(image, dst=0)                # RCCL over xGMI: W*H*4 floats (no-op at N=1)
        return rs

    trav_bytes = 0
    for i in range(max(args.warmup, 1)):       # the first warm-up render also counts algorithmic bytes (counting kernel variant)
        rs = one_step(False, i == 0)
        if i == 0:
            trav_bytes = rs.traversal_bytes
    barrier()
    t0 = time.perf_counter()
    rays = samples = launches = 0
    extend_ms = 0.0
    for _ in range(args.steps):
        rs = one_step(not args.no_extend_timing, False)
        rays += rs.total_rays; samples += rs.total_samples; extend_ms += rs.extend_ms; launches += rs.extend_launches
        steps_per_render = rs.steps
        launches_per_render = sum(r.steps for r in pt.last_pool_stats) if args.pools > 1 else rs.steps   # extend launches of all pools
        if rs.stack_overflows:
            raise SystemExit(f"traversal guard hit on {rs.stack_overflows} rays")
    barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(rays), float(samples)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    all_rays, all_samples = float(tot[1]), float(tot[2])

    if rank == 0:
        img = image.cpu().numpy().reshape(h, w, 4)
        assert np.all(img[..., 3] == spp * world), "every pixel must have received exactly spp samples"
        if args.png:
            disp = pta.init_display({"device": dev, "canvas": args.png, "image": {"buffer": image.data_ptr(), "width": w, "height": h}})
            disp.draw()
        # one extend launch in sixteen is bracketed by HIP events (pt_render_run); mean over those, scaled to all launches
        all_launches = int(launches_per_render) * args.steps
        avg_s = extend_ms / 1e3 / max(launches, 1)
        ext_s = avg_s * all_launches
        bytes_per_launch = trav_bytes / max(int(launches_per_render), 1)
        achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "Mrays/s", "value": round(all_rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {info.n_tris} tris / {info.n_objects} instances, {w}x{h}, {spp} spp per GPU, "
                                   f"{args.pools} pool(s) of {args.paths} path slots (reference: 1 pool of 1000000), reference bounce rules" + (", FAST MODE: 4-wide BVH traversal, not bit-exact" if args.fast else ""), "spp_total": spp * world,
                       "parallelism": f"spp-split x{world}, RCCL sum-reduce of the accumulation buffer"},
            "samples_per_s": round(all_samples / elapsed, 1),
            "rays_per_render_per_gpu": rays // args.steps, "wavefront_iterations_per_render": int(steps_per_render),
            "scene_build_s": round(build_s, 3),
            "roofline": {"bound": "hbm", "kernel": "k_extend_queue", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(f"{args.config}_{args.pools}x{args.paths}"),
                         "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
                         "avg_launch_us": round(avg_s * 1e6, 3), "launches": all_launches, "launches_timed": launches,
                         "extend_share_of_step": round(ext_s / elapsed, 4),
                         "note": "achieved = algorithmic traversal bytes / measured launch time; the kernel also runs the next step's "
                                 "logic, whose bytes are not counted.  A value above the HBM peak means the scene's nodes are served "
                                 "from L2/LDS-side caches: `traffic` is what actually crossed the HBM interface.  With several pools the "
                                 "launches of different pools overlap on the GPU, so a launch's measured time includes the time it shared "
                                 "the CUs with others (extend_share_of_step can exceed 1); --pools 1 gives the kernel alone"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pta, desc, w, h, args.cpu_spp, args.paths)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dev.destroy()


def pmc_traffic(config):
    """HBM-side bytes (reads + writes) per extend launch from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic_<config>_<pools>x<paths>.json, written by tools/pmc_traffic.py from TCC_EA0_RDREQ{,_32B,_128B} and WRITE_SIZE); None if absent.
    PMC counters cannot be collected from inside the timed run, so this is the profiled run's figure."""
    f = os.path.join(ROOT, "profiles", f"traffic_{config}.json")
    if not os.path.exists(f):
        return None
    t = json.load(open(f))
    return t.get("hbm_bytes_per_launch", t.get("hbm_read_bytes_per_launch"))


def cpu_baseline(pta, desc, w, h, spp, paths):
    """The CPU oracle (a port: strict-IEEE C restatement of the reference kernels, OpenMP over path slots) timed on
    this box's host cores on a bounded sample of the same workload: same scene, camera, resolution and slot count,
    spp reduced.  The reference's own WGSL needs a WebGPU adapter, which this image does not have."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    threads = min(16, os.cpu_count() or 1)
    s = orc.OrcScene(desc)
    r = orc.OrcRender(s, w, h, desc["camera"], spp, paths)
    t0 = time.perf_counter()
    steps, rays, _ = r.run(threads=threads)
    dt = time.perf_counter() - t0
    webgpu = "UNAVAILABLE (no WebGPU adapter)"
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same scene and camera at {w}x{h}, {spp} spp (bounded sample), {paths} slots: {rays} rays in {dt:.1f} s",
            "reference_wgsl_on_software_adapter": webgpu}


if __name__ == "__main__":
    main()
