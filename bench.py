#!/usr/bin/env python3
"""bench.py -- Msamples/s of the MI355X wavefront path tracer on BASELINE.json's headline config.

Workload (config.workload): configs[1] = materialtest 1280x720, Lambert + RoughConductor, Sobol,
adaptive sampling off, spp_step 16.  One "step" = one pass of the hot path (16 spp over the whole
image); the default K = 16 steps is the 256-spp render the metric's wall-clock is quoted on.

    python bench.py --gpus 1 --steps 16 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: 16x16 image tiles are dealt round-robin to ranks (weak scaling is NOT what happens
here: total work is fixed -> "scaling": "strong"); each rank renders its shard with no data-path
collective and one RCCL gather of the framebuffers to rank 0 closes the timed region.

Extra objects on the JSON line:
  roofline     the traversal kernel with the larger share of the step (k_extend_p: closest hit, persistent lanes; k_shadow_q:
               any hit, eight rays per lane), HBM bound.  achieved = algorithmic bytes per launch / average launch
               duration, both measured live after the timed region: bytes from the kernel's own visit counters (64 B per
               BVH node fetch, 48 B per triangle fetch, plus the queue traffic per ray / vertex), duration from HIP events
               recorded on the stream the kernel runs on.  `traffic` is NOT measured in this run: it is the HBM-side byte
               count of a committed rocprofv3 --pmc pass of the same command (`traffic_source` names it).
  roofline_kernels   the same object for both traversal kernels.
  cpu_baseline the CPU oracle (oracle/, a port -- test infrastructure) timed on this box's host
               cores on a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SCENE = os.path.join(REPO, "assets", "materialtest", "c2.json")
NODE_BYTES, TRI_BYTES, RAY_QUEUE_BYTES = 64, 48, 56  # DESIGN.md "algorithmic bytes"
SHADE_EVENT_BYTES = 574          # 96 B state in + 96 B out + 80 B shading record + 112 B NEE record + ~190 B env tables per surface event
RETIRE_ENTRY_BYTES, RETIRE_PATH_BYTES = 5, 2*128 + 16 + 128  # per list entry scanned; per ended path: 2 lines read, 16 B result, one 128-byte record written
HBM_PEAK_GBS = 8000.0


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota)/int(period))))
    except Exception:
        pass
    return n


def self_launch(n_ranks, argv=None, python=None):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child process on a free local
    port and pass its output through.  Returns the child's return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spp-step", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--save-image", default=None, help="write the gathered image as PFM (rank 0)")
    ap.add_argument("--scene", default=None, help="another scene file (measurement of the other configs; the default "
                    "and the only bench line the contract asks for is configs[1])")
    ap.add_argument("--workload", default=None, help="label for --scene")
    ap.add_argument("--pools", type=int, default=int(os.environ.get("TGPU_BENCH_POOLS", "1")),
                    help="contexts (path pools, each on its own stream) per GPU; the image tiles are dealt over all of them.  2 pools (the C++ "
                         "CLI's default) fill each other's kernel tails and host round trips: +5.7 %% Msamples/s on configs[1] (803 -> 846..851 on "
                         "one box, 861 on another; profiles/r03_ab_notes.txt).  The contract line stays at 1: with two pools the launches of "
                         "different pools overlap, and the average launch duration rocprofv3 reports for a kernel no longer describes the kernel "
                         "alone, which is what `roofline` prices.  (With --pools > 1 the roofline phases run one pool at a time.)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not imported torch or touched the GPU;
        # the ranks are CHILD processes (one per GPU, torch.distributed.run), rank 0's JSON line and the return code are relayed.
        raise SystemExit(self_launch(args.gpus))

    import torch  # device memory, streams and torch.distributed: plumbing only
    import numpy as np
    import tungsten_amd as ta
    from tungsten_amd.dist import gather_framebuffers

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(1, n_dev)  # == local_rank on a full node; rehearsals may have fewer GPUs than ranks
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # TGPU_BENCH_FORCE_DIST=1: a ONE-rank process group, so that a one-GPU box rehearses the RCCL calls of the N > 1 run (communicator set-up,
    # barrier, gather of the framebuffers, the MAX all-reduce of the time) -- tests/test_gpu_dist.py
    force_dist = world == 1 and os.environ.get("TGPU_BENCH_FORCE_DIST", "0") == "1"
    if force_dist and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl == RCCL on ROCm.  TGPU_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (the ranks
        # then share GPUs and the collectives run on host copies): it exercises launch, sharding, timing and gather, not RCCL.
        backend = os.environ.get("TGPU_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    scene = ta.HostScene(args.scene or SCENE)
    K = max(1, args.pools)
    shards = [ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, rank*K + j, world*K) for j in range(K)]
    ctxs = [ta.Context(dev_index) for _ in range(K)]
    for c, sh in zip(ctxs, shards):
        c.upload(scene, sh, int(os.environ.get("TGPU_BENCH_MAX_PATHS", "0")))  # 0 = the core's default pool (A/B knob)
    outs = [torch.zeros((sh.n_pixels, 3), dtype=torch.float32, device=device) for sh in shards]
    ctx, shard = ctxs[0], shards[0]
    spp_step = args.spp_step

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    done = 0
    for _ in range(args.warmup):
        for c in ctxs:
            c.render_pass(done, spp_step)
        done += spp_step
    for c in ctxs:
        c.wait()
    # warm the end-of-render path outside the timed region as well: the device-side resolve, torch's scatter into image order (its kernels
    # are loaded on first use: tens of ms on a cold box, measured as 690 vs 790 Msamples/s for the first process on a fresh machine) and,
    # with several ranks, the communicator
    for c, o in zip(ctxs, outs):
        c.resolve_to_device(o.data_ptr())
    for c in ctxs:
        c.wait()
    gather_framebuffers(outs, shards, world, rank, K, dist)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for c in ctxs:
            c.render_pass(done, spp_step)
        done += spp_step
    for c in ctxs:
        c.wait()
    for c, o in zip(ctxs, outs):
        c.resolve_to_device(o.data_ptr())
    for c in ctxs:
        c.wait()
    image = gather_framebuffers(outs, shards, world, rank, K, dist)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = scene.res_x*scene.res_y*spp_step*args.steps
    value = samples/elapsed/1e6
    result = {
        "metric": "Msamples/sec (whole node), materialtest 1280x720",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed/args.steps*1e3, 3),
        "wall_clock_s": round(elapsed, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic camera paths on the vendored materialtest scene (assets/materialtest/c2.json)",
        "config": {"workload": "materialtest 1280x720, Lambert+RoughConductor only, 1xMI355X wavefront (BASELINE configs[1])",
                   "spp": spp_step*args.steps, "spp_step": spp_step, "sampler": "sobol", "adaptive": False,
                   "max_bounces": scene.max_bounces, "parallelism": f"tile-shard x{world}" + (f", {K} pools/GPU" if K > 1 else "")},
    }

    if args.scene:
        result["metric"] = "Msamples/sec (whole node)"
        result["data"] = "synthetic camera paths on " + args.scene
        result["config"]["workload"] = args.workload or os.path.basename(args.scene)
        result["config"]["sampler"] = "sobol" if scene.use_sobol else "uniform"
        result["config"]["resolution"] = [scene.res_x, scene.res_y]

    if rank == 0 and args.save_image and image is not None:
        ta.save_pfm(args.save_image, image)

    # ---- roofline of the dominant kernel (rank 0's shard) -------------------------------------
    if not args.no_roofline and rank == 0:
        rsteps = max(1, min(args.steps, 8))
        def phase(flags, base):
            # one pool at a time: with several pools rendering at once their launches overlap and a HIP-event duration no longer
            # belongs to one kernel.  Each pool renders the same passes on its share of the tiles; the totals are per whole image.
            for cx in ctxs:
                cx.set_profiling(flags)
                cx.reset_stats()
            for cx in ctxs:
                for i in range(rsteps):
                    cx.render_pass(base + spp_step*i, spp_step)
                cx.wait()
            tot = {}
            for cx in ctxs:
                for k, v in cx.stats().items():
                    tot[k] = tot.get(k, 0) + v
            for cx in ctxs:
                cx.set_profiling(0)
            return tot
        c = phase(1, done)                       # visit counters (the counting build of the kernel: k_extend_p<true>)
        t = phase(2, done + spp_step*rsteps)     # HIP-event timing of the production kernels, each on its own stream
        # committed PMC pass of the same command on configs[1] (profiles/collect.sh + summarize.py): HBM-side bytes per launch.
        # NOT measured in this run -- the source is named next to the number; null for other scenes / when the file is absent.
        pmc, pmc_src = {}, None
        pmc_path = os.path.join(REPO, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and not args.scene:
            try:
                pmc = json.load(open(pmc_path))
                pmc_src = "committed profile " + str(pmc.get("_source", "profiles/pmc_latest.json")) + " (rocprofv3 --pmc, separate run)"
            except Exception:
                pmc = {}

        def roof(kernel, bytes_per_launch, ms, n_launch, extra):
            avg_ms = ms/max(1, n_launch)
            achieved = bytes_per_launch/(avg_ms*1e-3)/1e9 if avg_ms > 0 else 0.0
            traffic = None
            for name, v in pmc.items():
                if isinstance(v, dict) and name.startswith(kernel):
                    traffic = v.get("hbm_bytes_per_launch")
            d = {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(achieved/HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None,
                 "algorithmic_bytes_per_launch": round(bytes_per_launch, 1), "avg_launch_ms": round(avg_ms, 5)}
            if traffic:
                d["frac_of_peak_by_traffic"] = round(traffic/(avg_ms*1e-3)/1e9/HBM_PEAK_GBS, 4)
            d.update(extra)
            return d

        launches = max(1, c["extend_launches"])
        ext_bytes = (c["extend_node_visits"]*NODE_BYTES + c["extend_tri_tests"]*TRI_BYTES + c["extend_rays"]*RAY_QUEUE_BYTES)/launches
        # the closest-hit kernel of flat scenes: persistent lanes (k_extend_p) unless TGPU_EXTEND_P=0 selects the fixed four-rays-per-lane form;
        # scenes with instances / spheres / disks run k_extend<general>
        general_geometry = scene.desc.contents.n_instance_sets > 0 or scene.desc.contents.n_spheres > 0 or scene.desc.contents.n_disks > 0
        ext_kernel = "k_extend<general>" if general_geometry else ("k_extend_q" if os.environ.get("TGPU_EXTEND_P", "1") == "0" else "k_extend_p")
        ext = roof(ext_kernel, ext_bytes, t["extend_ms"], t["extend_launches"],
                   {"per_ray": {"node_visits": round(c["extend_node_visits"]/max(1, c["extend_rays"]), 3),
                                "tri_tests": round(c["extend_tri_tests"]/max(1, c["extend_rays"]), 3)}})
        # shadow kernel.  Queue form (k_shadow_q): per ray 36 B in (direction + far end, origin + near end, end cap), per vertex a
        # 4-byte list entry in and a 4-byte visibility word out.  k_shadow (general scenes): per vertex the 112-byte NEE record
        # and a 32-byte radiance read-modify-write.
        sl = max(1, c["shadow_launches"])
        queue_form = c.get("shadow_vertices", 0) > 0
        if queue_form:
            sh_queue = c["shadow_rays"]*36 + c["shadow_vertices"]*8
        else:
            sh_queue = (c["shadow_rays"]/2.0)*(112 + 32)
        sh_bytes = (c["shadow_node_visits"]*NODE_BYTES + c["shadow_tri_tests"]*TRI_BYTES + sh_queue)/sl
        sha = roof(("k_shadow_iq" if general_geometry else "k_shadow_q") if queue_form else "k_shadow", sh_bytes, t["shadow_ms"], t["shadow_launches"],
                   {"per_ray": {"node_visits": round(c["shadow_node_visits"]/max(1, c["shadow_rays"]), 3),
                                "tri_tests": round(c["shadow_tri_tests"]/max(1, c["shadow_rays"]), 3)}})
        # the state-traffic kernels: all k_shade<class> launches together, and the retire / regenerate step
        shade = roof("k_shade", c["surface_events"]*SHADE_EVENT_BYTES/max(1, c["shade_launches"]), t["shade_ms"], t["shade_launches"],
                     {"per_event_bytes": SHADE_EVENT_BYTES, "launches": "one per BSDF class and iteration"})
        retire = roof("k_retire_regenerate", (c["extend_rays"]*RETIRE_ENTRY_BYTES + c["samples"]*RETIRE_PATH_BYTES)/max(1, c["retire_launches"]),
                      t["retire_ms"], t["retire_launches"], {"per_entry_bytes": RETIRE_ENTRY_BYTES, "per_ended_path_bytes": RETIRE_PATH_BYTES})
        for entry, prefix in ((shade, "k_shade<"), (retire, "k_retire_regenerate")):  # traffic: the sum over the matching kernels of the PMC pass, per launch of ours
            rows = [v for name, v in pmc.items() if isinstance(v, dict) and name.startswith(prefix)]
            if rows:
                entry["traffic"] = sum(v.get("hbm_bytes_per_launch", 0.0)*v.get("launches", 0) for v in rows)/max(1, sum(v.get("launches", 0) for v in rows))
                entry["traffic_source"] = pmc_src
                entry["frac_of_peak_by_traffic"] = round(entry["traffic"]/(entry["avg_launch_ms"]*1e-3)/1e9/HBM_PEAK_GBS, 4) if entry["avg_launch_ms"] > 0 else None
        dominant = ext if t["extend_ms"] >= t["shadow_ms"] else sha
        result["roofline"] = dict(dominant)
        result["roofline"].update({
            "launches_per_step": round(launches/rsteps, 2), "pools": K, "measured_with": "one pool rendering at a time" if K > 1 else "the only pool",
            "per_sample": {"extend_rays": round(c["extend_rays"]/max(1, c["samples"]), 3),
                           "shadow_rays": round(c["shadow_rays"]/max(1, c["samples"]), 3),
                           "surface_events": round(c["surface_events"]/max(1, c["samples"]), 3)},
            "kernel_ms_per_step": {k: round(t[k]/rsteps, 3) for k in ("extend_ms", "shadow_ms", "shade_ms", "other_ms")},
            "l2_peak_gbs": 34500.0, "frac_of_l2_peak": round(dominant["achieved"]/34500.0, 4),
        })
        result["roofline_kernels"] = [ext, sha, shade, retire]
        # is the committed PMC pass still about these kernels?  It names the digest of the kernel sources it was taken with.
        if pmc:
            import hashlib
            digest = hashlib.sha256(b"".join(open(os.path.join(REPO, "tungsten_amd", "csrc", f), "rb").read() for f in ("kernels.h", "device.h"))).hexdigest()[:16]
            result["roofline"]["traffic_profile_current"] = pmc.get("_kernels_sha") == digest
        st0 = ctx.stats()
        result["upload"] = {"upload_ms": round(st0["upload_ms"], 1), "bvh_build_ms": round(st0["bvh_build_ms"], 1), "bvh_on_device": bool(st0["bvh_on_device"]),
                            "note": "tgpu_upload_scene of rank 0's first pool, outside the timed region"}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N = 1) ----------------------
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle.oracle_py import Oracle
        cores = host_cores()
        orc = Oracle(scene, rng=1, math=1, rcp=0)
        full = ta.Shard(scene.res_x, scene.res_y)
        tc = time.perf_counter()
        orc.render_shard(full, spp_step, spp_step=spp_step, threads=cores)
        dt = time.perf_counter() - tc
        n = scene.res_x*scene.res_y*spp_step
        result["cpu_baseline"] = {"value": round(n/dt/1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                  "sample": f"1 step ({spp_step} spp) of the same {scene.res_x}x{scene.res_y} workload = {n} samples in {dt:.1f} s"}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
