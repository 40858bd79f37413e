#!/usr/bin/env python3
"""Per-vertex timeline of the tail kernel's slowest path (profiling build of k_tail, TGPU_TAIL_TIMELINE=1): how long the end-of-render
drain takes and where.    python tests/tools/tail_timeline.py [ranks]   (default 8: one rank's shard of an 8-rank render of configs[1])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TGPU_TAIL_TIMELINE"] = "1"
import tungsten_amd as ta

repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = ta.HostScene(os.path.join(repo, "assets", "materialtest", "c2.json"))
shard = ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, 0, n)
ctx = ta.Context(0)
ctx.upload(scene, shard)
ctx.render(48, spp_step=16)
for prof in (0, 1):
    ctx.set_profiling(prof)
    t0 = time.perf_counter()
    for s in range(16):
        ctx.render_pass(48 + 256*prof + 16*s, 16)
    ctx.wait()
    print(f"profiling {prof}: {1e3*(time.perf_counter() - t0):.1f} ms", flush=True)
ctx.close()
