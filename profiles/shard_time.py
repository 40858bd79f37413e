#!/usr/bin/env python3
"""Wall time of ONE rank's shard of an N-rank render (tile t -> rank t % N) on one MI355X: the per-GPU time the N-GPU run
would see, without the gather.  Used for the projected scaling rows of profiles/rNN_ab_notes.txt.
    python profiles/shard_time.py [N ...]      (default 1 2 4 8; 256 spp in 16 passes of configs[1])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tungsten_amd as ta

scene = ta.HostScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "materialtest", "c2.json"))
ctx = ta.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    shard = ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, 0, n)
    ctx.upload(scene, shard)
    ctx.render(48, spp_step=16)          # warm-up
    best = 1e9
    for rep in range(3):
        ctx.set_profiling(0)
        ctx.reset_stats()
        t0 = time.perf_counter()
        for s in range(16):
            ctx.render_pass(48 + 256*rep + 16*s, 16)
        ctx.wait()
        best = min(best, time.perf_counter() - t0)
    st = ctx.stats()
    print(f"N={n}: shard of {shard.n_pixels} px, 256 spp: {best*1e3:.1f} ms  ({shard.n_pixels*256/best/1e6:.0f} Msamples/s per rank, "
          f"x{n} = {shard.n_pixels*256/best/1e6*n:.0f}; speed-up vs ideal {1.0:.2f})  extend launches {st['extend_launches']}, tail launches {st['tail_launches']}", flush=True)
ctx.close()
