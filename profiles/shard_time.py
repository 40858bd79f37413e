#!/usr/bin/env python3
"""Wall time of ONE rank's shard of an N-rank render (tile t -> rank t % N) on one MI355X: the per-GPU time the N-GPU run
would see, without the gather.  Used for the projected scaling rows of profiles/rNN_ab_notes.txt.
    python profiles/shard_time.py [N ...]      (default 1 2 4 8; 256 spp in 16 passes of configs[1]; TGPU_BENCH_POOLS=K: K path pools per rank,
                                                the rank's tiles dealt over them as bench.py --pools does)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tungsten_amd as ta

K = int(os.environ.get("TGPU_BENCH_POOLS", "1"))
scene = ta.HostScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "materialtest", "c2.json"))
ctxs = [ta.Context(0) for _ in range(K)]
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    shards = [ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, j, n*K) for j in range(K)]
    for c, sh in zip(ctxs, shards):
        c.upload(scene, sh)
        c.render_pass(0, 16); c.render_pass(16, 16); c.render_pass(32, 16)   # warm-up
    for c in ctxs:
        c.wait()
    best = 1e9
    for rep in range(3):
        for c in ctxs:
            c.set_profiling(0)
            c.reset_stats()
        t0 = time.perf_counter()
        for s in range(16):
            for c in ctxs:
                c.render_pass(48 + 256*rep + 16*s, 16)
        for c in ctxs:
            c.wait()
        best = min(best, time.perf_counter() - t0)
    px = sum(sh.n_pixels for sh in shards)
    st = ctxs[0].stats()
    print(f"N={n} pools={K}: shard of {px} px, 256 spp: {best*1e3:.1f} ms  ({px*256/best/1e6:.0f} Msamples/s per rank, "
          f"x{n} = {px*256/best/1e6*n:.0f})  extend launches {st['extend_launches']}, tail launches {st['tail_launches']}", flush=True)
for c in ctxs:
    c.close()
