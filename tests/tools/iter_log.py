import os, sys, time
sys.path.insert(0, "/root/repo")
import tungsten_amd as ta
scene = ta.HostScene("/root/repo/assets/materialtest/c2.json")
shard = ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 8)
ctx = ta.Context(0)
ctx.upload(scene, shard)
ctx.render(48, spp_step=16)
os.environ["TGPU_DEBUG"] = "1"
t0 = time.perf_counter()
for s in range(16):
    ctx.render_pass(48 + 16*s, 16)
ctx.wait()
print(f"total {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
ctx.close()
