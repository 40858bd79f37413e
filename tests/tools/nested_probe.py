import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import tungsten_amd as ta
from oracle import oracle_py
from conftest import scene_path
for name in sys.argv[1:]:
    hs = ta.HostScene(scene_path(f"next/{name}.json"), resolution=(128, 72))
    shard = ta.Shard(hs.res_x, hs.res_y)
    ctx = ta.Context(0)
    ctx.upload(hs, shard)
    ctx.render(8, spp_step=8)
    rgb, cnt = ctx.download()
    ctx.close()
    ref, rc = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 8, spp_step=8)
    print(name, "pixels differing:", int((rgb != ref).any(axis=1).sum()), "of", len(rgb), flush=True)
