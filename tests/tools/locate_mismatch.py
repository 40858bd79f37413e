"""Minimal repro of the full-size parity gap: renders ONE pixel through a one-pixel shard, one sample per
pass, on the GPU and with the oracle, and prints the first sample whose running mean differs.
Round 1 result: config 5 pixel (852,116) sample 26; config 4 pixel (53,184) sample 4 and pixel (446,314) sample 10.
Run on a GPU box from the repo root: python tests/tools/locate_mismatch.py"""
import os, sys, ctypes, subprocess
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle")
import numpy as np
import tungsten_amd as ta
from tungsten_amd import _abi
import oracle_py
os.makedirs("gpurun_out/c4", exist_ok=True)
subprocess.run([sys.executable, "assets/instances/make_c4.py", "gpurun_out/c4"], check=True, stdout=subprocess.DEVNULL)
for name, path, spp, pixels in [("C5", "assets/materialtest/c5.json", 32, [(852, 116)]), ("C4", "gpurun_out/c4/c4.json", 16, [(53, 184), (446, 314)])]:
    hs = ta.HostScene(path)
    full = ta.Shard(hs.res_x, hs.res_y)
    d = full.desc.contents
    seeds_all = np.ctypeslib.as_array(d.tile_seed, shape=(full.n_pixels,))
    for (x, y) in pixels:
        pid = y*hs.res_x + x
        i = int(np.nonzero(full.pixel_index == pid)[0][0])
        pix = np.ascontiguousarray(full.pixel_index[i:i+1]); seeds = np.ascontiguousarray(seeds_all[i:i+1])
        masked = _abi.ShardDesc(1, pix.ctypes.data_as(_abi.pu32), seeds.ctypes.data_as(_abi.pu32))
        class S:
            n_pixels = 1; desc = ctypes.pointer(masked); res_x = hs.res_x; res_y = hs.res_y; pixel_index = pix
        ctx = ta.Context(0)
        ctx.upload(hs, S, 4096)
        orc = oracle_py.Oracle(hs, **oracle_py.GPU)
        rgb_o = np.zeros((1, 3), np.float32); cnt_o = np.zeros(1, np.uint32)
        first = None
        for s in range(spp):
            ctx.render(1, spp_step=1, spp_begin=s)
            g, _ = ctx.download()
            orc.render_shard(S, 1, spp_step=1, rgb=rgb_o, cnt=cnt_o, spp_begin=s)
            if first is None and not np.array_equal(g, rgb_o):
                first = s
                print(name, (x, y), "first differing sample", s, "gpu mean", g[0].tolist(), "oracle mean", rgb_o[0].tolist(), flush=True)
        if first is None: print(name, (x, y), "no difference when rendered sample by sample (pass structure matters?)", flush=True)
        ctx.close()
