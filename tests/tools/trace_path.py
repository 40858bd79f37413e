"""Vertex-by-vertex dump of ONE sample on both sides: the oracle (ORACLE_TRACE) and the GPU (TGPU_TRACE_SLOTS, per-stage kernels,
no tail kernel), through a one-pixel shard.  Used to root-cause the round-1 full-size gaps.
Run on a GPU box from the repo root:  python tests/tools/trace_path.py [c5|c4a|c4b ...]   (stderr carries the dumps)"""
import os, sys, ctypes, subprocess
os.environ["ORACLE_TRACE"] = "1"
os.environ["TGPU_TRACE_SLOTS"] = "1"
os.environ.setdefault("TGPU_TAIL_MAX", "0")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle")
import numpy as np
import tungsten_amd as ta
from tungsten_amd import _abi
import oracle_py

CASES = {"c5": ("assets/materialtest/c5.json", (852, 116), 26),
         "c4a": ("gpurun_out/c4/c4.json", (53, 184), 4),
         "c4b": ("gpurun_out/c4/c4.json", (446, 314), 10)}
want = sys.argv[1:] or list(CASES)
if any(w.startswith("c4") for w in want):
    os.makedirs("gpurun_out/c4", exist_ok=True)
    subprocess.run([sys.executable, "assets/instances/make_c4.py", "gpurun_out/c4"], check=True, stdout=subprocess.DEVNULL)
scenes = {}
for name in want:
    path, (x, y), s = CASES[name]
    hs = scenes.get(path) or scenes.setdefault(path, ta.HostScene(path))
    full = ta.Shard(hs.res_x, hs.res_y)
    seeds_all = np.ctypeslib.as_array(full.desc.contents.tile_seed, shape=(full.n_pixels,))
    i = int(np.nonzero(full.pixel_index == y*hs.res_x + x)[0][0])
    pix = np.ascontiguousarray(full.pixel_index[i:i+1]); seeds = np.ascontiguousarray(seeds_all[i:i+1])
    masked = _abi.ShardDesc(1, pix.ctypes.data_as(_abi.pu32), seeds.ctypes.data_as(_abi.pu32))
    class S:
        n_pixels = 1; desc = ctypes.pointer(masked); res_x = hs.res_x; res_y = hs.res_y; pixel_index = pix
    sys.stderr.write(f"==== {name}: pixel {(x, y)} sample {s} ====\n"); sys.stderr.flush()
    g = np.zeros((1, 3), np.float32)
    if not os.environ.get("TRACE_ORACLE_ONLY"):
        ctx = ta.Context(0)
        ctx.upload(hs, S, 64)
        ctx.render(1, spp_step=1, spp_begin=s)
        g, _ = ctx.download()
        ctx.close()
    sys.stderr.write(f"---- oracle ----\n"); sys.stderr.flush()
    rgb_o = np.zeros((1, 3), np.float32); cnt_o = np.zeros(1, np.uint32)
    oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(S, 1, spp_step=1, rgb=rgb_o, cnt=cnt_o, spp_begin=s)
    sys.stderr.write(f"RESULT {name} gpu {g[0].view(np.uint32).tolist()} oracle {rgb_o[0].view(np.uint32).tolist()} equal={np.array_equal(g, rgb_o)}\n")
    sys.stderr.flush()
