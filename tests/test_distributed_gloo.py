"""N > 1 path on CPU: world_size-2 (and 3) gloo processes each render their tile shard (the oracle
stands in for the GPU on this GPU-less box: the shard/gather plumbing under test is the same code
bench.py uses) and rank 0's gathered image must equal the single-rank image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO, scene_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import tungsten_amd as ta
    from tungsten_amd.dist import gather_framebuffer
    from oracle.oracle_py import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(72, 40))
    shard = ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, rank, world)
    rgb, _ = Oracle(scene, rng=1, math=1).render_shard(shard, 4, spp_step=4, threads=2)
    img = gather_framebuffer(torch.from_numpy(rgb), shard, world, rank, dist)
    if rank == 0:
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


class _OracleContext:
    """Stands in for tungsten_amd.Context on this GPU-less box: same pass-level contract, CPU checker inside."""

    def __init__(self, oracle, shard):
        import tungsten_amd as ta
        self.o, self.shard = oracle, shard
        self.rgb = np.zeros((shard.n_pixels, 3), np.float32)
        self.cnt = np.zeros(shard.n_pixels, np.uint32)
        self.rec = np.zeros(((shard.res_x + 3)//4)*((shard.res_y + 3)//4), ta.RECORD_DTYPE)

    def render_pass_adaptive(self, idx, cnt):
        self.o.render_shard_adaptive(self.shard, idx, cnt, self.rgb, self.cnt, self.rec, threads=2)

    def wait(self):
        pass

    def download_records(self):
        # like tgpu_download_records: only this shard's variance tiles are filled in
        out = np.zeros_like(self.rec)
        px = self.shard.pixel_index
        vt = (px % self.shard.res_x)//4 + ((px//self.shard.res_x)//4)*((self.shard.res_x + 3)//4)
        out[np.unique(vt)] = self.rec[np.unique(vt)]
        return out


def _adaptive_worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import tungsten_amd as ta
    from tungsten_amd.dist import gather_framebuffer, exchange_records
    from tungsten_amd.passes import render_passes
    from oracle.oracle_py import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(72, 40))
    shard = ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, rank, world)
    ctx = _OracleContext(Oracle(scene, rng=1, math=1), shard)
    render_passes(ctx, scene.res_x, scene.res_y, 40, 8, adaptive=True, exchange=lambda t: exchange_records(t, dist))
    img = gather_framebuffer(torch.from_numpy(ctx.rgb), shard, world, rank, dist)
    if rank == 0:
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


def test_adaptive_sampling_across_ranks_equals_reference_loop(ta, oracle_py, tmp_path):
    """Adaptive sampling is the one place the path has an exchange step (the SampleRecord table, all-reduced
    once per pass): 2 ranks + scheduler + exchange must reproduce the single-process reference loop."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "img.npy")
    mp.spawn(_adaptive_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    scene = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(72, 40))
    img, cnt, _ = oracle_py.Oracle(scene, rng=1, math=1).render_reference(40, spp_step=8, adaptive=True, threads=4)
    assert cnt.min() < 40 < cnt.max()
    assert np.array_equal(got, img)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_shard_gather_equals_single_rank(ta, oracle_py, tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    scene = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(72, 40))
    full = ta.Shard(scene.res_x, scene.res_y)
    rgb, _ = oracle_py.Oracle(scene, rng=1, math=1).render_shard(full, 4, spp_step=4, threads=2)
    assert np.array_equal(got, full.scatter(rgb))


def test_shard_sizes_are_balanced(ta):
    from tungsten_amd.dist import shard_sizes
    for world in (2, 4, 8):
        sizes = shard_sizes(1280, 720, world)
        assert sum(sizes) == 1280*720
        assert max(sizes) - min(sizes) <= 256  # at most one 16x16 tile


def test_bench_launches_its_own_ranks_from_a_plain_command_line():
    """`python bench.py --gpus 2` (no WORLD_SIZE): the parent starts torch.distributed.run itself, before importing torch, and
    relays the children's return code.  On this GPU-less box both ranks must get as far as bench.py's own "needs a GPU"
    exit -- i.e. they were started with RANK / WORLD_SIZE set -- and the launcher must report the failure."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TGPU_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():  # a GPU box: the two gloo ranks share the card and the line must come out
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        import json
        assert p.returncode == 0 and json.loads(line)["n_gpus"] == 2
    else:
        assert p.returncode != 0
        assert (p.stdout + p.stderr).count("bench.py needs a GPU") >= 2
