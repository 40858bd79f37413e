"""RCCL rehearsal on a one-GPU box: the N > 1 bench run's collective calls -- communicator set-up on the rank's
device, barrier, the padded gather of the framebuffers on device tensors, the MAX all-reduce of the elapsed time, and the adaptive
pass' record all-reduce -- issued through a ONE-rank nccl (= RCCL) process group.  The sharding itself (tile t -> rank t % N, bit-identical
reassembly) is covered by the world-2 / world-3 gloo tests in test_distributed_gloo.py; what those cannot see is whether RCCL accepts the
tensors exactly as bench.py and tungsten_amd/dist.py hand them over."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(tmp_path, name, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    img = str(tmp_path / (name + ".pfm"))
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
                        "--save-image", img], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    return line, img


def test_bench_through_a_one_rank_rccl_group(ta, tmp_path):
    """bench.py with TGPU_BENCH_FORCE_DIST=1 takes every torch.distributed call of the multi-GPU run (backend nccl) and must write the same
    image, bit for bit, as the plain one-GPU run that issues no collective."""
    plain, img_plain = _bench(tmp_path, "plain", {})
    rccl, img_rccl = _bench(tmp_path, "rccl", {"TGPU_BENCH_FORCE_DIST": "1"})
    assert plain["n_gpus"] == rccl["n_gpus"] == 1 and rccl["value"] > 0
    a, b = ta.load_pfm(img_plain), ta.load_pfm(img_rccl)
    assert a.shape == b.shape == (720, 1280, 3)
    assert np.array_equal(a, b)


_RECORDS = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tungsten_amd.dist import exchange_records, gather_framebuffer
from tungsten_amd._host import RECORD_DTYPE, Shard
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(7)
t = np.zeros(57600, RECORD_DTYPE)
t["sample_count"] = rng.integers(0, 1 << 20, t.size)
t["mean"] = rng.random(t.size, np.float32)
t["running_variance"] = rng.random(t.size, np.float32)
out = exchange_records(t, dist, torch.device("cuda", 0))
assert out.tobytes() == t.tobytes(), "record all-reduce changed the table"
sh = Shard(80, 48)
rgb = torch.from_numpy(rng.random((sh.n_pixels, 3), np.float32)).cuda()
img = gather_framebuffer(rgb, sh, 1, 0, dist)
assert np.array_equal(img, sh.scatter(rgb.cpu().numpy())), "gather changed the image"
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_record_exchange_and_gather_on_device_tensors_through_rccl(tmp_path):
    """exchange_records (the adaptive pass' one exchange step) and gather_framebuffer with CUDA tensors on the nccl backend."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _RECORDS, REPO, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
