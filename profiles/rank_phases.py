"""Phases of what one rank of a RANKS-rank bench run (default 8; STEPS passes, default 20) does between its two barriers: one process, one GPU, rank 0 s shard."""
import os, sys, time
sys.path.insert(0,'.')
import torch, numpy as np
import tungsten_amd as ta
from tungsten_amd.dist import gather_framebuffers
scene=ta.HostScene('assets/materialtest/c2.json')
RANKS=int(os.environ.get('RANKS','8'))
sh=ta.Shard(scene.res_x, scene.res_y, 0xBA5EBA11, 0, RANKS)
c=ta.Context(0); c.upload(scene, sh)
out=torch.zeros((sh.n_pixels,3),dtype=torch.float32,device='cuda')
done=0
for _ in range(3): c.render_pass(done,16); done+=16
c.wait(); c.resolve_to_device(out.data_ptr()); c.wait()
K=int(os.environ.get('STEPS','20'))
for rep in range(4):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(K): c.render_pass(done,16); done+=16
    t1=time.perf_counter(); c.wait(); t2=time.perf_counter()
    c.resolve_to_device(out.data_ptr()); c.wait(); t3=time.perf_counter()
    img=torch.zeros((scene.res_y*scene.res_x,3),dtype=torch.float32,device='cuda'); idx=torch.from_numpy(sh.pixel_index.astype(np.int64)).cuda(); img[idx]=out; torch.cuda.synchronize(); t4=time.perf_counter()
    st=c.stats()
    print(f'steps {K}: enqueue {1e3*(t1-t0):.2f}  render+wait {1e3*(t2-t1):.2f}  resolve {1e3*(t3-t2):.2f}  assemble {1e3*(t4-t3):.2f} ms; total {1e3*(t4-t0):.2f}  -> x{RANKS} = {RANKS*sh.n_pixels*16*K/(t4-t0)/1e6:.0f} Msamples/s; launches ext {st["extend_launches"]} tail {st["tail_launches"]}', flush=True)
    c.reset_stats()
