"""Interleaved A/B of one build under different environment settings (read when a context is created), one process; also checks the images agree."""
import sys, time, os, numpy as np
sys.path.insert(0,'.')
import torch
import tungsten_amd as ta
from tungsten_amd._core import Context
variants = [v for v in sys.argv[1:]]   # e.g. "TGPU_EXTEND_P=1" "TGPU_EXTEND_P=2"
s=ta.HostScene(os.environ.get('AB_SCENE','assets/materialtest/c2.json'))
sh=ta.Shard(s.res_x,s.res_y)
ctxs={}
for v in variants:
    for kv in v.split(','):
        k,val=kv.split('='); os.environ[k]=val
    c=Context(0); c.upload(s,sh); c.render(16,spp_step=16)
    ctxs[v]=c
    for kv in v.split(','): os.environ.pop(kv.split('=')[0])
imgs={v:c.download()[0] for v,c in ctxs.items()}
ref=imgs[variants[0]]
for v in variants[1:]: print(v, 'image identical to', variants[0], ':', bool(np.array_equal(ref.view(np.uint32), imgs[v].view(np.uint32))), flush=True)
steps=int(os.environ.get('AB_STEPS','8'))
res={v:[] for v in variants}
for rnd in range(int(os.environ.get('AB_ROUNDS','6'))):
    for v in variants:
        c=ctxs[v]
        t=time.perf_counter()
        for i in range(steps): c.render_pass(16+16*i,16)
        c.wait(); dt=time.perf_counter()-t
        res[v].append(s.res_x*s.res_y*16*steps/dt/1e6)
for v in variants:
    r=np.array(res[v]); print(f'{v:28s} median {np.median(r):7.1f}  min {r.min():7.1f} max {r.max():7.1f} Msamples/s', flush=True)
for v in variants:
    c=ctxs[v]; c.set_profiling(2); c.reset_stats()
    for i in range(steps): c.render_pass(16+16*i,16)
    c.wait(); st=c.stats(); c.set_profiling(0)
    print(v, {k: round(st[k]/steps,2) for k in ('extend_ms','shadow_ms','shade_ms','other_ms')}, flush=True)
