"""Interleaved A/B of core builds in ONE process (cdna guide rule 24)."""
import sys, time, glob, os, numpy as np
sys.path.insert(0,'.')
import torch
import tungsten_amd as ta
from tungsten_amd._core import load_core, Context
names = sys.argv[1:] or [os.path.basename(p)[4:-3] for p in sorted(glob.glob('build_tmp/ab/lib_*.so'))]
s=ta.HostScene(os.environ.get('AB_SCENE','assets/materialtest/c2.json'))
sh=ta.Shard(s.res_x,s.res_y)
ctxs={}
for n in names:
    c=Context(0, lib=load_core(os.path.abspath(f'build_tmp/ab/lib_{n}.so')))
    c.upload(s,sh); c.render(16,spp_step=16)
    ctxs[n]=c
imgs={n:c.download()[0] for n,c in ctxs.items()}
for n in names[1:]: print(n, 'image identical to', names[0], ':', bool(np.array_equal(imgs[names[0]].view(np.uint32), imgs[n].view(np.uint32))), flush=True)
steps=int(os.environ.get('AB_STEPS','8'))
BASE=int(os.environ.get('AB_BASE','16'))  # first sample index of the timed passes (>= 256: the second Sobol' block)
res={n:[] for n in names}
for rnd in range(int(os.environ.get('AB_ROUNDS','5'))):
    for n in names:
        c=ctxs[n]
        t=time.perf_counter()
        for i in range(steps): c.render_pass(BASE+16*i,16)
        c.wait(); dt=time.perf_counter()-t
        res[n].append(s.res_x*s.res_y*16*steps/dt/1e6)
for n in names:
    r=np.array(res[n]); print(f'{n:12s} median {np.median(r):7.1f}  min {r.min():7.1f} max {r.max():7.1f} Msamples/s', flush=True)
if os.environ.get('AB_PROF'):
    for n in names:
        c=ctxs[n]; c.set_profiling(2); c.reset_stats()
        for i in range(steps): c.render_pass(BASE+16*i,16)
        c.wait(); st=c.stats(); c.set_profiling(0)
        print(n, {k: round(st[k]/steps,2) for k in ('extend_ms','shadow_ms','shade_ms','other_ms')}, 'launches', st['extend_launches'], flush=True)
