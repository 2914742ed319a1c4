import os, sys, numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
from cpulibs import CpuTracer
from raytracer_amd import scenes, capi
capi.init(0)
O=CpuTracer("oracle")
rng=np.random.default_rng(5)
sc=scenes.conference_scene(20000)
O.load(sc); G=capi.Scene(sc)
n=200000
lo = sc.pos.reshape(-1, 3).min(0) - 0.5; hi = sc.pos.reshape(-1, 3).max(0) + 0.5
o = rng.uniform(lo, hi, (n, 3)); d = rng.normal(size=(n, 3))
rays = np.zeros((n, 8), np.float32); rays[:, :3] = o; rays[:, 3:6] = d; rays[:, 6] = 1e-5; rays[:, 7] = np.inf
ci,ct=O.trace(rays)
nodes2,flat=G.bvh2()
for trav in (1,):
    gi,gt=G.trace(rays,traversal=trav)
    g2,t2=G.trace(rays,traversal=2)
    bad=np.nonzero((ci!=gi)|(ct!=gt))[0]
    print("trav",trav,len(bad), "trav2 bad", ((ci!=g2)|(ct!=t2)).sum())
    for b in bad[:10]:
        print(b, repr(rays[b]), "cpu", ci[b], ct[b:b+1].view(np.float32)[0], "gpu", gi[b], gt[b:b+1].view(np.float32)[0])
        for idx in (ci[b], gi[b]):
            if idx>=0: print("    tri", idx, sc.pos[flat[idx]])
