"""Few ROIs at the training call's shapes: the forward by path (auto / direct / tiled), us per call through ext.forward
(output allocation included in every arm)."""
import sys, os, torch, numpy as np
ROOT='/root/repo'
sys.path[:0]=[ROOT, ROOT+'/fots.pytorch_amd', ROOT+'/tests']
import workloads as Wk
from rroi_align._ext import rroi_align as ext
def timed(fn, warm=50, n=300):
    for _ in range(warm): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (R,C,H,W,ph,pw,B) in ((8,64,120,160,11,83,2),(32,64,120,160,11,83,2),(32,64,120,160,11,100,2),(64,64,120,160,11,83,2),(128,64,120,160,11,83,2),(32,64,176,320,11,96,1),(32,256,160,160,8,64,1)):
    f,r=Wk.bench_inputs(R=R,C=C,H=H,W=W,img=4*W,seed=5,batch=B)
    F,Rr=torch.from_numpy(f).cuda(),torch.from_numpy(r).cuda()
    out=torch.empty((R,C,ph,pw),device='cuda')
    res={}
    for name,p in (('auto',ext.PATH_AUTO),('direct',ext.PATH_DIRECT),('tiled',ext.PATH_TILED)):
        res[name]=timed(lambda: ext.forward(F,Rr,ph,pw,0.25,path=p))
    print(R,C,H,W,ph,pw,B,' '.join(f'{k} {v:6.1f}' for k,v in res.items()), flush=True)
