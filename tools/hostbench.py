import sys, time, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","."), "fots.pytorch_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","."))
from rroi_align._ext import rroi_align as ext
import bench
c=bench.CFG
f,r=bench.make_inputs(512)
dev=torch.device("cuda:0")
feats=torch.from_numpy(f).to(dev); rois=torch.from_numpy(r).to(dev)
out=torch.empty((512,256,8,64),device=dev)
nb=ext._lib.rroi_align_forward_workspace_bytes(1,256,160,160,512,0)
ws=torch.empty(nb,dtype=torch.uint8,device=dev)
st=torch.cuda.current_stream().cuda_stream
fn=ext._lib.rroi_align_forward_stages_hip
args=(feats.data_ptr(),0,0.25,1,512,160,160,256,8,64,rois.data_ptr(),out.data_ptr(),ws.data_ptr(),nb,2,3,st)
for _ in range(50): fn(*args)
torch.cuda.synchronize()
for K in (200,1000):
    t0=time.perf_counter()
    for _ in range(K): fn(*args)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(K,"enqueue us/step",(t1-t0)/K*1e6,"total us/step",(t2-t0)/K*1e6)
# pre-converted args
V=ctypes.c_void_p; I=ctypes.c_int; F=ctypes.c_float; S=ctypes.c_size_t
pargs=(V(feats.data_ptr()),I(0),F(0.25),I(1),I(512),I(160),I(160),I(256),I(8),I(64),V(rois.data_ptr()),V(out.data_ptr()),V(ws.data_ptr()),S(nb),I(2),I(3),V(st))
for K in (200,1000):
    t0=time.perf_counter()
    for _ in range(K): fn(*pargs)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(K,"preconv enqueue us/step",(t1-t0)/K*1e6,"total us/step",(t2-t0)/K*1e6)
