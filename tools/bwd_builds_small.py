import ctypes, os, sys
import numpy as np, torch
ROOT="/root/repo"
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
libs={}
for name,path in (("prev","tools/_explore/librroi_align_hip_prev.so"),("variant","tools/_explore/librroi_align_hip_variant.so"),("new","fots.pytorch_amd/rroi_align/_ext/rroi_align/librroi_align_hip.so")):
    l=ctypes.CDLL(os.path.join(ROOT,path))
    l.rroi_align_backward_hip.argtypes=[vp, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
    l.rroi_align_backward_workspace_bytes.restype=sz; l.rroi_align_backward_workspace_bytes.argtypes=[it]*7
    libs[name]=l
st=torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=30, iters=200):
    for _ in range(warm): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for (B,C,H,W,R,ph,pw) in ((2,64,120,160,32,11,96),(2,64,120,160,32,11,83),(2,64,120,160,8,11,96),(1,64,176,320,24,11,128)):
    rng=np.random.default_rng(1000+R+pw)
    h=rng.uniform(16,64,R)
    rois=torch.from_numpy(np.stack([rng.integers(0,B,R),rng.uniform(0,4*W,R),rng.uniform(0,4*H,R),h,h*rng.uniform(2,pw/float(ph),R),rng.uniform(-45,45,R)],1).astype(np.float32)).cuda()
    g=torch.randn((R,C,ph,pw),device="cuda"); gin=torch.empty((B,C,H,W),device="cuda")
    row=[]
    for rep in range(2):
        for name,l in libs.items():
            nb=l.rroi_align_backward_workspace_bytes(B,C,H,W,R,ph,pw); ws=torch.empty(nb,dtype=torch.uint8,device="cuda")
            def call(): assert l.rroi_align_backward_hip(g.data_ptr(),0.25,B,R,H,W,C,ph,pw,rois.data_ptr(),gin.data_ptr(),ws.data_ptr(),nb,0,st)==1
            row.append(f"{name} {timeit(call):5.1f}")
    print(f"C={C} R={R} {ph}x{pw}: "+"  ".join(row),flush=True)
