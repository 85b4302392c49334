import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context, PLANES
from oracle import pyoracle, numpy_ref
W,H=1920,1080
cam, batch = synth.make_batch(W,H,1,seed=41)
prm=synth.Params()
ctx=Context(W,H,1); ctx.set_camera(cam); ctx.set_params(prm); ws=ctx.workspace(1)
dev=ctx.device
b=ctx.make_batch(torch.from_numpy(batch["disparity_now"]).to(dev),torch.from_numpy(batch["disparity_prev"]).to(dev),torch.from_numpy(batch["flow"]).to(dev),batch["t"],batch["q"],batch["dt"])
ctx.process(b,ws); ctx.synchronize()
lib=ctx.lib
lib.mod_debug_read.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_uint64]
N=W*H
mb=np.zeros(N,np.uint32); lib.mod_debug_read(ctx.h,0,mb.ctypes.data,mb.nbytes); mp=np.zeros(N,np.uint32); lib.mod_debug_read(ctx.h,4,mp.ctypes.data,mp.nbytes); mem=np.stack([mb,mp],1)
cl=np.zeros((16,8),np.int32); lib.mod_debug_read(ctx.h,1,cl.ctypes.data,cl.nbytes)
labels=ws["labels"][0].cpu().numpy()
vx,vy,vz=[ws["planes"][i,0].cpu().numpy() for i in (3,4,5)]
K=int(ws["n_clusters"][0])
print("K",K)
for k in range(K):
    comp,size,off,medpix,medbits,amb=cl[k][:6]
    seg=mem[off:off+size]
    pix=np.sort(seg[:,1]); exp=np.sort(np.nonzero(labels.ravel()==k)[0])
    ok=np.array_equal(pix,exp)
    nb=numpy_ref.norm3(vx.ravel()[seg[:,1]],vy.ravel()[seg[:,1]],vz.ravel()[seg[:,1]]).view(np.uint32)
    okb=np.array_equal(nb,seg[:,0])
    true=np.sort(seg[:,0])[::-1][size//2]
    print(k,"size",size,"cursor",cur[k],"pixels ok",ok,"bits ok",okb,"med_bits",np.uint32(medbits),"true",true, "amb",amb, "uniq pix", np.unique(seg[:,1]).size)
