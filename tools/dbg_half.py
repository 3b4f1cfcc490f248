import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np, oracle_lib, vfhip
w,h=64,16
rng=np.random.default_rng(0)
raw=rng.integers(0,256,vfhip.plane_layout("NV12",w,h)[1],dtype=np.uint8)
cs=vfhip.ConvertScale(0); cs.configure("NV12",w,h,"BGRA",w//2,h//2,colorimetry="bt2020",chroma_site="mpeg2")
got=cs.process(raw).reshape(h//2,w//2,4).astype(int)
want=oracle_lib.load().convertscale("NV12",w,h,raw,"bt2020","mpeg2","bilinear","BGRA",w//2,h//2).astype(int)
d=got-want
print("kernel",cs.kernel_name,"nbad",(d!=0).sum(),"of",d.size,"max",np.abs(d).max())
for c in range(4): print("chan",c,"bad",(d[...,c]!=0).sum(), "maxabs", np.abs(d[...,c]).max())
ys,xs,cc=np.nonzero(d)
print("cols mod 4 hist", np.bincount(xs%4, minlength=4), "rows", sorted(set(ys))[:10])
print(got[0,:6]); print(want[0,:6])
