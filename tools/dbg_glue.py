import torch, sys
sys.path.insert(0, '.')
import rpnet_amd.functional as RF
from tests.helpers import rnd
DEV='cuda:0'
B,K,h,w,Cx,planes,soft = 2,2,16,24,256,2,False
H,W,Fc=4*h,4*w,64
y=(rnd(51,B,h,w,Fc)*2).to(DEV)
sc=(0.5+torch.rand(Fc,generator=torch.Generator().manual_seed(1))).to(DEV); sh=(rnd(52,Fc)*0.3).to(DEV)
proto=rnd(53,B,K,Fc).to(DEV); x=rnd(54,B,h,w,Cx).to(DEV); xs=torch.tensor([2.0**-13],device=DEV)
z1=torch.empty_like(y)
ex={"deferred":(y,sc,sh),"mask":True,"soft":soft,"x":x,"x_scale":xs,"planes":planes}
l,p=RF.CosineMatchUp.apply(z1,proto,H,W,20.0,ex)
m=ex["mask_out"]
for mode,name in ((1,"xk"),(2,"xq")):
    ref=RF.split_f16(x,xs,None,m,mode,want_scale=False,planes=planes)[0]
    got=ex[name]
    d=(ref.view(torch.int16)!=got.view(torch.int16))
    print(name,"mismatches",int(d.sum()),"of",d.numel())
    if d.any():
        idx=d.nonzero()[:8]
        for i in idx:
            i=tuple(i.tolist()); print(i, float(ref[i]), float(got[i]), float(x[i[1:]]), float(m[i[1:4]]))
