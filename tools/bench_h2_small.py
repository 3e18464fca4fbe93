import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from preworld_amd import ops, _lib
from bench_h2 import timeit
DEV='cuda:0'
for (B,D,H,W),cin,cout in [((1,8,100,100),64,64),((1,4,50,50),128,128)]:
    x=torch.randn(B,D,H,W,cin,device=DEV); w=torch.randn(cout,cin,3,3,3,device=DEV)*0.05
    xh=ops.f32_to_h2(x); wpk,inv=ops.pack_conv_weight_h2(w); y=torch.empty(B,D,H,W,cout,device=DEV)
    for algo in (0,2,3):
        for nt in ('1','2'):
            os.environ['PW_H2_NT']=nt
            t=timeit(lambda: ops.conv3d_h2(xh,wpk,inv,relu0=True,out0=y,algo=algo))
            print((B,D,H,W),cin,cout,'algo',algo,'NT',nt,_lib.lib().pw_last_kernel().decode(),'%.1f us'%t,flush=True)
