import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from preworld_amd import ops, _lib
DEV='cuda:0'
rs=np.random.RandomState(0)
shape=(1,32,4,8,8)
x=rs.standard_normal(shape).astype(np.float32); w=(rs.standard_normal((32,32,3,3,3))*0.05).astype(np.float32)
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
cl=lambda a: T(a.transpose(0,2,3,4,1))
xh=ops.f32_to_h2(cl(x)); wpk,inv=ops.pack_conv_weight_h2(T(w))
base=O.conv3d(x,w,None,1,1)
for name,res in [('zeros',np.zeros((1,32,4,8,8),np.float32)),('ones',np.ones((1,32,4,8,8),np.float32)),
                 ('iota',np.arange(32*256,dtype=np.float32).reshape(1,4,8,8,32).transpose(0,4,1,2,3)*0.001)]:
    rh=ops.f32_to_h2(cl(res))
    y=ops.conv3d_h2(xh,wpk,inv,residual=rh,out_h2=(True,True))
    yf=ops.h2_to_f32(y).permute(0,4,1,2,3).cpu().numpy()
    d=yf-(base+res)
    bad=np.argwhere(np.abs(d)>1e-3)
    print(name,'n bad',len(bad))
    for b_ in bad[:8]:
        i=tuple(int(v) for v in b_)
        print('   ',i,'got-conv = %.5f'%(yf[i]-base[i]),'res',res[i])
