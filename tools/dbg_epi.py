import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from preworld_amd import ops, _lib
DEV='cuda:0'
rs=np.random.RandomState(0)
for shape in [(1,32,4,8,8),(1,32,8,40,48)]:
    x=rs.standard_normal(shape).astype(np.float32); w=(rs.standard_normal((32,32,3,3,3))*0.05).astype(np.float32)
    res=rs.standard_normal((shape[0],32)+shape[2:]).astype(np.float32)
    T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    cl=lambda a: T(a.transpose(0,2,3,4,1))
    xh=ops.f32_to_h2(cl(x)); wpk,inv=ops.pack_conv_weight_h2(T(w))
    base=O.conv3d(x,w,None,1,1)
    for name,kw,want in [('epi1 h2 nores',dict(out_h2=(True,True)),base),('epi3 f32 nores',dict(out_h2=(False,False)),base),
                    ('epi2 h2 res',dict(out_h2=(True,True),residual=ops.f32_to_h2(cl(res))),base+res)]:
        y=ops.conv3d_h2(xh,wpk,inv,**kw)
        k=_lib.lib().pw_last_kernel().decode()
        yf=(ops.h2_to_f32(y) if isinstance(y,ops.H2) else y).permute(0,4,1,2,3).cpu().numpy()
        d=np.abs(yf-want)
        bad=np.argwhere(d>1e-3)
        print(shape,name,k,'max err %.3e'%d.max(),'n bad',len(bad), bad[:4].tolist())
        if len(bad) and shape[2]==4:
            for b_ in bad[:6]:
                i=tuple(b_)
                print('   idx',i,'got',yf[i],'want',want[i],'conv',base[i],'res',res[i])
            # which channels / positions are bad overall
            print('   bad channels', sorted(set(bad[:,1].tolist())), 'bad w', sorted(set(bad[:,4].tolist())), 'bad h', sorted(set(bad[:,3].tolist())), 'bad d', sorted(set(bad[:,2].tolist())))
