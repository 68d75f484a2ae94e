import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from coclr_amd import ops, engine
def rel(a,b): return float((a.double().cpu()-b.double().cpu()).abs().max()/(b.double().abs().max()+1e-30))
for (N,Cin,Cout,dims,k,s,p) in [(4,144,288,(4,4,4),(1,3,3),(1,1,1),(0,1,1)), (4,288,288,(4,4,4),(3,1,1),(1,1,1),(1,0,0)),
                                 (4,128,256,(4,4,4),(1,3,3),(1,1,1),(0,1,1)), (4,144,288,(4,8,8),(1,3,3),(1,1,1),(0,1,1)),
                                 (2,144,288,(4,4,4),(1,3,3),(1,1,1),(0,1,1)), (4,144,64,(4,4,4),(1,3,3),(1,1,1),(0,1,1)),(4,16,64,(4,4,4),(1,3,3),(1,1,1),(0,1,1))]:
    torch.manual_seed(0)
    x = torch.randn(N,Cin,*dims, requires_grad=True); w=(torch.randn(Cout,Cin,*k)*0.05).requires_grad_(True)
    ref = F.conv3d(x,w,None,s,p); dy=torch.randn_like(ref); ref.backward(dy)
    g = ops.ConvGeom(N,Cin,Cout,dims,k,s,p); run=engine.Run(torch.device('cuda'),False)
    y = torch.empty(N,Cout,*g.odim,device='cuda'); ops.conv_fwd(g,x.detach().cuda(),run.pack(w.detach().cuda(),False),y)
    dx = torch.empty(N,Cin,*dims,device='cuda'); ops.conv_fwd(g.dgrad(),dy.cuda(),run.pack(w.detach().cuda(),True),dx)
    dw = torch.empty(Cout,Cin,*k,device='cuda'); ws=torch.empty(g.wgrad_workspace(),device='cuda'); kk=k[0]*k[1]*k[2]
    ops.conv_wgrad(g,x.detach().cuda(),dy.cuda(),dw,ws,Cin*kk,kk,0)
    e = (dw.cpu()-w.grad).abs()
    bad = (e > 1e-3*w.grad.abs().max()).nonzero()
    print(N,Cin,Cout,dims,k,'fwd',rel(y,ref),'dgrad',rel(dx,x.grad),'wgrad',rel(dw,w.grad), 'nbad', len(bad), bad[:5].tolist(), bad[-3:].tolist())
