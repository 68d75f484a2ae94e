import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch
from oracle import coclr_oracle as orc
from test_gpu_engine import randomise, rel, _oracle_run
from backbone.s3dg import SepInception, STConv3d
torch.manual_seed(0)
def run(m, fn, x):
    m.train()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    probe = fn(orc.training_state({"m." + k: v for k, v in state.items()}, "m."), x)
    torch.manual_seed(11); dout = torch.randn_like(probe)
    sd32, x32, ref32 = _oracle_run(state, fn, x, dout, torch.float32, True)
    sd64, x64, ref64 = _oracle_run(state, fn, x, dout, torch.float64, True)
    m = m.cuda(); xg = x.cuda().requires_grad_(True)
    out = m(xg); out.backward(dout.cuda())
    print('fwd', rel(out, ref32), 'dx', rel(xg.grad, x64.grad), rel(x32.grad, x64.grad))
    for k,p in m.named_parameters():
        eg, er = rel(p.grad, sd64['m.'+k].grad), rel(sd32['m.'+k].grad, sd64['m.'+k].grad)
        if eg > 5e-4: print('  %-40s gpu %.2e cpu32 %.2e' % (k, eg, er))
    # per-channel forward error of output
    e = (out.detach().cpu()-ref32.detach()).abs().amax((0,2,3,4)) / (ref32.detach().abs().amax((0,2,3,4))+1e-9)
    print('  worst per-channel fwd rel err', float(e.max()), int(e.argmax()))
m = SepInception(832, [384,192,384,48,128,128]); randomise(m, 3)
run(m, lambda sd,xx: orc.sep_inception(sd,'m',xx,True,False), torch.relu(torch.randn(4,832,2,2,2)))
m = STConv3d(192, 384, 3, 1, 1); randomise(m, 4)
run(m, lambda sd,xx: orc.st_conv3d(sd,'m',xx,True,1,1), torch.relu(torch.randn(4,192,2,2,2)))
m = STConv3d(192, 384, 3, 1, 1); randomise(m, 4)
run(m, lambda sd,xx: orc.st_conv3d(sd,'m',xx,True,1,1), torch.relu(torch.randn(4,192,4,4,4)))
