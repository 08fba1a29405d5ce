import sys; sys.path.insert(0,'.')
import torch
from gapartnet_amd.optim import FusedAdam
cuda=torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
shapes = [(16, 27, 6), (16,), (48, 27, 48), (5000, 3), (1,), (112, 27, 112)]
base = [torch.randn(s, generator=g).to(cuda) for s in shapes]
a = [torch.nn.Parameter(t.clone()) for t in base]; b = [torch.nn.Parameter(t.clone()) for t in base]
oa = FusedAdam(a, lr=1e-3); ob = torch.optim.Adam(b, lr=1e-3, foreach=False, fused=False)
for step in range(4):
    for i,(p,q) in enumerate(zip(a,b)):
        if i==3 and step<2: p.grad=q.grad=None; continue
        gr=torch.randn(p.shape, generator=g).to(cuda); p.grad,q.grad=gr.clone(),gr.clone()
    oa.step(); ob.step()
    print(step, [float((p-q).abs().max()) for p,q in zip(a,b)])
sd = oa.state_dict()
print([float(oa.state[p]["step"]) for p in a], [float(ob.state[q]["step"]) for q in b])
print([float((oa.state[p]["exp_avg_sq"]-ob.state[q]["exp_avg_sq"]).abs().max()/ob.state[q]["exp_avg_sq"].abs().max()) for p,q in zip(a,b)])
