import sys, torch
sys.path.insert(0, '.')
from gapartnet_amd.optim import FusedAdam
dev = torch.device('cuda:0')
ps = [torch.nn.Parameter(torch.randn(100, device=dev)) for _ in range(4)]
opt = FusedAdam(ps, lr=1e-2)
keep = []
for it in range(6):
    opt.zero_grad(set_to_none=True)
    for i, p in enumerate(ps):
        g = torch.randn(100, device=dev)
        if it % 2: keep.append(torch.empty(100, device=dev))  # shift allocator state: addresses move
        p.grad = g
    if it == 3:
        ps[1].grad = None
    before = ps[1].detach().clone()
    opt.step()
    torch.cuda.synchronize()
    print(it, 'own', len(opt._own_grad), 'p1 moved', not torch.equal(before, ps[1].detach()), 'nstep', [opt._nstep.get(p) for p in ps])
# against torch.optim.Adam
torch.manual_seed(0)
a = [torch.nn.Parameter(torch.randn(100, device=dev)) for _ in range(4)]
b = [torch.nn.Parameter(p.detach().clone()) for p in a]
oa, ob = FusedAdam(a, lr=1e-2), torch.optim.Adam(b, lr=1e-2)
for it in range(8):
    gs = [torch.randn(100, device=dev) for _ in a]
    for opt, ps in ((oa, a), (ob, b)):
        opt.zero_grad(set_to_none=True)
        for i, p in enumerate(ps):
            p.grad = gs[i].clone() if not (it in (3, 6) and i in (1, 2)) else None
        opt.step()
print('max diff vs torch Adam', max(float((x - y).abs().max()) for x, y in zip(a, b)))
