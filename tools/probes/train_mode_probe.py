"""which side of the full-size training-mode comparison (tests/test_gpu_configs.py) carries the 3e-3 gradient difference?
GPU executor (BatchNorm sums in the conv epilogues) / the same with separate statistics launches / the per-layer path, each
against the CPU oracle, worst tensors printed."""
import copy, functools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from gapartnet_amd import _C, backend
from gapartnet_amd.network.backbone import SparseUNet
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud
from oracle import torch_ops

cuda = torch.device("cuda:0")
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scenes = make_batch(n_scenes, 20000, seed0=9100)
torch.manual_seed(11)
net = SparseUNet.build(6, [16, 32], 2, functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)).train()
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.3, 0.3)
VOX = (0.01, 0.01, 0.01)
with backend.using(torch_ops):
    cbatch = PointCloud.collate(scenes, voxel_size=VOX)
    cnet = copy.deepcopy(net)
    c_out = cnet(cbatch.voxel_tensor).features
    w = torch.linspace(-1.0, 1.0, c_out.shape[1])
    (c_out * w).sum().backward()
ref = {k: p.grad for k, p in cnet.named_parameters()}
gbatch0 = PointCloud.collate([pc.to(cuda) for pc in scenes], voxel_size=VOX)
L = _C.lib()
for label, fusion, native in (("executor, sums in epilogues", 1, True), ("executor, separate statistics", 0, True), ("per-layer path", 1, False)):
    L.gpn_net_bn_fusion(fusion)
    gnet = copy.deepcopy(net).to(cuda)
    gnet.use_native_executor = native
    for m in gnet.modules():
        if hasattr(m, "use_native_executor"):
            m.use_native_executor = native
    gbatch = PointCloud.collate([pc.to(cuda) for pc in scenes], voxel_size=VOX)
    g_out = gnet(gbatch.voxel_tensor).features
    (g_out * w.to(cuda)).sum().backward()
    errs = []
    for k, p in gnet.named_parameters():
        q = ref[k]
        if q is None or float(q.abs().max()) < 1e-9:
            continue
        errs.append((float((p.grad.cpu() - q).abs().max()) / float(q.abs().max()), k, float(q.abs().max())))
    errs.sort(reverse=True)
    print(f"== {label}: features |d| {float((g_out.detach().cpu() - c_out.detach()).abs().max()):.2e}")
    for e, k, s in errs[:6]:
        print(f"   {e:.2e}  {k}  (max|g| {s:.3e})")
L.gpn_net_bn_fusion(1)
