"""CCL on the ball-query graph of perfectly segmented scenes (the trained-network regime): per-call time, for rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gapartnet_amd import hip_ops as H
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud
dev = torch.device("cuda:0")
pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
sem = batch.sem_labels
fg = torch.nonzero(sem > 0).squeeze(1)
q = batch.points[fg, :3].contiguous()
bi = batch.batch_indices[fg].contiguous()
bo = torch.searchsorted(bi.long(), torch.arange(9, device=dev)).to(torch.int32)
lab = sem[fg].to(torch.int32)
Q = q.shape[0]
for K in (50, 300):
    idx, cnt = H.ball_query(q, q, bi, bo, 0.04, K, lab, lab)
    begin = torch.arange(Q, device=dev, dtype=torch.int32) * K
    be = torch.stack([begin, begin + cnt], 1).reshape(-1).contiguous()
    edges = idx.reshape(-1)
    for _ in range(3):
        labels = H.ccl(be, edges)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        labels = H.ccl(be, edges)
    b.record(); torch.cuda.synchronize()
    print(f"K={K}: Q={Q} E={int(cnt.sum())} components={int(torch.unique(labels).numel())}  {a.elapsed_time(b) * 100:.1f} us/call")
