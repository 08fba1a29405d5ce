"""Furthest point sampling at pre-processing sizes (reference: convert_rendered_into_input.py:115, N ~ 1e5..1e6 -> 20 000
samples): one workgroup per cloud (gpn_pn2_furthest_point_sampling) vs workgroups sharing the cloud
(gpn_pn2_furthest_point_sampling_ws), microseconds per sample, identical indices checked."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gapartnet_amd import _C

dev = torch.device("cuda:0")
L = _C.lib()
vp = ctypes.c_void_p
stream = vp(torch.cuda.current_stream().cuda_stream)
for n, m in ((20000, 2048), (100000, 2048), (1000000, 1024)):
    xyz = torch.rand(1, n, 3, device=dev)
    out = {}
    for name in ("single", "shared"):
        temp = torch.full((1, n), 1e10, device=dev)
        idx = torch.zeros((1, m), dtype=torch.int32, device=dev)
        need = L.gpn_pn2_furthest_point_sampling_ws_bytes(1, n)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "single":
            rc = L.gpn_pn2_furthest_point_sampling(1, n, m, vp(xyz.data_ptr()), vp(temp.data_ptr()), vp(idx.data_ptr()), stream)
        else:
            rc = L.gpn_pn2_furthest_point_sampling_ws(1, n, m, vp(xyz.data_ptr()), vp(temp.data_ptr()), vp(idx.data_ptr()),
                                                      vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream)
        torch.cuda.synchronize()
        assert rc == 0, L.gpn_last_error()
        out[name] = ((time.perf_counter() - t0) / m * 1e6, idx.cpu())
    same = bool(torch.equal(out["single"][1], out["shared"][1]))
    print(f"N={n:8d} m={m}: one workgroup {out['single'][0]:8.2f} us/sample   shared {out['shared'][0]:8.2f} us/sample "
          f"(ws {need} B)  identical={same}")
