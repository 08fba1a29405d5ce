"""NPCS -> camera-frame similarity fit for ALL proposals of a batch at once.  On the GPU the work is
the HIP entry point ``gpn_pose_fit`` (csrc/pose.hip: every 5-point hypothesis of every proposal in one launch, selection +
inlier fit + box in a second); the torch formulation below is the same procedure on CPU tensors (the host-side restatement the
CPU tests compare with the sequential function) and documents the algorithm.

``misc/pose_fitting.py`` restates the reference (gapartnet/misc/pose_fitting.py:4-147): per proposal, a Python loop of up
to 100 RANSAC iterations, each a 5-point Umeyama fit (3x3 SVD) and a residual pass over the proposal's points, on CPU
numpy.  Here the same procedure runs for every proposal and every hypothesis together:

* the 5-point fits of all P x H hypotheses are ONE batched 3x3 SVD;
* the residual of every hypothesis over every point of its proposal is a gather of the proposal's transform per point and a
  segmented sum (hypotheses in chunks to bound memory);
* the reference's sequential choice - keep the hypothesis with the smallest total residual, stop at the first iteration
  whose running minimum is below ``stop_thrsh`` - is evaluated after the fact: running minimum along H, first index below
  the threshold, first arg-min over the iterations up to it (NaN residuals never win, as ``nan < best`` is False there);
* the final fit on the inliers is a weighted, segmented Umeyama (segment sums of outer products, one more batched SVD).

Quirks kept: single-point proposals are duplicated, which makes every hypothesis NaN (zero variance) and therefore leaves
them without a pose; hypotheses with a residual of 1e10 or more never replace the initial "best"; the pass threshold is max(|src|/|tgt|, |tgt|/|src|) of the MEAN point
norms; the inlier ratio counts non-zero inlier INDICES (a proposal whose only inlier is its first point has ratio 0);
proposals with ratio < 0.01 have no pose.  Random draws: the reference consumes ``np.random.randint(n, size=5)`` per
iteration until it stops; ``draw_picks`` draws all H iterations of a proposal up front, so the stream of the global
generator differs once a proposal stops early - for identical picks the results agree to float64 round-off
(tests/test_pose_fitting_batched.py feeds the sequential function the same picks).  All arithmetic is float64 like numpy's.
"""
from typing import Dict, Optional

import numpy as np
import torch

_CORNER_SIGNS = [[-1, -1, -1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1], [1, 1, -1], [1, -1, 1], [-1, 1, 1], [1, 1, 1]]


def draw_picks(sizes, max_iters: int = 100) -> torch.Tensor:
    """[P, max_iters, 5] int64 sample indices, proposal by proposal from numpy's global generator (single-point proposals
    are fitted on their duplicated point: range 2)"""
    return torch.from_numpy(np.stack([np.random.randint(max(int(n), 2) if int(n) != 1 else 2, size=(max_iters, 5))
                                      for n in sizes]).astype(np.int64))


def _umeyama(src: torch.Tensor, dst: torch.Tensor, weight: Optional[torch.Tensor] = None):
    """batched similarity fit dst ~ src @ (s R) + t.  src, dst [B, n, 3] (weight [B, n] of 0/1 selects the points).
    -> scale [B], rotation [B,3,3], translation [B,3], transform [B,4,4] (pose_fitting.py:4-43)"""
    if weight is None:
        weight = torch.ones(src.shape[:2], dtype=src.dtype, device=src.device)
    n = weight.sum(1)                                                       # [B]
    w = weight[:, :, None]
    mu_s, mu_d = (src * w).sum(1) / n[:, None], (dst * w).sum(1) / n[:, None]
    cs, cd = (src - mu_s[:, None]) * w, (dst - mu_d[:, None]) * w
    cov = torch.einsum("bni,bnj->bij", cd, cs) / n[:, None, None]           # (dst - mu_d)(src - mu_s)^T / n
    finite = torch.isfinite(cov).all(-1).all(-1)
    U, S, Vh = torch.linalg.svd(torch.where(finite[:, None, None], cov, torch.zeros_like(cov)))
    flip = torch.linalg.det(U) * torch.linalg.det(Vh) < 0.0
    sign = torch.ones_like(S)
    sign[:, -1] = torch.where(flip, -1.0, 1.0).to(S.dtype)
    S, U = S * sign, U * sign[:, None, :]
    var = (cs * cs).sum(1).sum(-1) / n                                      # sum over axes of the population variance
    scale = S.sum(-1) / var
    rotation = (U @ Vh).transpose(1, 2)
    translation = mu_d - torch.einsum("bi,bij->bj", mu_s, scale[:, None, None] * rotation)
    transform = torch.zeros(src.shape[0], 4, 4, dtype=src.dtype, device=src.device)
    transform[:, :3, :3] = scale[:, None, None] * rotation                  # diag(s, s, s) @ R
    transform[:, :3, 3] = translation
    transform[:, 3, 3] = 1.0
    nan = torch.full_like(scale, float("nan"))
    scale = torch.where(finite, scale, nan)                                 # the reference raises on NaN input; here: no pose
    return scale, rotation, translation, torch.where(finite[:, None, None], transform, nan[:, None, None])


def _fit_hip(xyz, npcs, offsets, picks, P, M, H, stop_thrsh):
    import ctypes
    from .. import _C
    from ..hip_ops import _stream, _ws, i64, i32, ptr, szt
    dev, f64 = xyz.device, torch.float64
    out = {"valid": torch.empty(P, dtype=torch.uint8, device=dev), "scale": torch.empty(P, dtype=f64, device=dev),
           "rotation": torch.empty(P, 3, 3, dtype=f64, device=dev), "translation": torch.empty(P, 3, dtype=f64, device=dev),
           "transform": torch.empty(P, 4, 4, dtype=f64, device=dev), "bbox": torch.empty(P, 8, 3, dtype=f64, device=dev),
           "inlier_mask": torch.empty(M, dtype=torch.uint8, device=dev),
           "best_iteration": torch.empty(P, dtype=torch.int64, device=dev), "residual": torch.empty(P, H, dtype=f64, device=dev)}
    L = _C.lib()
    ws = _ws(L.gpn_pose_fit_ws_bytes(i64(P), i32(H)), dev)
    _C.check(L.gpn_pose_fit(ptr(xyz), ptr(npcs), ptr(offsets), ptr(picks), i64(P), i64(M), i32(H), ctypes.c_double(stop_thrsh),
                            ptr(out["valid"]), ptr(out["scale"]), ptr(out["rotation"]), ptr(out["translation"]),
                            ptr(out["transform"]), ptr(out["bbox"]), ptr(out["inlier_mask"]), ptr(out["best_iteration"]),
                            ptr(out["residual"]), ptr(ws), szt(ws.numel()), _stream()), "gpn_pose_fit")
    out["valid"], out["inlier_mask"] = out["valid"].bool(), out["inlier_mask"].bool()
    return out


@torch.no_grad()
def estimate_pose_from_npcs_batched(xyz: torch.Tensor, npcs: torch.Tensor, offsets: torch.Tensor,
                                    picks: Optional[torch.Tensor] = None, stop_thrsh: float = 0.5, max_iters: int = 100,
                                    chunk: int = 10) -> Dict[str, torch.Tensor]:
    """xyz, npcs [M,3] (points of all proposals, proposal p = rows offsets[p]:offsets[p+1]); picks [P, H, 5] or None (drawn).
    -> dict: valid [P] bool, scale [P], rotation [P,3,3], translation [P,3], transform [P,4,4], bbox [P,8,3] (NaN where not
    valid), inlier_mask [M] bool - per proposal what ``estimate_pose_from_npcs`` returns."""
    dev, f64 = xyz.device, torch.float64
    offsets = offsets.to(dev).long()
    sizes = offsets[1:] - offsets[:-1]
    P, M = sizes.shape[0], xyz.shape[0]
    if picks is None:
        picks = draw_picks(sizes.tolist(), max_iters)
    picks = picks.to(dev).long()
    H = picks.shape[1]
    if xyz.is_cuda:  # the HIP entry point (csrc/pose.hip): two launches for all proposals and hypotheses
        from .. import backend
        if backend.raw().name != "hip":
            raise RuntimeError("estimate_pose_from_npcs_batched on a GPU tensor needs the HIP library as the operator backend")
        return _fit_hip(xyz.to(f64).contiguous(), npcs.to(f64).contiguous(), offsets.contiguous(), picks.contiguous(), P, M, H,
                        stop_thrsh)
    src_pts, dst_pts = npcs.to(f64), xyz.to(f64)
    pid = torch.repeat_interleave(torch.arange(P, device=dev), sizes, output_size=M)
    local = torch.arange(M, device=dev) - offsets[:-1][pid]
    single = sizes == 1
    # a duplicated single point: both picks 0 and 1 address the point itself
    pick_rows = offsets[:-1][:, None, None] + torch.where(single[:, None, None], torch.zeros_like(picks), picks)
    s_norm = torch.zeros(P, dtype=f64, device=dev).index_add_(0, pid, src_pts.norm(dim=1)) / sizes
    t_norm = torch.zeros(P, dtype=f64, device=dev).index_add_(0, pid, dst_pts.norm(dim=1)) / sizes
    pass_thrsh = torch.maximum(s_norm / t_norm, t_norm / s_norm)

    # ---- all P x H five-point hypotheses ----
    hyp_T = _umeyama(src_pts[pick_rows].reshape(P * H, 5, 3), dst_pts[pick_rows].reshape(P * H, 5, 3))[3].reshape(P, H, 4, 4)
    residual = torch.empty(P, H, dtype=f64, device=dev)
    for h0 in range(0, H, chunk):
        T = hyp_T[:, h0:h0 + chunk][pid]                                     # [M, c, 4, 4]
        err = dst_pts[:, None, :] - (torch.einsum("mcij,mj->mci", T[:, :, :3, :3], src_pts) + T[:, :, :3, 3])
        sq = torch.zeros(P, err.shape[1], dtype=f64, device=dev).index_add_(0, pid, (err * err).sum(-1))
        residual[:, h0:h0 + chunk] = torch.sqrt(sq)
    # ---- the sequential choice, after the fact ----
    res = torch.where(torch.isnan(residual) | (residual >= 1e10), torch.full_like(residual, float("inf")), residual)
    running = torch.cummin(res, dim=1)[0]
    below = running < stop_thrsh
    stop = torch.where(below.any(1), below.to(torch.int64).argmax(1), torch.full((P,), H - 1, device=dev))
    considered = torch.arange(H, device=dev)[None, :] <= stop[:, None]
    best = torch.where(considered, res, torch.full_like(res, float("inf"))).argmin(1)
    never = ~torch.isfinite(res.gather(1, best[:, None]).squeeze(1))        # no hypothesis ever accepted: ratio stays 0, no pose
    best_T = hyp_T[torch.arange(P, device=dev), best][pid]
    err = dst_pts - (torch.einsum("mij,mj->mi", best_T[:, :3, :3], src_pts) + best_T[:, :3, 3])
    inlier = (err.norm(dim=1) < pass_thrsh[pid]) & ~never[pid]
    counted = inlier & (local != 0)                                          # np.count_nonzero on the INDEX array
    ratio = torch.zeros(P, dtype=f64, device=dev).index_add_(0, pid, counted.to(f64)) / sizes.to(f64)
    n_in = torch.zeros(P, dtype=f64, device=dev).index_add_(0, pid, inlier.to(f64))
    valid = (ratio >= 0.01) & (n_in > 0) & ~never & ~single

    # ---- final fit on the inliers: padded [P, Lmax] layout with 0/1 weights ----
    L = int(sizes.max()) if P else 0
    pad_s = torch.zeros(P, L, 3, dtype=f64, device=dev)
    pad_d = torch.zeros(P, L, 3, dtype=f64, device=dev)
    pad_w = torch.zeros(P, L, dtype=f64, device=dev)
    pad_s[pid, local], pad_d[pid, local], pad_w[pid, local] = src_pts, dst_pts, inlier.to(f64)
    scale, rotation, translation, transform = _umeyama(pad_s, pad_d, pad_w)
    valid &= torch.isfinite(scale)
    in_frame = torch.einsum("mi,mij->mj", dst_pts - translation[pid], torch.linalg.pinv(rotation)[pid]) / scale[pid, None]
    mag = torch.where(inlier[:, None], in_frame.abs(), torch.zeros_like(in_frame))
    half = torch.zeros(P, 3, dtype=f64, device=dev).scatter_reduce_(0, pid[:, None].expand(-1, 3), mag, "amax")
    signs = torch.tensor(_CORNER_SIGNS, dtype=f64, device=dev)
    bbox = torch.einsum("pki,pij->pkj", signs[None] * half[:, None, :] * scale[:, None, None], rotation) + translation[:, None, :]
    nan = float("nan")
    bad = ~valid
    return {"valid": valid, "scale": scale.masked_fill(bad, nan), "rotation": rotation.masked_fill(bad[:, None, None], nan),
            "translation": translation.masked_fill(bad[:, None], nan), "transform": transform.masked_fill(bad[:, None, None], nan),
            "bbox": bbox.masked_fill(bad[:, None, None], nan), "inlier_mask": inlier & valid[pid],
            "best_iteration": best, "residual": residual}
