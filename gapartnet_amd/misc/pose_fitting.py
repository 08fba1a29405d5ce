"""NPCS -> camera-frame similarity fit and oriented bounding box (reference: gapartnet/misc/pose_fitting.py:4-147).

CPU numpy, as in the reference (it runs only in test-time visualisation, SURVEY.md §8f rank 3: a batched GPU version
is a "next" row).  Same function names, argument meaning and return tuples; RANSAC draws from the global numpy
generator in the reference's order, so results agree for the same seed.  tests/test_golden.py pins the Umeyama fit
against values captured from the reference function.
"""
import numpy as np

_CORNER_SIGNS = np.asarray([[-1, -1, -1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1], [1, 1, -1], [1, -1, 1], [-1, 1, 1],
                            [1, 1, 1]], dtype=np.float64)


def estimate_similarity_umeyama(source_hom: np.ndarray, target_hom: np.ndarray):
    """least-squares similarity (uniform scale s, rotation R, translation t) with target ~ s R^T-applied source in the
    reference's row-vector convention: target_xyz = source_xyz @ (s R) + t  (pose_fitting.py:4-43).
    Inputs are homogeneous [4, N]; returns (scale [3], rotation [3,3], translation [3], transform [4,4])."""
    src, dst = source_hom[:3], target_hom[:3]
    n = src.shape[1]
    mu_s, mu_d = src.mean(axis=1), dst.mean(axis=1)
    cov = (dst - mu_d[:, None]) @ (src - mu_s[:, None]).T / n
    if np.isnan(cov).any():
        raise RuntimeError("There are NANs in the input.")
    U, S, Vh = np.linalg.svd(cov, full_matrices=True)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0.0:  # reflection: flip the weakest axis
        S[-1] = -S[-1]
        U[:, -1] = -U[:, -1]
    s = S.sum() / np.var(src, axis=1).sum()
    rotation = (U @ Vh).T
    translation = mu_d - mu_s.dot(s * rotation)
    scale = np.array([s, s, s])
    transform = np.identity(4)
    transform[:3, :3] = np.diag(scale) @ rotation
    transform[:3, 3] = translation
    return scale, rotation, translation, transform


def evaluate_model(out_transform: np.ndarray, source_hom: np.ndarray, target_hom: np.ndarray, pass_thrsh: float):
    """residual norm, inlier ratio and inlier indices of a candidate transform (pose_fitting.py:46-51).  The ratio
    counts NON-ZERO inlier indices, as the reference does (np.count_nonzero on the index array)."""
    per_point = np.linalg.norm((target_hom - out_transform @ source_hom)[:3], axis=0)
    inliers = np.where(per_point < pass_thrsh)[0]
    return np.linalg.norm(per_point), np.count_nonzero(inliers) / source_hom.shape[1], inliers


def get_RANSAC_inliers(source_hom: np.ndarray, target_hom: np.ndarray, max_iters: int, pass_thrsh: float,
                       stop_thrsh: float):
    """5-point RANSAC keeping the hypothesis with the smallest total residual (pose_fitting.py:54-80)."""
    best_residual, best_ratio = 1e10, 0
    best_idx = np.arange(source_hom.shape[1])
    for _ in range(max_iters):
        pick = np.random.randint(source_hom.shape[1], size=5)
        transform = estimate_similarity_umeyama(source_hom[:, pick], target_hom[:, pick])[3]
        residual, ratio, idx = evaluate_model(transform, source_hom, target_hom, pass_thrsh)
        if residual < best_residual:
            best_residual, best_ratio, best_idx = residual, ratio, idx
        if best_residual < stop_thrsh:
            break
    return best_ratio, best_idx


def estimate_similarity_transform(source: np.ndarray, target: np.ndarray, stop_thrsh: float = 0.5, max_iters: int = 100):
    """RANSAC + Umeyama on the inliers; the pass threshold is max(|src|/|tgt|, |tgt|/|src|) of the mean point norms
    (pose_fitting.py:83-118).  Returns (scale, rotation, translation, transform, inlier_idx) or Nones."""
    if source.shape[0] == 1:
        source, target = np.repeat(source, 2, axis=0), np.repeat(target, 2, axis=0)
    src_h = np.hstack([source, np.ones([source.shape[0], 1])]).T
    dst_h = np.hstack([target, np.ones([target.shape[0], 1])]).T
    s_norm = np.mean(np.linalg.norm(source, axis=1))
    t_norm = np.mean(np.linalg.norm(target, axis=1))
    pass_thrsh = max(s_norm / t_norm, t_norm / s_norm)
    ratio, idx = get_RANSAC_inliers(src_h, dst_h, max_iters=max_iters, pass_thrsh=pass_thrsh, stop_thrsh=stop_thrsh)
    if ratio < 0.01:
        return np.asarray([None, None, None]), None, None, None, None
    scale, rotation, translation, transform = estimate_similarity_umeyama(src_h[:, idx], dst_h[:, idx])
    return scale, rotation, translation, transform, idx


def estimate_pose_from_npcs(xyz, npcs):
    """fit npcs -> xyz and return the 8 corners of the NPCS-aligned box that covers the inliers
    (pose_fitting.py:121-147): (bbox [8,3], scale, rotation, translation, transform, inlier_idx)."""
    scale, rotation, translation, transform, idx = estimate_similarity_transform(npcs, xyz)
    if scale[0] is None:
        return None, np.asarray([None, None, None]), None, None, None, idx
    in_npcs_frame = np.dot(xyz - translation, np.linalg.pinv(rotation)) / scale[0]
    half = np.abs(in_npcs_frame[idx]).max(0)
    bbox = np.dot(_CORNER_SIGNS * half * scale[0], rotation) + translation
    return bbox, scale, rotation, translation, transform, idx
