"""ctypes binding of ``libgpn_hip.so`` (the C-ABI declared in ``include/gpn.h``).

The library is the product: there is no CPU fallback.  ``lib()`` raises if the shared object has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C gapartnet_amd/csrc``).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgpn_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None

_SIZE_T_FUNCS = (
    "gpn_voxelize_ws_bytes", "gpn_voxelize_scenes_ws_bytes", "gpn_rulebook_subm3_ws_bytes", "gpn_rulebook_down_ws_bytes",
    "gpn_rulebook_down_lists_ws_bytes", "gpn_rulebook_level_counts_ws_bytes", "gpn_spconv_fwd_ws_bytes", "gpn_spconv_fwd_w_ws_bytes", "gpn_net_ws_bytes", "gpn_linear_bwd_ws_bytes", "gpn_ball_query_grid_ws_bytes", "gpn_point_losses_ws_bytes", "gpn_rulebook_tile_order_ws_bytes", "gpn_bn_ws_bytes", "gpn_spconv_wgrad_ws_bytes", "gpn_ccl_ws_bytes", "gpn_nms_ws_bytes", "gpn_pn2_furthest_point_sampling_ws_bytes", "gpn_pose_fit_ws_bytes", "gpn_proposals_revoxelize_ws_bytes", "gpn_proposals_postprocess_ws_bytes", "gpn_backbone_prepare_arena_bytes", "gpn_scene_prepare_ws_bytes",
)


class GpnError(RuntimeError):
    pass


def build(jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 into gapartnet_amd/libgpn_hip.so (in-tree)."""
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j", str(jobs)])
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise GpnError(
                f"{SO_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        # torch first: it brings its own libamdhip64; loading ours before it would put two HIP runtimes in the
        # process (kernel launches then fail with "no ROCm-capable device")
        import torch  # noqa: F401
        _lib = ctypes.CDLL(SO_PATH)
        _lib.gpn_last_error.restype = ctypes.c_char_p
        _lib.gpn_entry_point_name.restype = ctypes.c_char_p
        for name in _SIZE_T_FUNCS:
            getattr(_lib, name).restype = ctypes.c_size_t
        # prototypes of the per-layer hot calls: with argtypes set, plain Python ints (data_ptr()) and floats are
        # converted by ctypes itself, which is several times cheaper than building c_void_p / c_int64 objects per call
        vp, i64t, i32t, f32t, st = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
        _lib.gpn_spconv_fwd_w.argtypes = [vp, vp, i32t, i32t, i32t, i32t, vp, i64t, vp, vp, st, vp]
        _lib.gpn_spconv_wgrad.argtypes = [vp, vp, vp, vp, vp, i32t, i64t, i32t, i32t, i32t, vp, vp, st, vp]
        _lib.gpn_bn_fwd_train.argtypes = [vp, vp, vp, vp, i64t, i32t, f32t, f32t, i32t, vp, vp, vp, vp, vp, vp, st, vp]
        _lib.gpn_bn_fwd_eval.argtypes = [vp, vp, vp, vp, vp, vp, i64t, i32t, i32t, vp, vp]
        _lib.gpn_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64t, i32t, i32t, i32t, vp, vp, vp, vp, vp, st, vp]
        _lib.gpn_gather_rows.argtypes = [vp, vp, i64t, i32t, vp, vp]
        _lib.gpn_scatter_rows_csr.argtypes = [vp, vp, vp, i64t, i32t, vp, vp]
        _lib.gpn_spconv_tiles_min_tiles.argtypes = [i64t]
        _lib.gpn_spconv_tiles_min_tiles.restype = i64t
        _lib.gpn_spconv_direct_split.argtypes = [i64t, i64t]
        _lib.gpn_spconv_msplit.argtypes = [i32t, i32t, i32t]
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().gpn_last_error().decode("utf-8", "replace")
        raise GpnError(f"{what or 'gpn call'} failed (code {rc}): {msg}")


# argument helpers ------------------------------------------------------------------------------------
def ptr(t):
    """device pointer of a (contiguous) torch tensor, or NULL for None."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def i64(v):
    return ctypes.c_int64(int(v))


def i32(v):
    return ctypes.c_int(int(v))


def f32(v):
    return ctypes.c_float(float(v))


def szt(v):
    return ctypes.c_size_t(int(v))


def host_f32x3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def host_i32x3(v):
    return (ctypes.c_int32 * 3)(*[int(x) for x in v])
