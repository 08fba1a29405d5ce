"""The drop-in boundary: libgpn_hip.so must load (no GPU needed for that) and export every symbol include/gpn.h
declares; argument checking must work without touching the device; the product path must refuse CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from gapartnet_amd import _C
    if not os.path.exists(_C.SO_PATH):
        _C.build()
    return _C.lib()


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("gpn_voxelize", "gpn_rulebook_subm3", "gpn_rulebook_down", "gpn_spconv_fwd", "gpn_spconv_wgrad",
                 "gpn_ball_query", "gpn_ccl", "gpn_segmented_reduce", "gpn_segmented_maxpool_fwd", "gpn_instance_iou",
                 "gpn_nms", "gpn_pn2_furthest_point_sampling", "gpn_pn2_three_nn", "gpn_pn2_ball_query"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_entry_point_registry_matches_header(lib):
    n = lib.gpn_num_entry_points()
    names = {lib.gpn_entry_point_name(i).decode() for i in range(n)}
    declared = set(declared_symbols()) - {"gpn_num_entry_points", "gpn_entry_point_name"}
    assert declared <= names | {"gpn_voxelize_ex"}, declared - names
    assert lib.gpn_version() >= 1


def test_the_library_reads_no_environment_switches(lib):
    """round 6: the library has no ``getenv`` left.  Every switch that survived is an entry point (gpn_spconv_msplit,
    gpn_spconv_tiles_min_tiles, gpn_spconv_direct_split, gpn_net_bn_fusion, gpn_net_wgrad_group,
    gpn_proposals_postprocess_lds_proposals) that a -m gpu test flips; the ~20 environment readers of rounds 2 - 5 were either
    measured and fixed as constants or duplicated such an entry point (and one of them, for a whole round, silently read another
    one's variable: profiles/r03_findings.md)."""
    csrc = os.path.join(ROOT, "gapartnet_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), fn
    for name in ("gpn_spconv_msplit", "gpn_spconv_tiles_min_tiles", "gpn_spconv_direct_split", "gpn_net_bn_fusion",
                 "gpn_net_wgrad_group", "gpn_proposals_postprocess_lds_proposals"):
        assert hasattr(lib, name), name
    # queries leave the settings alone and report the defaults
    assert lib.gpn_net_bn_fusion(-1) == 1 and lib.gpn_net_wgrad_group(-1) == 4
    assert lib.gpn_spconv_msplit(-1, -1, -1) == 1
    lib.gpn_spconv_tiles_min_tiles.restype = ctypes.c_int64
    assert lib.gpn_spconv_tiles_min_tiles(ctypes.c_int64(-1)) == 4096
    lib.gpn_proposals_postprocess_lds_proposals.restype = ctypes.c_int64
    assert lib.gpn_proposals_postprocess_lds_proposals(ctypes.c_int64(-1)) == 131072


def test_argument_errors_do_not_touch_the_device(lib):
    lib.gpn_last_error.restype = ctypes.c_char_p
    rc = lib.gpn_spconv_fwd(None, None, None, ctypes.c_int(27), ctypes.c_int64(10), ctypes.c_int(15), ctypes.c_int(16), None,
                            None, ctypes.c_size_t(0), None)
    assert rc == 1 and b"bad argument" in lib.gpn_last_error()
    rc = lib.gpn_segmented_reduce(None, None, None, ctypes.c_int64(4), ctypes.c_int(3), ctypes.c_int(7), None, None)
    assert rc == 1
    assert lib.gpn_spconv_wgrad_ws_bytes(ctypes.c_int(27), ctypes.c_int(32), ctypes.c_int(32), ctypes.c_int64(100000)) > 0


def test_workspace_queries_are_pure(lib):
    assert lib.gpn_voxelize_ws_bytes(ctypes.c_int64(20000), ctypes.c_int(6)) > 20000 * 8
    assert lib.gpn_rulebook_subm3_ws_bytes(ctypes.c_int64(1000)) > 27 * 1000 * 4
    assert lib.gpn_ccl_ws_bytes(ctypes.c_int64(1000)) >= 3 * 4000


def test_product_path_has_no_cpu_fallback():
    from gapartnet_amd import _C, hip_ops
    with pytest.raises(_C.GpnError):
        hip_ops.segmented_reduce(torch.zeros(4, 3), torch.zeros(1, dtype=torch.int32), torch.ones(1, dtype=torch.int32), "sum")
    from gapartnet_amd import backend
    assert backend.raw() is hip_ops


def test_executor_struct_layouts_match_the_header(tmp_path):
    """the numpy mirrors in network/net_exec.py must have exactly the C layout of the gpn_net_* structs"""
    import subprocess
    from gapartnet_amd.network import net_exec as NX
    src = tmp_path / "sizes.c"
    fields = {
        "gpn_net_slot_t": (NX.SLOT_DT, ["data", "grad", "rows", "channels", "grad_state", "rows_dev", "rows_plan"]),
        "gpn_net_rulebook_t": (NX.RB_DT, ["nbr", "nbr_t", "nbr_p", "perm", "nbr_t_p", "perm_t", "pair_src", "pair_dst",
                                          "tile_off", "n_src", "n_dst", "K", "reverse_taps"]),
        "gpn_net_conv_t": (NX.CONV_DT, ["W", "dW", "cin", "cout"]),
        "gpn_net_bn_t": (NX.BN_DT, ["weight", "bias", "running_mean", "running_var", "save_mean", "save_invstd",
                                    "dweight", "dbias", "eps", "momentum", "C", "reserved"]),
        "gpn_net_op_t": (NX.OP_DT, ["kind", "src0", "src1", "dst", "rulebook", "param", "flags", "reserved"]),
    }
    body = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/gpn.h"', "int main(void) {"]
    for name, (_, names) in fields.items():
        body.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in names:
            body.append(f'  printf(" %zu", offsetof({name}, {f}));')
        body.append('  printf("\\n");')
    body.append("  return 0; }")
    src.write_text("\n".join(body))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])  # also proves the header is plain C
    for line in subprocess.check_output([str(exe)], text=True).strip().splitlines():
        name, size, *offs = line.split()
        dt, names = fields[name]
        assert dt.itemsize == int(size), (name, dt.itemsize, size)
        assert [dt.fields[f][1] for f in names] == [int(o) for o in offs], name


def test_executor_rejects_malformed_programs_without_touching_the_device(lib):
    """gpn_net_forward validates the whole program (slot / table indices, shapes) before its first launch"""
    import numpy as np
    from gapartnet_amd.network import net_exec as NX
    lib.gpn_last_error.restype = ctypes.c_char_p

    def vp(a):
        return ctypes.c_void_p(a.ctypes.data)

    slots = np.zeros(2, NX.SLOT_DT)
    slots["rows"], slots["channels"] = 100, 16
    rbs = np.zeros(1, NX.RB_DT)
    rbs["n_src"], rbs["n_dst"], rbs["K"] = 100, 100, 27
    convs = np.zeros(1, NX.CONV_DT)
    convs["cin"], convs["cout"] = 16, 32  # does not match slot 1's 16 channels
    bns = np.zeros(1, NX.BN_DT)
    ops = np.zeros(1, NX.OP_DT)
    ops[0] = (NX.OP_CONV, 0, -1, 1, 0, 0, 0, 0)

    def run(ops_arr):
        return lib.gpn_net_forward(vp(ops_arr), len(ops_arr), vp(slots), 2, vp(rbs), 1, vp(convs), 1, vp(bns), 1, 1, None,
                                   ctypes.c_size_t(0), None)

    assert run(ops) == 1 and b"does not match" in lib.gpn_last_error()
    bad = ops.copy()
    bad["dst"] = 7
    assert run(bad) == 1 and b"out of range" in lib.gpn_last_error()
    bad = ops.copy()
    bad["kind"] = 9
    assert run(bad) == 1 and b"unknown op" in lib.gpn_last_error()
    assert lib.gpn_net_forward(None, 1, None, 0, None, 0, None, 0, None, 0, 1, None, ctypes.c_size_t(0), None) == 1
