"""CPU: the C-ABI boundary -- the shared library loads and exports every symbol that
include/ccn_hip.h declares; the product never falls back to a CPU path; the product does not
import the oracle."""
import ast
import ctypes
import os
import re

import pytest
import torch

from tests.util import ROOT


def test_library_exports_every_declared_symbol():
    from curvecloudnet_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 45
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.ccn_abi_version() == 4
    assert lib.ccn_stats_rows(129) == 2
    assert lib.ccn_frnn_grid_bytes(2, 1000) > 0 and lib.ccn_curve_fps_workspace_bytes(1000) > 0


def test_header_cites_reference_call_sites():
    text = open(os.path.join(ROOT, "include", "ccn_hip.h")).read()
    for cite in ("point_ops.py:47-54", "point_ops.py:20-44", "fast_conv1d.py:190-205", "fps_ops.py:16-39",
                 "point_ops.py:143-193", "point_ops.py:196-260", "point_ops.py:459", "dgcnn.py:158-207",
                 "point_conv.py:60-93"):
        assert cite in text, cite
    assert "extern \"C\"" in text
    assert "#include <torch" not in text and "at::Tensor" not in text      # plain pointers and sizes only


def test_argument_errors_are_reported_not_thrown():
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    rc = lib.ccn_gemm_nt(None, 0, None, 0, None, None, 0, 4, 4, 4, None, None)
    assert rc == -1 and b"gemm_nt" in lib.ccn_last_error()
    rc = lib.ccn_frnn_query(None, None, None, 1, 1, 1, None, 1, None, None, None, None)
    assert rc == -1 and b"frnn_query" in lib.ccn_last_error()


def test_no_cpu_fallback():
    from curvecloudnet_amd import ops
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear_bn_act(x, torch.zeros(2, 3), None, None, False, None)
    with pytest.raises(TypeError):
        ops.fast_knn(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), torch.tensor([4]), torch.tensor([4]), 2, 0.1)
    with pytest.raises(RuntimeError):
        ops.batch2ptr(torch.tensor([0, 0, 1]))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "curvecloudnet_amd")
    for fn in os.listdir(pkg):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.startswith("oracle") and "torch_ref" not in n, (fn, n)


def test_frnn_compat_offers_the_third_party_signatures():
    """reference call sites: src/models/utils/point_ops.py:459 (frnn_grid_points, six positional arguments, four results) and
    src/models/modules/dgcnn.py:172 (frnn_gather, three positional arguments)."""
    import inspect
    import sys
    from curvecloudnet_amd import frnn_compat
    sig = inspect.signature(frnn_compat.frnn_grid_points)
    assert list(sig.parameters)[:6] == ["points1", "points2", "lengths1", "lengths2", "K", "r"]
    for name in ("grid", "return_nn", "return_sorted", "radius_cell_ratio"):
        assert name in sig.parameters
    assert list(inspect.signature(frnn_compat.frnn_gather).parameters) == ["x", "idxs", "lengths"]
    saved = sys.modules.get("frnn")
    try:
        assert frnn_compat.install() is frnn_compat
        import frnn
        assert frnn.frnn_grid_points is frnn_compat.frnn_grid_points and frnn.frnn_gather is frnn_compat.frnn_gather
        with pytest.raises(RuntimeError, match="GPU only"):
            frnn.frnn_gather(torch.zeros(1, 4, 3), torch.zeros(1, 4, 2, dtype=torch.int64), torch.tensor([4]))
    finally:
        if saved is None:
            sys.modules.pop("frnn", None)
        else:
            sys.modules["frnn"] = saved


def test_debug_hooks_go_through_one_door():
    """include/ccn_hip_debug.h: ccn_debug_set(key, value) serialises every A/B / test hook (VERDICT r5 Next 9); unknown keys and
    out-of-range values are refused with a message."""
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    for key in (b"gemm_use_dma", b"gemm_pair_opt", b"frnn_query_mode", b"fps_use_cluster", b"fps_debug_fault", b"gemm_tn_background"):
        assert lib.ccn_debug_set(key, 1 if key in (b"gemm_use_dma", b"fps_use_cluster") else 0) == 0, key
    assert lib.ccn_debug_set(b"no_such_hook", 1) < 0 and b"unknown key" in lib.ccn_last_error()
    assert lib.ccn_debug_set(b"gemm_pair_opt", 1 << 40) < 0 and b"out of range" in lib.ccn_last_error()
    assert lib.ccn_debug_set(None, 0) < 0
