"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/rass_engine.h declares; the ctypes table binds exactly that set; no
compute call is made here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rass_engine.h")


def _declared_functions():
    text = open(HEADER, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rass_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built_lib():
    from rassengine_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "rassengine_amd", "csrc")], check=True)
    return _native


def test_header_declares_the_hot_path_entry_points():
    names = _declared_functions()
    for required in ("rass_engine_create", "rass_index_open", "rass_index_add", "rass_index_count",
                     "rass_index_search", "rass_index_search_device", "rass_scan_topk_f32", "rass_topk_merge",
                     "rass_normalize_rows_f32", "rass_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], check=True, capture_output=True,
                         text=True).stdout
    exported = set(re.findall(r" T (rass_[a-z0-9_]+)", out))
    declared = set(_declared_functions())
    assert declared - exported == set(), f"declared but not exported: {sorted(declared - exported)}"
    assert exported - declared == set(), f"exported but not declared: {sorted(exported - declared)}"


def test_ctypes_table_matches_header(built_lib):
    assert set(built_lib.SIGNATURES) == set(_declared_functions())


def test_library_loads_without_gpu_and_reports_errors(built_lib):
    L = built_lib.lib()  # binds every symbol; raises on a mismatch
    assert L.rass_abi_version() == 1
    assert L.rass_last_error() is not None
    # pure host-side queries only: no kernels, no device memory
    assert L.rass_scan_workspace_bytes(32, 32) > 0
    assert L.rass_scan_workspace_bytes(33, 10) == 0
    assert L.rass_scan_kernel_name(1024, 32) == b"scan_topk_f32_kernel<8, 2, 0, false>"
    assert L.rass_scan_kernel_name(1024, 7) == b"scan_topk_f32_kernel<8, 1, 0, false>"
    assert L.rass_scan_kernel_name(5000, 1) == b""


def test_no_dual_hip_runtime_rpath_surprise(built_lib):
    """The library must name the HIP runtime by soname only, so that inside a torch process it
    binds to the libamdhip64.so.7 torch already mapped."""
    out = subprocess.run(["readelf", "-d", built_lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "libamdhip64.so.7" in out
    assert "libtorch" not in out  # no torch types / libs behind the C ABI


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under rassengine_amd/ may import it."""
    pkg = os.path.join(ROOT, "rassengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
            if fn.endswith((".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "rass_oracle_" not in src and not re.search(r"#include\s*[\"<].*oracle", src), fn


def test_torch_library_ops_are_registered_and_refuse_cpu_tensors():
    """north_star: the kernels "under PyTorch-ROCm custom ops" — the three stateless launchers are torch.library ops
    (namespace rass) with shape-inferring fake implementations; like everything else they have no CPU path."""
    import pytest
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from rassengine_amd import ops  # noqa: F401  (registers)
    for name in ("scan_topk_packed", "topk_merge", "normalize_rows"):
        assert hasattr(torch.ops.rass, name)
    with FakeTensorMode():
        s, i = torch.ops.rass.scan_topk_packed(torch.empty((64, 1024)), 60, torch.empty((5, 1024)), 10)
        assert s.shape == (5, 10) and i.dtype == torch.int64
        ms, mi = torch.ops.rass.topk_merge(torch.empty((8, 5, 10)), torch.empty((8, 5, 10), dtype=torch.int64))
        assert ms.shape == (5, 10) and mi.shape == (5, 10)
        assert torch.ops.rass.normalize_rows(torch.empty((3, 1000)), 1024).shape == (3, 1024)
    with pytest.raises(ValueError, match="no CPU path"):
        torch.ops.rass.normalize_rows(torch.zeros((2, 8)))
    # the encoder's kernels (VERDICT r3 #4 / housekeeping c): rass::gemm_bf16, rass::attention_bf16, rass::encode
    for name in ("gemm_bf16", "attention_bf16", "attention_out_bf16", "encode"):
        assert hasattr(torch.ops.rass, name)
    with FakeTensorMode():
        y = torch.ops.rass.gemm_bf16(torch.empty((300, 1024), dtype=torch.bfloat16), torch.empty((4096, 1024), dtype=torch.bfloat16),
                                     torch.empty((4096,)), None, 2)
        assert y.shape == (300, 4096) and y.dtype == torch.bfloat16
        c = torch.ops.rass.attention_bf16(torch.empty((700, 3072), dtype=torch.bfloat16), torch.empty((4,), dtype=torch.int32), 512, 16)
        assert c.shape == (700, 1024) and c.dtype == torch.bfloat16
        yo = torch.ops.rass.attention_out_bf16(torch.empty((12, 3072), dtype=torch.bfloat16), torch.empty((2,), dtype=torch.int32), 16,
                                               torch.empty((1024, 1024), dtype=torch.bfloat16), torch.empty((1024,)),
                                               torch.empty((12, 1024), dtype=torch.bfloat16))
        assert yo.shape == (12, 1024) and yo.dtype == torch.bfloat16
        e = torch.ops.rass.encode(0, torch.empty((700,), dtype=torch.int32), torch.empty((4,), dtype=torch.int32), 512, 1024)
        assert e.shape == (3, 1024) and e.dtype == torch.float32
    with pytest.raises(ValueError, match="no CPU path"):
        torch.ops.rass.gemm_bf16(torch.zeros((2, 64), dtype=torch.bfloat16), torch.zeros((128, 64), dtype=torch.bfloat16),
                                 torch.zeros((128,)), None, 0)
