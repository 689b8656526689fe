"""CPU: the C-ABI shared library loads and exports every symbol include/wfae.h
declares (no compute calls without a GPU), and the product path fails loudly —
it never falls back to a CPU implementation."""
import ctypes
import os

import pytest
import torch

from weatherforecastingtoolkit_amd import _lib


def test_header_parses_all_entry_points():
    d = _lib.parse_header()
    assert len(d) >= 35
    for must in ["wfae_conv1x1_fwd", "wfae_conv4x4s2_down", "wfae_conv4x4s2_up", "wfae_conv4x4s2_wgrad",
                 "wfae_dconv_fwd", "wfae_bn_stats_train", "wfae_bn_act_bwd", "wfae_sigmoid_l1_fwd", "wfae_ssim_fwd",
                 "wfae_adamw", "wfae_linear_fwd", "wfae_version", "wfae_last_error_string", "wfae_workspace_bytes"]:
        assert must in d, must
    # every stream-taking entry point has the stream as its LAST argument
    for name, (_, argtypes, argnames) in d.items():
        if "stream" in argnames:
            assert argnames[-1] == "stream", name


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libwfae.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(lib, name), f"libwfae.so does not export {name}"
    bound = _lib.load()
    assert bound.wfae_version() == 103
    assert bound.wfae_workspace_bytes(1 << 24) >= (1 << 24) * 4
    assert bound.wfae_last_error_string() is not None


def test_argument_validation_without_gpu():
    """Null pointers / bad shapes are rejected on the host before any launch."""
    lib = _lib.load()
    rc = lib.wfae_conv1x1_fwd(None, None, None, None, 0, None, 1, 1, 1, 1, None)
    assert rc == -2 and b"null" in lib.wfae_last_error_string()
    rc = lib.wfae_adamw(None, None, None, None, 10, 0.1, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.001, 1.0, None)
    assert rc == -2
    with pytest.raises(_lib.WfaeError):
        _lib.call("wfae_gelu_fwd", None, None, 16, None)


def test_product_path_has_no_cpu_fallback():
    from weatherforecastingtoolkit_amd import ops
    from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import Bottleneck
    with pytest.raises(_lib.WfaeError):
        ops.gelu_fwd(torch.zeros(8))
    with pytest.raises(_lib.WfaeError):
        Bottleneck(32)(torch.zeros(1, 32, 8, 8))


def test_product_never_imports_the_oracle():
    import pathlib
    root = pathlib.Path(_lib._PKG)
    for f in root.rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_host_layer_under_address_and_ub_sanitizers():
    """a host-only ASan + UBSan build of the C-ABI layer (csrc/build_asan.sh, no device code) is
    driven through argument validation, workspace carving and launch-geometry arithmetic of every kernel family at the
    model's real sizes (tests/asan_driver.py).  CPU box only: GPU sanitizers are not available on the pool."""
    import glob
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("sanitizer builds run on the CPU box only")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "weatherforecastingtoolkit_amd", "csrc")
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no shared ASan runtime in this image")
    subprocess.run(["bash", os.path.join(csrc, "build_asan.sh")], check=True, capture_output=True, timeout=600)
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "asan_driver.py"),
                        os.path.join(csrc, "build_asan", "libwfae_asan.so"), root],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert "no sanitizer report" in r.stdout


def test_mha_entry_points_validate_alignment_and_dropout():
    """wfae_mha_fwd reads qkv through float4 -> 16-byte alignment is part of the contract; wfae_mha_bwd
    checks the dropout probability like the forward (keep = 1/(1-p) would be inf)"""
    lib = _lib.load()
    p = 0x7F0000000000
    assert lib.wfae_mha_fwd(p + 4, p + 4096, p + 8192, 8, 64, 8, 8, 0, 0.0, 1, None) == -1
    assert b"aligned" in lib.wfae_last_error_string()
    assert lib.wfae_mha_bwd(p, p + 4096, p + 8192, p + 12288, 8, 64, 8, 8, 0, 1.0, 1, None) == -1
    assert b"dropout" in lib.wfae_last_error_string()
