"""The C-ABI library must load and export exactly the entry points include/mort_hip.h declares
(no compute calls here: this runs without a GPU), and must refuse to work without a device."""
import ctypes as C
import os
import re

import pytest

from mort_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(hip.LIB_PATH):
        pytest.fail(f"{hip.LIB_PATH} missing: run __graft_entry__.build()")
    L = C.CDLL(hip.LIB_PATH)
    names = declared("mort_hip.h", "mort_hip_")
    assert names == sorted(hip.EXPORTS)
    for n in names:
        assert getattr(L, n) is not None


def test_host_library_exports_every_declared_symbol():
    from mort_amd import host
    L = host.lib()
    for n in declared("mort_host.h", "mort_"):
        assert getattr(L, n) is not None, n


def test_status_strings_and_argument_checks():
    L = hip.lib()
    assert L.mort_hip_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert L.mort_hip_strerror(code) not in (b"ok", b"unknown status")
    assert L.mort_hip_init(0, None) == -1  # MORT_ERR_INVALID: out pointer required
    assert L.mort_hip_upload_world(None, None) == -1
    assert L.mort_hip_rng_seed(None, 1, 4, 4) == -1


def test_no_gpu_means_no_render_path():
    """Without a device mort_hip_init must fail (no CPU fallback anywhere in the product)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.MortHipError) as e:
        hip.Context(0)
    assert e.value.status == -2  # MORT_ERR_NO_DEVICE
