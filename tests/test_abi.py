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


def test_state_machine_kernels_have_no_private_memory():
    """The code objects inside the built library (llvm-objdump --offloading + llvm-readelf --notes): every kernel's register allocation can
    host the workgroup its launch bounds promise, and mega_bvh_kernel / mega_gen_kernel at <= 768 threads neither spill a vector register nor
    touch private memory beyond a callee frame (round 2's 1 984 B per lane are gone, and must stay gone); the 1 024-thread
    builds trade a few spilled registers in the shade step for a fourth wave per SIMD (DESIGN.md 4.1 h) and are bounded instead
    (mega_bvh_kernel<1024>: at most 32 registers / 128 B).  Needs the ROCm LLVM tools, no GPU."""
    import shutil
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or not shutil.which("c++filt"):
        pytest.skip("ROCm LLVM tools not installed")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_resources.py"), hip.LIB_PATH, "--check"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    rows = {l.split()[0] + " " + " ".join(l.split()[1:-7]): l.split()[-7:] for l in p.stdout.splitlines() if l.startswith("mega_")}
    gen = [v for k, v in rows.items() if k.startswith("mega_gen_kernel<768") or k.startswith("mega_gen_kernel<512")]
    assert gen and all(v[3] == "0" and v[5] == "0" for v in gen), rows  # no spilled VGPR, 0 B of private memory
    wide = [v for k, v in rows.items() if k.startswith("mega_bvh_kernel<1024")]
    assert wide and all(int(v[3]) <= 32 and int(v[5]) <= 128 and v[1] == "128" for v in wide), rows  # 128 VGPRs = four waves per SIMD


@pytest.mark.gpu
def test_roofline_calibration_kernels(gpu_ctx):
    """mort_hip_calib_valu / mort_hip_calib_hbm_copy (bench.py's roofline calibration): a SIMD cannot issue a wave64 VALU instruction faster than
    the SIMD-32's two cycles, a lone wave is slower than a full SIMD, fp64 and packed fp32 are slower than fp32, and the copy moves at a plausible HBM rate."""
    one = gpu_ctx.calib_valu(1, 0)
    four = gpu_ctx.calib_valu(4, 0)
    f64 = gpu_ctx.calib_valu(4, 2)
    pk = gpu_ctx.calib_valu(4, 4)
    assert one["simds_seen"] == four["simds_seen"] == 1024 and four["resident_waves_per_simd"] == 4
    assert 1.9 < four["cycles_per_valu_per_simd"] < 3.0 < one["cycles_per_valu_per_simd"] < 6.0
    assert f64["cycles_per_valu_per_simd"] > 1.5 * four["cycles_per_valu_per_simd"]
    assert pk["cycles_per_valu_per_simd"] > 1.5 * four["cycles_per_valu_per_simd"]  # v_pk_fma_f32: two fma per lane at the issue cost of two instructions
    assert 1.0 < one["clock_ghz"] < 2.6
    assert 2000 < gpu_ctx.calib_hbm_copy(1 << 29, 2) < 8000


@pytest.mark.parametrize("sid", [1, 10])
def test_four_wide_tree_is_the_binary_tree_regrouped(sid):
    """The BVH megakernel's four-wide nodes (scene_compile.h collapse_own_tree) are this build's binary tree over the reference's
    leaf nodes with inner nodes opened in place: every leaf is reached exactly once, every box and margin is one of the binary
    tree's bit for bit, and the pending-children bound the kernel's LDS stack is sized for holds (host only, no GPU)."""
    import ctypes as C
    from mort_amd import hip, host
    world, _ = host.build_scene(sid, width=64, spp=1)
    fn = hip.lib().mort_hip_debug_own_tree
    fn.restype = C.c_int
    out = (C.c_int * 9)()
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert fn(C.cast(world.ptr, C.c_void_p), out) == 0
    n2, leaves, depth2, n4, stack4, reached, bad, slots, same = list(out)
    assert n2 == leaves - 1 and leaves >= 19 and depth2 <= 15
    assert 0 < n4 <= n2 // 2                    # three binary nodes fold into one in the best case
    assert reached == leaves and bad == 0
    assert slots == n4 - 1 + leaves              # every node but the root and every leaf is somebody's child
    assert slots >= 2.9 * n4                     # three of four slots in use on average (the lowest nodes hold two or three leaves)
    assert 3 <= stack4 <= 24
    assert same == 1


def test_four_wide_tree_on_small_and_random_worlds():
    """The same invariants on worlds the built-in scenes do not contain: a handful of spheres (coincident, concentric), every material kind,
    and random sphere fields of 5..400 spheres -- every leaf reached exactly once, boxes and margins the binary tree's, the pending-children
    bound within the kernel's LDS stack; worlds with fewer than two reference leaf nodes have no own tree at all (generic kernel)."""
    import ctypes as C
    import numpy as np
    from mort_amd import hip
    from tests.worlds import custom_bvh_world, BVH_WORLDS
    fn = hip.lib().mort_hip_debug_own_tree
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    rng = np.random.default_rng(11)
    worlds = dict(BVH_WORLDS)
    for n in (5, 9, 17, 33, 100, 257, 400):
        worlds[f"random{n}"] = [((float(rng.uniform(-8, 8)), float(rng.uniform(0, 2)), float(rng.uniform(-8, 8))), float(rng.uniform(0.05, 0.9)),
                                 ("lamb", (.5, .5, .5))) for _ in range(n)]
    for name, spheres in worlds.items():
        w = custom_bvh_world(spheres)
        out = (C.c_int * 9)()
        assert fn(C.cast(w.ptr, C.c_void_p), out) == 0, name
        n2, leaves, depth2, n4, stack4, reached, bad, slots, same = list(out)
        if leaves < 2:
            assert n2 == 0 and n4 == 0, name
            continue
        assert n2 == leaves - 1 and depth2 <= 15, (name, list(out))
        assert 1 <= n4 <= max(1, n2 // 2 + 1) and reached == leaves and bad == 0 and same == 1, (name, list(out))
        assert slots == n4 - 1 + leaves and 1 <= stack4 <= 24, (name, list(out))
