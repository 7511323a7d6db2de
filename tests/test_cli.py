"""The `mort` binary -- the named drop-in surface (mort.cu:633-725: `mort <scene_id>`) -- run as a program.

CPU tests: usage / argument errors, `--mode host` (BASELINE config 1 as stated: Scene 1 200x112, 4 spp, host-side
serial loop) against the golden vectors and the oracle through the files it writes (PPM, fp32 dump, state dump) and
through tools/mort_diff.py.  GPU tests: the same files from the HIP path, `--gpus 2` with both ranks on the one GPU
(`--gather shm`; RCCL needs one GPU per rank and is exercised by bench.py on a multi-GPU node), wavefront mode on the
final scene, and the default earth texture."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mort_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MORT = os.path.join(ROOT, "mort_amd", "bin", "mort")
DIFF = os.path.join(ROOT, "tools", "mort_diff.py")
GOLD = np.load(os.path.join(HERE, "golden", "oracle_golden.npz"))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(MORT):
        subprocess.check_call(["make", "-C", ROOT, "host", "hip", "cli"])


def run(*args, cwd=ROOT, check=True):
    p = subprocess.run([MORT, *map(str, args)], cwd=cwd, capture_output=True, text=True, timeout=600)
    if check:
        assert p.returncode == 0, p.stdout + p.stderr
    return p


def read_ppm(path):
    data = open(path, "rb").read()
    hdr, rest = data.split(b"255\n", 1)
    w, h = int(hdr.split()[1]), int(hdr.split()[2])
    return np.frombuffer(rest, dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)[::-1]  # file rows are top-down, buffer rows bottom-up


def last_json(p):
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_usage_and_argument_errors():
    p = run(check=False)
    assert p.returncode != 0 and p.stdout.startswith("Usage: mort <number_between_1_and_10>")  # mort.cu:638-641
    assert run(1, "--bogus", check=False).returncode != 0
    assert run(1, "--mode", "nonsense", check=False).returncode != 0
    assert run(1, "--mode", "host", "--gpus", 2, check=False).returncode != 0
    p = run(3, "--mode", "host", "--earth", "/nonexistent/earth.jpg", "--width", 16, "--spp", 1, check=False)
    assert p.returncode != 0 and "Could not load image file" in p.stderr  # img_loader.h:33


def test_config1_host_mode_files(tmp_path, oracle):
    """BASELINE config 1: `mort 1 --width 200 --spp 4 --mode host` (one thread): image, accumulators and final streams."""
    ppm, raw, stf = tmp_path / "c1.ppm", tmp_path / "c1.raw", tmp_path / "c1.states"
    p = run(1, "--width", 200, "--spp", 4, "--mode", "host", "--out", ppm, "--dump-f32", raw, "--states-out", stf)
    j = last_json(p)
    assert j["mode"] == "host" and (j["width"], j["height"], j["spp_effective"]) == (200, 112, 4) and "Avg. time per frame" in p.stdout
    assert (read_ppm(ppm) == GOLD["s1_c1_rgba"][..., :3]).all()
    acc = np.fromfile(raw, dtype=np.float32).reshape(112, 200, 3)
    assert (acc.view(np.uint32) == GOLD["s1_c1_accum"].view(np.uint32)).all()
    world, cam = host.build_scene(1, width=200, spp=4)
    ref = oracle.render(world, cam, nthreads=4)
    st = np.fromfile(stf, dtype=oracle.STATE_DTYPE)
    assert (st["d"] == ref["states"]["d"]).all() and (st["v"] == ref["states"]["v"]).all() and j["segments"] == ref["segments"]
    # the comparison tool on the files: identical -> exit 0; against a different render -> exit 1
    other = tmp_path / "other.raw"
    run(1, "--width", 200, "--spp", 4, "--mode", "host", "--seed", 7, "--dump-f32", other, "--threads", 4)
    assert subprocess.run([sys.executable, DIFF, str(raw), str(raw), "--f32", "200", "112"], capture_output=True).returncode == 0
    assert subprocess.run([sys.executable, DIFF, str(raw), str(other), "--f32", "200", "112"], capture_output=True).returncode == 1


def test_host_mode_tree_threads_and_state_files(tmp_path, oracle):
    """Final scene through the host loop with the unified tree, two frames chained through a state file; the default
    earth texture (tests/golden/earthmap.jpg through the product's JPEG decoder) is found from another directory."""
    s1, ppm = tmp_path / "f1.states", tmp_path / "f2.ppm"
    run(9, "--width", 40, "--spp", 4, "--mode", "host", "--tree", "--threads", 4, "--states-out", s1, cwd=str(tmp_path))
    p = run(9, "--width", 40, "--spp", 4, "--mode", "host", "--tree", "--threads", 3, "--states-in", s1, "--out", ppm, cwd=str(tmp_path))
    assert "unified tree" in last_json(p)["kernel"]
    world, cam = host.build_scene(9, width=40, spp=4)
    r1 = oracle.render(world, cam, nthreads=8)
    r2 = oracle.render(world, cam, nthreads=8, states=r1["states"].copy())
    assert (read_ppm(ppm) == r2["rgba"][..., :3]).all()


@pytest.mark.gpu
def test_gpu_cli_files_match_oracle(tmp_path, oracle):
    ppm, raw, stf = tmp_path / "g.ppm", tmp_path / "g.raw", tmp_path / "g.states"
    p = run(1, "--width", 200, "--spp", 4, "--out", ppm, "--dump-f32", raw, "--states-out", stf)
    j = last_json(p)
    assert j["mode"] == "mega" and j["kernel"].startswith("mega_bvh_kernel")
    world, cam = host.build_scene(1, width=200, spp=4)
    ref = oracle.render(world, cam, nthreads=8)
    assert (read_ppm(ppm) == ref["rgba"][..., :3]).all()
    assert (np.fromfile(raw, dtype=np.float32).reshape(112, 200, 3).view(np.uint32) == ref["accum"].view(np.uint32)).all()
    st = np.fromfile(stf, dtype=oracle.STATE_DTYPE)
    assert (st["v"] == ref["states"]["v"]).all() and j["segments"] == ref["segments"]
    # host mode and GPU mode write identical files
    hraw = tmp_path / "h.raw"
    run(1, "--width", 200, "--spp", 4, "--mode", "host", "--threads", 8, "--dump-f32", hraw)
    assert subprocess.run([sys.executable, DIFF, str(raw), str(hraw), "--f32", "200", "112"], capture_output=True).returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene,extra", [(1, []), (8, ["--depth", "6"]), (6, ["--mode", "wave"]), (1, ["--mode", "throughput"]),
                                         (8, ["--depth", "6", "--mode", "throughput"])])
def test_gpu_cli_two_ranks_compose_the_single_gpu_frame(tmp_path, scene, extra):
    """`--gpus 2`: two forked ranks (both on device 0 here, rows gathered through the shared mapping) = one rank."""
    one, two = tmp_path / "one.ppm", tmp_path / "two.ppm"
    run(scene, "--width", 96, "--spp", 4, "--out", one, *extra)
    p = run(scene, "--width", 96, "--spp", 4, "--gpus", 2, "--devices", "0,0", "--gather", "shm", "--out", two, *extra)
    assert last_json(p)["gpus"] == 2
    assert (read_ppm(one) == read_ppm(two)).all()


@pytest.mark.gpu
def test_gpu_cli_final_scene_wavefront_and_default_earth(tmp_path, oracle):
    """BASELINE config 5's command shape (`mort 8 --mode wave`, here 64x64 x 4 spp) with the earth texture found by default."""
    ppm = tmp_path / "w.ppm"
    p = run(8, "--width", 64, "--spp", 4, "--mode", "wave", "--out", ppm, cwd=str(tmp_path))
    assert last_json(p)["kernel"].startswith("wf_trav_gen")
    world, cam = host.build_scene(8, width=64, spp=4)
    assert (read_ppm(ppm) == oracle.render(world, cam, nthreads=16)["rgba"][..., :3]).all()


@pytest.mark.gpu
def test_rccl_path_one_rank_rehearsal(gpu_ctx):
    """librccl loads, a communicator forms, a grouped ncclSend / ncclRecv of uchar rows on the context's stream returns
    the same bytes through the de-interleave kernel (one rank sending to itself: all a one-GPU box can host)."""
    import ctypes as C
    from mort_amd import hip
    L = hip.lib()
    L.mort_hip_comm_selftest.argtypes = [C.c_void_p]; L.mort_hip_comm_selftest.restype = C.c_int
    st = L.mort_hip_comm_selftest(gpu_ctx._h)
    assert st == 0, L.mort_hip_last_error(gpu_ctx._h).decode()


def test_scripted_input_frames(tmp_path, oracle):
    """--frames with --keys / --mouse: the reference's idle loop (mort.cu:49-120) -- input(), initialize(), render with the
    streams continuing -- against the oracle driven the same way, and mort_camera_input against its closed form."""
    import ctypes as C
    L = host.lib()
    L.mort_camera_input.argtypes = [C.POINTER(type(host.build_scene(10)[1])), C.c_int, C.c_int, C.c_int, C.c_int]
    L.mort_camera_input.restype = None
    world, cam = host.build_scene(10, width=64, spp=1)
    w = np.array(list(cam.w.e)); u = np.array(list(cam.u.e)); lf = np.array(list(cam.lookfrom.e)); la = np.array(list(cam.lookat.e))
    r1 = oracle.render(world, cam, nthreads=4)
    L.mort_camera_input(C.byref(cam), 1, 0, 0, 0)  # W: one unit along -w (mort.cu:52-55)
    assert np.allclose(list(cam.lookfrom.e), lf - w, atol=1e-6) and np.allclose(list(cam.lookat.e), la - w, atol=1e-6)
    r2 = oracle.render(world, cam, nthreads=4, states=r1["states"].copy())
    ppm = tmp_path / "f2.ppm"
    run(10, "--width", 64, "--spp", 1, "--mode", "host", "--frames", 2, "--keys", "W", "--out", ppm)
    assert (read_ppm(ppm) == r2["rgba"][..., :3]).all() and not (r1["rgba"] == r2["rgba"]).all()
    # a drag turns lookat about vup by -dx / 500 rad and keeps its distance (vec3.cuh:215-227)
    before = np.array(list(cam.lookat.e)) - np.array(list(cam.lookfrom.e))
    L.mort_camera_input(C.byref(cam), 0, 100, 0, 1)
    after = np.array(list(cam.lookat.e)) - np.array(list(cam.lookfrom.e))
    assert np.isclose(np.linalg.norm(after), np.linalg.norm(before), rtol=1e-5)
    up = np.array(list(cam.vup.e))
    b_perp = before - before.dot(up) / up.dot(up) * up; a_perp = after - after.dot(up) / up.dot(up) * up
    cosang = b_perp.dot(a_perp) / np.linalg.norm(b_perp) / np.linalg.norm(a_perp)
    assert np.isclose(np.arccos(np.clip(cosang, -1, 1)), 0.2, atol=1e-4) and np.isclose(before.dot(up), after.dot(up), atol=1e-4)
    # A / D move along -u / +u
    lf2 = np.array(list(cam.lookfrom.e)); u2 = np.array(list(cam.u.e))
    L.mort_camera_input(C.byref(cam), 8, 0, 0, 0)
    assert np.allclose(list(cam.lookfrom.e), lf2 + u2, atol=1e-5)


@pytest.mark.parametrize("env", [{"MORT_TEST_FAIL_RANK": "1", "MORT_TEST_BLOCK_RANK0": "1"}, {"MORT_TEST_FAIL_RANK": "2", "MORT_TEST_BLOCK_RANK0": "1"}, {"MORT_TEST_FAIL_RANK": "0"}])
def test_a_failed_rank_takes_the_others_down(env):
    """`--gpus N`: the frame gather is collective, so a rank that fails must not leave rank 0 waiting in ncclRecv (or a peer in ncclSend)
    for ever.  mort.c's hooks make one rank fail right after the fork while rank 0 / the healthy peers wait as they would inside the
    gather: the whole job exits non-zero at once and leaves no process behind.  No GPU needed."""
    import time
    t0 = time.time()
    p = subprocess.run([MORT, "1", "--width", "64", "--spp", "1", "--gpus", "3", "--gather", "shm"], cwd=ROOT, capture_output=True, text=True,
                       timeout=50, env={**os.environ, **env})
    assert p.returncode not in (0, 3), p.stdout + p.stderr
    assert time.time() - t0 < 20, "the job waited for a rank that could never arrive"
    assert "MORT_TEST_FAIL_RANK" in p.stderr
    time.sleep(0.3)
    left = subprocess.run(["ps", "-eo", "args"], capture_output=True, text=True).stdout
    assert not [l for l in left.splitlines() if l.startswith(MORT + " 1 --width 64 --spp 1 --gpus 3")], "a rank outlived the failed job"


@pytest.mark.gpu
def test_gpu_scripted_input_frames(tmp_path, oracle):
    """SURVEY f4 on the GPU: `mort 10 --frames 2 --keys W` -- input(), initialize(), render with the per-pixel streams continuing from
    frame 1 (mort.cu:49-120) -- against the oracle driven the same way."""
    import ctypes as C
    L = host.lib()
    world, cam = host.build_scene(10, width=96, spp=4)
    L.mort_camera_input.argtypes = [C.POINTER(type(cam)), C.c_int, C.c_int, C.c_int, C.c_int]
    L.mort_camera_input.restype = None
    r1 = oracle.render(world, cam, nthreads=8)
    L.mort_camera_input(C.byref(cam), 1, 0, 0, 0)
    r2 = oracle.render(world, cam, nthreads=8, states=r1["states"].copy())
    ppm, sts = tmp_path / "f2.ppm", tmp_path / "f2.states"
    p = run(10, "--width", 96, "--spp", 4, "--frames", 2, "--keys", "W", "--out", ppm, "--states-out", sts)
    assert last_json(p)["kernel"].startswith("mega_bvh_kernel")
    assert (read_ppm(ppm) == r2["rgba"][..., :3]).all() and not (r1["rgba"] == r2["rgba"]).all()
    st = np.fromfile(sts, dtype=oracle.STATE_DTYPE)
    assert (st["d"] == r2["states"]["d"].reshape(-1)).all() and (st["v"] == r2["states"]["v"].reshape(-1, 5)).all()
