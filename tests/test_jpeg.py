"""The product's own baseline JPEG decoder (mort_amd/csrc/host/mort_jpeg.c) against the reference's decode of its one
image asset: tests/golden/earthmap.jpg is imgs/earthmap.jpg (a data file of the reference), tests/golden/earthmap_rgb.npz
the bytes its vendored stb_image produces from it (oracle/_ref/stb_decode, tests/golden/make_golden.py).  JPEG decoding
is integer arithmetic: the match must be exact."""
import ctypes as C
import os

import numpy as np

from mort_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))


def _decode(path):
    L = host.lib()
    L.mort_read_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mort_read_image.restype = C.c_void_p
    w, h = C.c_int(0), C.c_int(0)
    p = L.mort_read_image(path.encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
    C.CDLL(None).free(C.c_void_p(p))
    return out


def test_earthmap_decodes_to_the_reference_texels():
    got = _decode(os.path.join(HERE, "golden", "earthmap.jpg"))
    want = np.load(os.path.join(HERE, "golden", "earthmap_rgb.npz"))["rgb"]
    assert got is not None and got.shape == want.shape == (512, 1024, 3)
    assert (got == want).all(), f"{int((got != want).sum())} bytes differ"


def test_rejects_what_it_does_not_decode(tmp_path):
    data = open(os.path.join(HERE, "golden", "earthmap.jpg"), "rb").read()
    bad = tmp_path / "prog.jpg"
    bad.write_bytes(data.replace(bytes([0xff, 0xc0]), bytes([0xff, 0xc2]), 1))  # SOF0 -> SOF2 (progressive)
    assert _decode(str(bad)) is None
    trunc = tmp_path / "trunc.jpg"
    trunc.write_bytes(data[:200])
    assert _decode(str(trunc)) is None
    assert _decode(str(tmp_path / "missing.jpg")) is None


def test_live_reference_decoder_agrees_when_built(tmp_path):
    """oracle/_ref/stb_decode is the reference's own vendored stb_image compiled from where it lies (oracle/Makefile `ref`; only where
    /root/reference exists).  When it is there, its output must be the committed fixture and the product decoder's output."""
    import subprocess
    import pytest
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "stb_decode")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    ppm = tmp_path / "ref.ppm"
    subprocess.check_call([exe, os.path.join(HERE, "golden", "earthmap.jpg"), str(ppm)])
    raw = ppm.read_bytes()
    assert raw.startswith(b"P6")
    header, n = [], 0
    pos = 0
    while len(header) < 4:  # magic, width, height, maxval
        end = pos
        while raw[end:end + 1] not in (b" ", b"\n", b"\t", b"\r"):
            end += 1
        header.append(raw[pos:end]); pos = end + 1
    w, h = int(header[1]), int(header[2])
    ref = np.frombuffer(raw[pos:pos + w * h * 3], dtype=np.uint8).reshape(h, w, 3)
    want = np.load(os.path.join(HERE, "golden", "earthmap_rgb.npz"))["rgb"]
    assert (ref == want).all()
    assert (_decode(os.path.join(HERE, "golden", "earthmap.jpg")) == ref).all()


def _stb(exe, path, tmp_path):
    import subprocess
    ppm = tmp_path / "stb.ppm"
    if ppm.exists():
        ppm.unlink()
    p = subprocess.run([exe, str(path), str(ppm)], capture_output=True)
    if p.returncode != 0 or not ppm.exists():
        return None
    raw = ppm.read_bytes()
    hdr = raw[:32].split()
    w, h = int(hdr[1]), int(hdr[2])
    off = raw.index(b"255") + 4
    return np.frombuffer(raw[off:off + w * h * 3], dtype=np.uint8).reshape(h, w, 3)


def test_damaged_files_decode_like_the_reference_decoder(tmp_path):
    """Truncated scans and flipped bytes: whatever the reference's stb_image makes of the file (it pads a short scan with zeros), the
    product decoder makes the same bytes of it, or both refuse it."""
    import pytest
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "stb_decode")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    data = open(os.path.join(HERE, "golden", "earthmap.jpg"), "rb").read()
    rng = np.random.default_rng(11)
    cases = [data[:n] for n in (600, 2000, 20000, len(data) - 1000, len(data) - 3)]
    for _ in range(60):  # damage inside the entropy-coded scan (headers stay valid)
        d = bytearray(data)
        for i in rng.integers(1000, len(d) - 2, size=int(rng.integers(1, 6))):
            d[int(i)] = int(rng.integers(0, 255))  # never 0xff: no new markers
        cases.append(bytes(d))
    compared = 0
    for k, blob in enumerate(cases):
        f = tmp_path / f"case{k}.jpg"
        f.write_bytes(blob)
        ours, ref = _decode(str(f)), _stb(exe, f, tmp_path)
        if ref is None:
            assert ours is None
            continue
        assert ours is not None and ours.shape == ref.shape
        assert (ours == ref).all(), f"case {k}: {int((ours != ref).sum())} bytes differ"
        compared += 1
    assert compared >= 5
