"""CPU model of the BVH megakernel's ordered walk.

scripts/bvh_order_experiment2.c renders a small frame with the oracle and, for every segment it traces, walks an SAH
tree over the reference's leaf nodes near-child-first with the kernel's pruning rule (mega_bvh.h own_prune, operation
for operation), checks the winner against the reference's box test of its leaf node, and compares with the reference's
own left-first walk.  Every ray that the model does not hand to the reference walk must agree with it bit for bit.
The GPU parity tests check the same property of the real kernel against whole images; this one checks it per ray.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("own_tree") / "bvh_model")
    srcs = [os.path.join(ROOT, "scripts", "bvh_order_experiment2.c")] + \
           [os.path.join(ROOT, "mort_amd", "csrc", "host", f) for f in ("mort_host.c", "mort_scenes.c")]
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "oracle"), *srcs, "-lm", "-lpthread", "-o", out], check=True)
    return out


@pytest.mark.parametrize("scene,width,spp", [(1, 240, 4), (10, 240, 4)])
def test_ordered_walk_agrees_with_reference_walk(model_binary, scene, width, spp):
    r = subprocess.run([model_binary, str(width), str(spp), "1", "0", str(scene)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"rays (\d+)", r.stdout)
    f = re.search(r"fallbacks (\d+) mismatches (\d+)", r.stdout)
    steps = re.search(r"ordered walk: node steps/ray ([0-9.]+)", r.stdout)
    ref = re.search(r"reference walk: box tests/ray ([0-9.]+)", r.stdout)
    assert m and f and steps and ref, r.stdout
    rays, fallbacks, mismatches = int(m.group(1)), int(f.group(1)), int(f.group(2))
    assert rays > 100000
    assert mismatches == 0
    assert fallbacks < rays // 1000          # the reference walk is the exception (about 1 ray in 10^5)
    assert 2 * float(steps.group(1)) < float(ref.group(1))  # and the ordered walk tests fewer boxes
