"""bench.py's N-rank path on a one-GPU box: MORT_BENCH_REHEARSAL=1 puts every rank on GPU 0 and gathers through gloo (RCCL refuses ranks
that share a device), so what runs is the real launch contract -- torch.distributed.run, RANK / WORLD_SIZE from the environment,
set_partition per rank, the per-step gather, the max-over-ranks timing, rank 0's one JSON line -- and the composed frame is compared
with a single-rank render.  The RCCL transport itself is rehearsed by tests/test_cli.py::test_rccl_path_one_rank_rehearsal."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,extra", [(2, []), (3, ["--scene", "6", "--width", "200"])])
def test_bench_n_ranks_compose_the_single_rank_frame(nranks, extra):
    env = dict(os.environ, MORT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + nranks), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1",
           "--spp", "4", "--cpu-spp", "0"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["rehearsal"]["composed_frame_equals_single_rank"] is True
