"""CPU: the N>1 path (clip sharding + single gather) under gloo, world_size 2."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from modulation_mfcc_amd.dist import shard_bounds


def test_shard_bounds():
    assert shard_bounds(8192, 8) == [(i * 1024, (i + 1) * 1024) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    b = shard_bounds(1001, 7)
    assert b[0][0] == 0 and b[-1][1] == 1001 and all(x[1] == y[0] for x, y in zip(b, b[1:]))


@pytest.mark.parametrize("n_clips", [5, 2, 1])
def test_sharded_gather_world2(n_clips):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + n_clips),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_clips)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
