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


def _run_world(world, n_clips, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_clips)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        assert f"rank {rank} ok" in r.stdout


@pytest.mark.parametrize("n_clips", [5, 2, 1])
def test_sharded_gather_world2(n_clips):
    _run_world(2, n_clips, 29600 + n_clips)


@pytest.mark.parametrize("n_clips", [1, 5])
def test_sharded_gather_world4(n_clips):
    """Four ranks: n_clips 1 leaves three ranks EMPTY (shard_bounds(1, 4)), n_clips 5 gives shards of 2 / 1 / 1 / 1 --
    the padded equal-size slabs, the per-rank counts and the root-side concatenation with empty and uneven ranks."""
    _run_world(4, n_clips, 29620 + n_clips)


def test_sharded_gather_world8():
    """Eight ranks -- the shape of the driver's N = 8 launch (`torch.distributed.run --nproc-per-node 8`), on gloo: 19 clips
    give shards of 3 / 3 / 3 / 2 / 2 / 2 / 2 / 2."""
    assert [b - a for a, b in shard_bounds(19, 8)] == [3, 3, 3, 2, 2, 2, 2, 2]
    _run_world(8, 19, 29650)
