"""Worker for tests/test_dist_gloo.py: world_size-2 gloo run of the sharded driver on CPU.

The HIP kernels cannot run here, so the per-rank compute is the oracle (allowed in tests); what is
under test is the sharding arithmetic, the slab layout and the single gather of dist.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mfcc_oracle as O  # noqa: E402
from modulation_mfcc_amd import MfccConfig  # noqa: E402
from modulation_mfcc_amd.dist import SlabLayout, mfcc_modspec_sharded, shard_bounds  # noqa: E402

KW = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)


def oracle_compute(local, lay_local, lay, slab):
    m, ms = lay.views(slab)
    cfg = O.OracleConfig(**KW)
    for i in range(local.shape[0]):
        mi = O.mfcc(local[i].numpy(), cfg)
        m[i] = torch.from_numpy(mi)
        if ms is not None:
            ms[i] = torch.from_numpy(O.modspec(mi))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_clips, n = int(sys.argv[1]), 4000
    clips = np.stack([O.synth_clip(s, n, 16000, "am") for s in range(n_clips)])
    audio = torch.from_numpy(clips)
    cfg = MfccConfig(**KW)
    assert shard_bounds(n_clips, world)[-1][1] == n_clips
    def root_modspec(mfcc_all):        # CPU stand-in of plan.modspec for the root-side variant
        return torch.from_numpy(np.stack([O.modspec(x) for x in mfcc_all.numpy()]))

    for is_local, on_root in ((False, False), (True, False), (False, True), (True, True)):
        s, e = shard_bounds(n_clips, world)[rank]
        arg = audio[s:e] if is_local else audio
        m, ms = mfcc_modspec_sharded(arg, cfg, with_modspec=True, dst=0, is_local=is_local,
                                     compute=oracle_compute, modspec_on_root=on_root,
                                     root_modspec=root_modspec)
        if rank == 0:
            ocfg = O.OracleConfig(**KW)
            want = np.stack([O.mfcc(c, ocfg) for c in clips])
            assert m.shape == want.shape, (m.shape, want.shape)
            np.testing.assert_array_equal(m.numpy(), want)
            want_ms = np.stack([O.modspec(w) for w in want])
            np.testing.assert_array_equal(ms.numpy(), want_ms)
        else:
            assert m is None and ms is None
    lay = SlabLayout.make(cfg, 3, n)
    assert lay.mod_offset % 2 == 0 and lay.numel == lay.mod_offset + lay.mod_numel
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
