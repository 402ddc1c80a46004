"""Clip-sharded multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL on
ROCm, over xGMI), ONE gather of the per-rank output slab to the root at the end.

Clips are independent (every reduction of the hot path -- the per-clip top_db max, the trajectory
rFFT -- stays inside a clip), so there is no mid-pipeline exchange: rank r computes
its contiguous block of clips and the only collective is the final gather.  MFCC and modulation
spectrum share one flat float32 slab per rank so that a single collective moves both.
"""
from __future__ import annotations

from dataclasses import dataclass

from .plan import MfccConfig


def shard_bounds(n_clips: int, world: int):
    """Contiguous [start, stop) per rank; the first n_clips % world ranks get one extra clip."""
    base, extra = divmod(int(n_clips), int(world))
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


@dataclass(frozen=True)
class SlabLayout:
    """Flat float32 slab of one rank: [mfcc B*n_mfcc*T | pad to even | modspec B*n_mfcc*(n/2+1)*2]."""
    batch: int
    n_mfcc: int
    n_frames: int
    n_mod: int          # 0 = no modulation spectrum

    @classmethod
    def make(cls, cfg: MfccConfig, batch: int, n_samples: int, with_modspec: bool = True):
        T = cfg.num_frames(n_samples)
        return cls(int(batch), cfg.n_mfcc, T, cfg.mod_fft_len(T) if with_modspec else 0)

    @property
    def mfcc_numel(self):
        return self.batch * self.n_mfcc * self.n_frames

    @property
    def mod_offset(self):          # complex64 view needs 8-byte alignment
        return self.mfcc_numel + (self.mfcc_numel & 1)

    @property
    def mod_numel(self):
        return self.batch * self.n_mfcc * (self.n_mod // 2 + 1) * 2 if self.n_mod else 0

    @property
    def numel(self):
        return self.mod_offset + self.mod_numel

    def views(self, slab):
        """(mfcc [B, n_mfcc, T] float32, modspec [B, n_mfcc, n/2+1] complex64 or None) into slab."""
        import torch
        m = slab[:self.mfcc_numel].view(self.batch, self.n_mfcc, self.n_frames)
        if not self.n_mod:
            return m, None
        c = slab[self.mod_offset:self.mod_offset + self.mod_numel]
        return m, torch.view_as_complex(c.view(self.batch, self.n_mfcc, self.n_mod // 2 + 1, 2))


def gather_slabs(slab, dst: int = 0, group=None, out=None):
    """The single collective of the path.  Every rank passes an equally sized slab; the root gets
    the list of all slabs (``out`` may provide pre-allocated receive buffers), others get None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = [torch.empty_like(slab) for _ in range(world)]
        dist.gather(slab, out, dst=dst, group=group)
        return out
    dist.gather(slab, None, dst=dst, group=group)
    return None


def mfcc_modspec_sharded(audio_all_or_local, cfg: MfccConfig, *, with_modspec=True, dst=0, group=None,
                         is_local=False, compute=None):
    """Run the hot path on this rank's clips and gather everything on ``dst``.

    audio_all_or_local  [B_total, n] (every rank holds or can index the full batch) or, with
                        is_local=True, this rank's own [B_local, n] block
    compute             callable(audio_local, layout, slab) filling ``slab``; defaults to the HIP
                        plan (tests inject a CPU stand-in to exercise the collective under gloo)
    Returns on dst: (mfcc [B_total, n_mfcc, T], modspec or None); elsewhere (None, None).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if is_local:
        local = audio_all_or_local
        counts_t = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_counts = [torch.zeros_like(counts_t) for _ in range(world)]
        dist.all_gather(all_counts, counts_t, group=group)
        counts = [int(c.item()) for c in all_counts]
    else:
        bounds = shard_bounds(audio_all_or_local.shape[0], world)
        counts = [e - s for s, e in bounds]
        s, e = bounds[rank]
        local = audio_all_or_local[s:e]
    n = local.shape[1]
    bmax = max(counts)
    lay = SlabLayout.make(cfg, bmax, n, with_modspec)          # equal-size slabs (padded shards)
    slab = torch.zeros(lay.numel, dtype=torch.float32, device=local.device)
    if local.shape[0]:
        lay_local = SlabLayout.make(cfg, local.shape[0], n, with_modspec)
        if compute is None:
            from .plan import get_plan
            plan = get_plan(cfg)
            m, ms = lay.views(slab)
            plan.mfcc(local, out=m[:local.shape[0]])
            if with_modspec:
                plan.modspec(m[:local.shape[0]], out=ms[:local.shape[0]])
        else:
            compute(local, lay_local, lay, slab)
    got = gather_slabs(slab, dst=dst, group=group)
    if got is None:
        return None, None
    ms_all, mod_all = [], []
    for r, g in enumerate(got):
        m, ms = lay.views(g)
        ms_all.append(m[:counts[r]])
        if with_modspec:
            mod_all.append(ms[:counts[r]])
    return torch.cat(ms_all, 0), (torch.cat(mod_all, 0) if with_modspec else None)
