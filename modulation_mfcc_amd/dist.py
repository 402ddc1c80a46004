"""Clip-sharded multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL on
ROCm, over xGMI), ONE gather of the per-rank output slab to the root at the end.

Clips are independent (every reduction of the hot path -- the per-clip top_db max, the trajectory
rFFT -- stays inside a clip), so there is no mid-pipeline exchange: rank r computes
its contiguous block of clips and the only collective is the final gather.

Two variants of what travels.  The default (``modspec_on_root=False``) is the literal one: every rank computes the
MFCCs AND the modulation spectrum of its clips (one fused launch where the plan can) and both arrays share one flat
float32 slab per rank, so that a single collective moves both.  ``modspec_on_root=True`` exploits that the modulation
spectrum is a fixed linear map (zero-padded rFFT) of the MFCC trajectories, i.e. redundancy on the wire -- as many bytes
again as the MFCCs themselves over point-to-point xGMI links: only the MFCC slab is gathered and the root runs the
trajectory rFFT over the gathered [B_total, n_mfcc, T] block (one launch, ~0.3 ms for 8192 clips) -- half the bytes, at
the price of extra work on the root.  bench.py times both with equal steps and reports them side by side.
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


class PipelinedGather:
    """Double-buffered gather on a side stream: the gather of batch k overlaps the kernels of
    batch k+1 (xGMI fan-in into the root is slower than one batch of kernels at 8 GPUs).

        pg = PipelinedGather(layout_numel, device)
        for k in range(steps):
            slab = pg.acquire()          # compute stream waits until slab's previous gather is done
            ... launch kernels writing into slab on the current stream ...
            pg.submit()                  # gather it on the side stream
        pg.finish()                      # root: pg.received[i] = list of per-rank slabs of buffer i
    """

    def __init__(self, numel, device, dst=0, group=None, depth=2):
        import torch
        import torch.distributed as dist
        self.dst, self.group, self.depth = dst, group, depth
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.slabs = [torch.empty(numel, dtype=torch.float32, device=device) for _ in range(depth)]
        # root: one contiguous [world, numel] block per buffer (the per-rank receive tensors are its rows)
        self.recv_block = [torch.empty((self.world, numel), dtype=torch.float32, device=device)
                           if self.rank == dst else None for _ in range(depth)]
        self.received = [list(blk.unbind(0)) if blk is not None else None for blk in self.recv_block]
        self.stream = torch.cuda.Stream(device=device)
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]     # gather of buffer i finished
        self.ev_ready = [torch.cuda.Event() for _ in range(depth)]    # kernels of buffer i finished
        self.k = 0
        self._used = [False] * depth
        self._timed = []               # (start, end) event pairs on the side stream, see time_every()
        self._time_every = 0

    def time_every(self, every: int):
        """Record an event pair around every ``every``-th gather on the side stream (0 = off); gather_ms() reads them."""
        self._time_every = int(every)

    def gather_ms(self):
        """(average ms, count) of the timed gathers on this rank -- the collective as this rank's side stream saw it
        (a sender's time is its own send; the root's spans the receipt of every peer's slab).  Synchronises."""
        ms = [a.elapsed_time(b) for a, b in self._timed if (b.synchronize() or True)]
        self._timed = []
        return (sum(ms) / len(ms) if ms else 0.0), len(ms)

    def acquire(self):
        import torch
        i = self.k % self.depth
        if self._used[i]:
            torch.cuda.current_stream().wait_event(self.ev_done[i])
        return self.slabs[i]

    def submit(self, post=None):
        """Gather the current slab on the side stream; ``post(i)`` (root only) is called inside the side
        stream's context right after it, e.g. to run the root-side modulation spectrum on
        ``recv_block[i]``."""
        import torch
        i = self.k % self.depth
        self.ev_ready[i].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.ev_ready[i])
            timed = self._time_every and self.k % self._time_every == 0
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            gather_slabs(self.slabs[i], dst=self.dst, group=self.group, out=self.received[i])
            if timed:
                e1.record(self.stream)
                self._timed.append((e0, e1))
            if post is not None and self.rank == self.dst:
                post(i)
            self.ev_done[i].record(self.stream)
        self._used[i] = True
        self.k += 1

    def finish(self):
        import torch
        torch.cuda.current_stream().wait_stream(self.stream)


def mfcc_modspec_sharded(audio_all_or_local, cfg: MfccConfig, *, with_modspec=True, dst=0, group=None,
                         is_local=False, compute=None, modspec_on_root=False, root_modspec=None):
    """Run the hot path on this rank's clips and gather everything on ``dst``.

    audio_all_or_local  [B_total, n] (every rank holds or can index the full batch) or, with
                        is_local=True, this rank's own [B_local, n] block
    compute             callable(audio_local, layout, slab) filling ``slab``; defaults to the HIP
                        plan (tests inject a CPU stand-in to exercise the collective under gloo)
    modspec_on_root     False (default): every rank sends MFCC + modulation spectrum; True: gather the MFCCs only and
                        compute the modulation spectrum of the gathered block on ``dst`` (see the module docstring)
    root_modspec        callable(mfcc_all) -> modspec for the root-side variant (default: the HIP plan)
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
    on_root = bool(with_modspec and modspec_on_root)
    send_mod = with_modspec and not on_root
    lay = SlabLayout.make(cfg, bmax, n, send_mod)              # equal-size slabs (padded shards)
    slab = torch.zeros(lay.numel, dtype=torch.float32, device=local.device)
    if local.shape[0]:
        lay_local = SlabLayout.make(cfg, local.shape[0], n, send_mod)
        if compute is None:
            from .plan import get_plan
            plan = get_plan(cfg)
            m, ms = lay.views(slab)
            if send_mod:     # one launch where the plan can (fused tail)
                plan.mfcc_modspec(local, out=m[:local.shape[0]], out_mod=ms[:local.shape[0]])
            else:
                plan.mfcc(local, out=m[:local.shape[0]])
        else:
            compute(local, lay_local, lay, slab)
    got = gather_slabs(slab, dst=dst, group=group)
    if got is None:
        return None, None
    ms_all, mod_all = [], []
    for r, g in enumerate(got):
        m, ms = lay.views(g)
        ms_all.append(m[:counts[r]])
        if send_mod:
            mod_all.append(ms[:counts[r]])
    mfcc_all = torch.cat(ms_all, 0)
    if on_root:
        if root_modspec is None:
            from .plan import get_plan
            root_modspec = get_plan(cfg).modspec
        return mfcc_all, root_modspec(mfcc_all)
    return mfcc_all, (torch.cat(mod_all, 0) if with_modspec else None)
