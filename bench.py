#!/usr/bin/env python3
"""bench.py -- MFCC + modulation-spectrum frames/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (MFCC, then the trajectory rFFT) over one batch of synthetic
clips that is already resident in HBM.  Per-GPU workload = BASELINE.json configs[2]: 1024 clips x
10 s x 16 kHz, win 400 / hop 160 / n_fft 512 / 40 mel / 13 MFCC + modulation spectrum; with N GPUs
every rank processes its own 1024 clips (weak scaling; N = 8 is configs[4], 8192 clips) and ONE RCCL
gather moves every rank's output slab to rank 0 inside the timed region.

Prints one JSON line (rank 0) with `value` = whole-job frames/s, plus
  roofline     -- the dominant kernel of the timed region (fused frame+window+rFFT+power+mel+log),
                  algorithmic bytes / average launch duration from HIP events on the launch stream;
  rfft_stage   -- the stage-isolated batched rFFT kernel (the "% HBM roofline (rFFT)" figure),
                  measured in the same process right after the timed region;
  cpu_baseline -- the NumPy oracle (a port of the reference's librosa path) on the host cores,
                  bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: (clips per GPU, seconds, cfg kwargs, with_modspec)
    "c3": (1024, 10.0, dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40,
                            n_mfcc=13, fmin=100.0, fmax=8000.0), True),
    "c2": (1024, 10.0, dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40,
                            n_mfcc=13, fmin=100.0, fmax=8000.0), False),
    "c4": (1024, 10.0, dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80,
                            n_mfcc=40, fmin=100.0, fmax=10000.0), False),
}


def pmc_traffic(stage):
    """HBM bytes per launch of a stage's kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc.csv, written by tools/summarize_prof.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this same command; FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes).  None if absent."""
    import csv
    import glob
    key = {"logmel": "logmel512", "dct": "dct_clamp", "modspec": "rfft16_kernel<2", "rfft": "rfft16_kernel<1, true>"}.get(stage)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.csv")))
    if not key or not files:
        return None, None
    for r in csv.DictReader(open(files[-1])):
        if key in r["Kernel"]:
            return int(r["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    return None, None


def synth_batch(torch, device, batch, n, sr, seed0):
    """0.3 sin(2 pi 220 t)(1 + 0.5 sin(2 pi 4 t)) + 0.05 N(0,1), generated on device."""
    import math
    g = torch.Generator(device=device).manual_seed(seed0)
    t = torch.arange(n, device=device, dtype=torch.float64) / sr
    base = (0.3 * torch.sin(2 * math.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * math.pi * 4 * t))).float()
    x = torch.randn((batch, n), generator=g, device=device, dtype=torch.float32)
    x.mul_(0.05).add_(base[None, :])
    return x


# ---- CPU baseline (oracle) ------------------------------------------------------------------
def _cpu_init():
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


def _cpu_one(args):
    clip, kw, with_mod = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mfcc_oracle as O
    m = O.mfcc(clip, O.OracleConfig(**kw))
    if with_mod:
        O.modspec(m)
    return m.shape[1]


def cpu_baseline(clips_np, kw, with_mod, budget_s=12.0):
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))   # a 1-GPU box's CPU share is 16 cores (the node shows 256)
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_cpu_init) as pool:
        pool.map(_cpu_one, [(clips_np[0], kw, with_mod)] * cores)           # warm-up / page-in
        t0 = time.perf_counter()
        frames = sum(pool.map(_cpu_one, [(clips_np[i % len(clips_np)], kw, with_mod) for i in range(cores)]))
        dt1 = time.perf_counter() - t0
        rounds = max(1, min(2000, int(budget_s / max(dt1, 1e-3))))
        n_clips = cores * rounds
        t0 = time.perf_counter()
        frames = sum(pool.map(_cpu_one, [(clips_np[i % len(clips_np)], kw, with_mod) for i in range(n_clips)],
                              chunksize=1))
        dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_clips} clips x {frames // n_clips} frames (same synthetic clips as the GPU "
                      f"run), NumPy oracle of the librosa path, {cores} worker processes x 1 thread, "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="override clips per GPU (debug)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic kernels")
    ap.add_argument("--gather", default="mfcc", choices=["mfcc", "full"],
                    help="N > 1: 'mfcc' gathers the MFCC slab and the root computes the modulation spectrum of the "
                         "gathered trajectories (default: half the bytes over xGMI); 'full': every rank computes its "
                         "own modulation spectrum and both arrays are gathered")
    a = ap.parse_args()

    # Anything libraries print on stdout (RCCL prints a version banner there at communicator init)
    # goes to stderr: stdout carries exactly ONE line, the JSON result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from modulation_mfcc_amd import MfccConfig, MfccPlan
    from modulation_mfcc_amd.dist import PipelinedGather, SlabLayout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MM_BENCH_FORCE_DIST=1: take the N>1 code path (process group, pipelined gather) even with a
    # single rank -- a rehearsal of that path on a 1-GPU box
    use_dist = world > 1 or bool(os.environ.get("MM_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    B, secs, kw, with_mod = WORKLOADS[a.workload]
    if a.clips:
        B = a.clips
    cfg = MfccConfig(**kw)
    n = int(secs * cfg.sr)
    T = cfg.num_frames(n)
    plan = MfccPlan(cfg)
    if a.generic:
        plan.force_generic(True)
    audio = synth_batch(torch, dev, B, n, cfg.sr, seed0=1000 * rank)

    # one flat output slab per rank so that a single gather per step moves everything; with N > 1
    # the slabs are double-buffered and the gather of step k runs on a side stream under the
    # kernels of step k+1 (every step's gather still completes inside the timed region)
    # N > 1, --gather mfcc: the modulation spectrum is a linear map of the MFCC trajectories, so only
    # the MFCC slab travels and the root runs ONE trajectory rFFT over the gathered block on the
    # gather's side stream (modulation_mfcc_amd/dist.py)
    mod_on_root = use_dist and with_mod and a.gather == "mfcc"
    lay = SlabLayout.make(cfg, B, n, with_mod and not mod_on_root)
    n_mod = cfg.mod_fft_len(T) if with_mod else 0
    pg = PipelinedGather(lay.numel, dev) if use_dist else None
    slab1 = torch.empty(lay.numel, dtype=torch.float32, device=dev) if not use_dist else None
    plan.workspace(B, n)
    mod_all = None
    if mod_on_root and rank == 0:
        mod_all = [torch.empty((world * B, cfg.n_mfcc, n_mod // 2 + 1), dtype=torch.complex64, device=dev)
                   for _ in range(pg.depth)]

    def root_modspec(i):        # runs inside the gather's side stream, root only
        got = pg.recv_block[i][:, :lay.mfcc_numel].reshape(world * B, cfg.n_mfcc, T)
        plan.modspec(got, out=mod_all[i])

    def step():
        slab = pg.acquire() if pg else slab1
        mfcc_out, mod_out = lay.views(slab)
        plan.mfcc(audio, out=mfcc_out)
        if with_mod and not mod_on_root:
            plan.modspec(mfcc_out, out=mod_out)
        if pg:
            pg.submit(post=root_modspec if mod_on_root else None)

    def drain():
        if pg:
            pg.finish()

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    # HIP events around the dominant kernel only inside the timed region (two hipEventRecord per timed
    # launch cost ~5 us of stream time: all four kernels timed = 4 % of a step); the other kernels'
    # durations come from a short untimed pass afterwards
    plan.timing_enable(True, stages=["logmel"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    plan.timing_enable(False)
    stage = plan.timing_read()
    plan.timing_enable(True)
    for _ in range(min(a.steps, 5)):
        step()
    drain()
    torch.cuda.synchronize()
    plan.timing_enable(False)
    for k, v in plan.timing_read().items():
        if k != "logmel":
            stage[k] = v
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames_total = world * B * T * a.steps
    res = {
        "metric": "MFCC+mod-spectrum frames/sec" if with_mod else "MFCC frames/sec",
        "value": frames_total / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{'2' if with_mod else '1'}] per GPU: {B} clips x "
                               f"{secs:g} s x {cfg.sr:g} Hz, win {cfg.win_length} hop {cfg.hop_length} "
                               f"n_fft {cfg.n_fft}, {cfg.n_mels} mel, {cfg.n_mfcc} MFCC"
                               + (f" + modulation spectrum (rFFT {n_mod} over trajectories)" if with_mod else ""),
                   "frames_per_clip": T, "clips_total": world * B, "kernel_path": plan.kernel_path,
                   "parallelism": f"clips sharded x{world}" + (", one RCCL gather per step (overlapped with the next step's kernels)" if world > 1 else "")
                                  + ("; MFCC slab gathered, modulation spectrum of the gathered trajectories computed on the root" if mod_on_root else ""),
                   "gather": (a.gather if use_dist else None)},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events recorded around every launch -----
        per_stage = {k: {"avg_ms": v[0] / v[1], "launches": v[1]} for k, v in stage.items()}
        res["kernels_ms"] = {k: round(v["avg_ms"], 4) for k, v in per_stage.items()}
        dom = max(per_stage, key=lambda k: per_stage[k]["avg_ms"])
        alg_bytes_per_frame = {
            "logmel": 4 * cfg.hop_length + 4 * cfg.n_mels,     # unique audio in + log-mel out
            "dct": 4 * cfg.n_mels + 4 * cfg.n_mfcc,            # log-mel in + MFCC out
            "modspec": (4 * T + 8 * (n_mod // 2 + 1)) * cfg.n_mfcc / T if with_mod else 0,
        }
        if dom in alg_bytes_per_frame:
            bytes_launch = alg_bytes_per_frame[dom] * B * T
            ach = bytes_launch / (per_stage[dom]["avg_ms"] * 1e-3) / 1e9
            traffic, src = pmc_traffic(dom) if a.workload == "c3" else (None, None)
            res["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": src, "algorithmic_bytes_per_launch": bytes_launch,
                               "algorithmic_bytes_per_frame": alg_bytes_per_frame[dom],
                               "avg_launch_ms": per_stage[dom]["avg_ms"]}

        # ---- stage-isolated batched rFFT (frames in -> complex bins out), same process ---------
        rows = B * T
        frames_buf = torch.randn((rows, cfg.n_fft), device=dev, dtype=torch.float32)
        spec = torch.empty((rows, cfg.n_bins), dtype=torch.complex64, device=dev)
        for _ in range(3):
            plan.rfft(frames_buf, cfg.n_fft, out=spec)
        torch.cuda.synchronize()
        plan.timing_enable(True)
        for _ in range(10):
            plan.rfft(frames_buf, cfg.n_fft, out=spec)
        plan.timing_enable(False)
        ms, cnt = plan.timing_read()["rfft"]
        bpf = 4 * cfg.n_fft + 8 * cfg.n_bins
        ach = rows * bpf / (ms / cnt * 1e-3) / 1e9
        res["rfft_stage"] = {"kernel": "batched rFFT-%d, %d rows" % (cfg.n_fft, rows), "bound": "hbm",
                             "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": bpf,
                             "frames_per_s": rows / (ms / cnt * 1e-3), "avg_launch_ms": ms / cnt}
        # practical HBM ceiling of this device: a plain device-to-device copy (bytes read + written)
        src_c = frames_buf.view(-1)[: (1 << 28)]              # 1 GiB
        dst_c = torch.empty_like(src_c)
        for _ in range(2):
            dst_c.copy_(src_c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst_c.copy_(src_c)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src_c.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        res["rfft_stage"]["device_copy_GBs"] = copy_gbs
        res["rfft_stage"]["frac_of_device_copy"] = ach / copy_gbs
        if "roofline" in res:
            res["roofline"]["device_copy_GBs"] = copy_gbs
        del frames_buf, spec, src_c, dst_c

        if world == 1 and not a.no_cpu:
            ns = min(B, 64)
            res["cpu_baseline"] = cpu_baseline(audio[:ns].cpu().numpy(), kw, with_mod)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)

    if use_dist:
        if pg is not None and rank == 0 and os.environ.get("MM_BENCH_FORCE_DIST"):
            # rehearsal check: what arrived at the root is what the last steps produced
            last = (pg.k - 1) % pg.depth
            assert torch.equal(pg.received[last][0], pg.slabs[last]), "gathered slab differs"
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
