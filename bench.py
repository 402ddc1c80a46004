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
                  measured in the same process right after the timed region, beside a float4
                  grid-stride device copy (mm_devcopy_f32) as the practical HBM ceiling;
  c2, c4       -- BASELINE configs[1] (MFCC only) and configs[3] (48 kHz stereo, n_fft 2048, 80 mel,
                  40 MFCC, batch 512) timed in the same process after the headline region (N = 1);
  gather_full  -- N > 1: the literal north-star variant (every rank computes its modulation spectrum,
                  MFCC + modulation spectrum travel in one gather) timed beside the default;
  cpu_baseline -- the NumPy oracle (a port of the reference's librosa path) on the host cores,
                  bounded sample, rank 0 at N = 1 only.  It runs FIRST, before anything touches the
                  GPU (its worker pool is forked from a process that has not initialised HIP).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

C16K = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
C48K = dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0)
WORKLOADS = {
    # name: (BASELINE configs index, rows per GPU, channels, seconds, cfg kwargs, with_modspec)
    "c3": (2, 1024, 1, 10.0, C16K, True),
    "c2": (1, 1024, 1, 10.0, C16K, False),
    "c4": (3, 512, 2, 10.0, C48K, False),
}


def workload_label(name, B, T, cfg, n_mod):
    idx, _, ch, secs, _, with_mod = WORKLOADS[name]
    what = f"{B} clips" if ch == 1 else f"{B} stereo clips (both channels transformed = {B * ch} channel-rows [B, 2, n], row stride n)"
    return (f"BASELINE configs[{idx}] per GPU: {what} x {secs:g} s x {cfg.sr:g} Hz, win {cfg.win_length} "
            f"hop {cfg.hop_length} n_fft {cfg.n_fft}, {cfg.n_mels} mel, {cfg.n_mfcc} MFCC"
            + (f" + modulation spectrum (rFFT {n_mod} over trajectories)" if with_mod else ""))


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary that lists it
    (profiles/*_pmc.csv, written by tools/summarize_prof.py from separate FETCH_SIZE / WRITE_SIZE passes of
    this same command; FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes).  None if absent."""
    import csv
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.csv")), reverse=True):
        for r in csv.DictReader(open(f)):
            if kernel_key in r["Kernel"]:
                return int(r["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
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
    # one thread per worker process: the BLAS / OpenMP pools must be limited BEFORE numpy is first imported
    # in the worker (the parent has not imported it when it forks)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[v] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mfcc_oracle  # noqa: F401  (numpy / scipy load here, once per worker)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


_CLIPS = {}


def _cpu_one(args):
    seed, n, kw, with_mod = args
    import mfcc_oracle as O
    key = (seed % 4, n, kw["sr"])                       # four clips per worker, generated once (not in the timed work)
    if key not in _CLIPS:
        _CLIPS[key] = O.synth_clip(seed % 4, n, kw["sr"], "am")   # the GPU batch's signal model
    clip = _CLIPS[key]
    t0 = time.perf_counter()
    m = O.mfcc(clip, O.OracleConfig(**kw))
    if with_mod:
        O.modspec(m)
    return m.shape[1], time.perf_counter() - t0


def cpu_baseline(kw, n, with_mod, budget_s=12.0):
    """Runs before the first GPU call: fork()ing workers from a HIP-initialised parent is not safe."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))   # a 1-GPU box's CPU share is 16 cores (the node shows 256)
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_cpu_init) as pool:
        pool.map(_cpu_one, [(i, n, kw, with_mod) for i in range(4 * cores)], chunksize=4)   # warm-up: clips generated, pages in
        t0 = time.perf_counter()
        pool.map(_cpu_one, [(100 + i, n, kw, with_mod) for i in range(cores)])
        dt1 = time.perf_counter() - t0
        rounds = max(1, min(2000, int(budget_s / max(dt1, 1e-3))))
        n_clips = cores * rounds
        t0 = time.perf_counter()
        res = pool.map(_cpu_one, [(1000 + i, n, kw, with_mod) for i in range(n_clips)], chunksize=1)
        dt = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_clips} clips x {frames // n_clips} frames (the GPU run's signal model, four clips per worker generated "
                      f"beforehand), NumPy oracle of the librosa path, {cores} worker processes x "
                      f"1 thread, {dt:.1f} s wall, {busy:.0f} core-seconds in the path"}


EVENT_EVERY = 4      # dominant-kernel HIP events on every 4th timed step


def time_steps(torch, plan, fn, steps, warmup, stages, sync=None):
    """warmup untimed + `steps` timed calls of fn; HIP events around `stages` only inside the timed region --
    and only on every EVENT_EVERY-th step there (an event pair costs the stream two barrier packets, ~6 us each:
    rocprofv3 shows them as gaps on both sides of the kernel) -- the other stages from a short pass afterwards.
    Returns (seconds, {stage: (ms_sum, launches)})."""
    for _ in range(warmup):
        fn()
    if sync:
        sync()
    torch.cuda.synchronize()
    every = max(1, min(EVENT_EVERY, steps // 5))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % every == 0:
            plan.timing_enable(True, stages=stages)
        elif i % every == 1 or every == 1:
            plan.timing_enable(False)
        fn()
    if sync:
        sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    plan.timing_enable(False)
    stage = plan.timing_read()
    plan.timing_enable(True)
    for _ in range(min(steps, 5)):
        fn()
    if sync:
        sync()
    torch.cuda.synchronize()
    plan.timing_enable(False)
    for k, v in plan.timing_read().items():
        if k not in stages:
            stage[k] = v
    return dt, stage


def time_next_rows(torch, dev):
    """Device times of the rows around the hot path at BASELINE-like sizes (ms per call, wall clock over 5 calls after
    2 warm-up calls): N1 the MFCC-change tail of 1024 clips, N3 RMS and Hilbert envelopes, N4 PCM decode + 44.1 -> 16
    kHz resampling of 256 ten-second clips."""
    import ctypes as C
    from modulation_mfcc_amd import MfccConfig, MfccPlan, tail, calc, audio_io, _lib
    from modulation_mfcc_amd.batch import rms_batch

    def t(fn, k=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / k * 1e3, 4)

    out = {}
    plan = MfccPlan(MfccConfig(**WORKLOADS["c3"][4]))
    m = torch.randn((1024, 13, 1001), device=dev)
    sos1 = tail.design_lowpass(6, 12, 0.01)
    out["N1_change_tail_1024x13x1001_ms"] = t(lambda: plan.mfcc_change(m, sos1, sos1))
    x = torch.randn((256, 160000), device=dev)
    out["N3_rms_256x160000_ms"] = t(lambda: rms_batch(x, 400, 160, True))
    out["N3_hilbert_256x160000_ms"] = t(lambda: calc.hilbert_envelope_batch(x))
    x44 = torch.randn((256, 441000), device=dev)
    out["N4_resample_44100_to_16000_256x441000_ms"] = t(lambda: audio_io.resample_batch(x44, 44100, 16000))
    raw = torch.randint(0, 255, (256 * 441000 * 2 * 2,), dtype=torch.uint8, device=dev)
    pcm = torch.empty((2, 256 * 441000), dtype=torch.float32, device=dev)
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out["N4_pcm_decode_s16_stereo_256x441000_ms"] = t(lambda: lib.mm_pcm_decode_f32(raw.data_ptr(), 2, 2, 256 * 441000, pcm.data_ptr(),
                                                                                   256 * 441000, st))
    return out


def roofline_of(cfg, B, T, n_mod, with_mod, per_stage, fused_dct, traffic_key=None, fused_tail=False):
    alg = {
        # unique audio in + log-mel out (+ the unclamped MFCC rows where the kernel also applies the DCT, + the
        # modulation-spectrum rows where the whole tail runs in the launch)
        "logmel": 4 * cfg.hop_length + 4 * cfg.n_mels + (4 * cfg.n_mfcc if fused_dct else 0)
                  + (8 * (n_mod // 2 + 1) * cfg.n_mfcc / T if fused_tail else 0),
        "dct": 4 * cfg.n_mels + 4 * cfg.n_mfcc,            # log-mel in + MFCC out
        "modspec": (4 * T + 8 * (n_mod // 2 + 1)) * cfg.n_mfcc / T if with_mod else 0,
    }
    dom = max(per_stage, key=lambda k: per_stage[k]["avg_ms"])
    if dom not in alg:
        return None
    bytes_launch = alg[dom] * B * T
    ach = bytes_launch / (per_stage[dom]["avg_ms"] * 1e-3) / 1e9
    traffic, src = pmc_traffic(traffic_key) if traffic_key else (None, None)
    out = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_frame": alg[dom],
           "avg_launch_ms": per_stage[dom]["avg_ms"], "launches_timed": per_stage[dom]["launches"]}
    if dom == "logmel":
        # for information: the fused kernel is issue / LDS bound, not HBM bound -- its
        # algorithmic flops (rFFT 2.5 N log2 N + power + sparse mel + log + DCT) against the FP32 vector peak
        n = cfg.n_fft
        import math
        flops = 2.5 * n * math.log2(n) + 3 * (n // 2 + 1) + 4 * (n // 2 + 1) + cfg.n_mels + (2 * cfg.n_mels * cfg.n_mfcc if fused_dct else 0)
        tf = flops * B * T / (per_stage[dom]["avg_ms"] * 1e-3) / 1e12
        out["fp32_vector"] = {"algorithmic_flops_per_frame": flops, "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                              "frac": tf / 157.3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="override clips per GPU (debug)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c2 / c4 / rFFT-stage passes (profiling runs)")
    ap.add_argument("--generic", action="store_true", help="force the generic kernels")
    ap.add_argument("--no-fuse-tail", action="store_true", help="separate launches for the clamp fix-up and the trajectory rFFT (development A/B)")
    ap.add_argument("--variant", default=None, help="pin a fused-kernel variant (m12, w16s, w16, w8, wpf): development A/B")
    ap.add_argument("--gather", default="mfcc", choices=["mfcc", "full"],
                    help="N > 1: 'mfcc' gathers the MFCC slab and the root computes the modulation spectrum of the "
                         "gathered trajectories (default: half the bytes over xGMI); 'full': every rank computes its "
                         "own modulation spectrum and both arrays are gathered.  The other variant is timed too and "
                         "reported under 'gather_other'")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # before anything initialises HIP/HSA: the host driver only supports dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_dist = world > 1 or bool(os.environ.get("MM_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")

    # Anything libraries print on stdout (RCCL prints a version banner there at communicator init)
    # goes to stderr: stdout carries exactly ONE line, the JSON result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    idx, B, ch, secs, kw, with_mod = WORKLOADS[a.workload]
    if a.clips:
        B = a.clips
    cpu = None
    if world == 1 and rank == 0 and not a.no_cpu:
        cpu = cpu_baseline(kw, int(secs * kw["sr"]), with_mod)      # no torch / HIP yet in this process

    import torch
    import torch.distributed as dist
    from modulation_mfcc_amd import MfccConfig, MfccPlan, _lib
    from modulation_mfcc_amd.dist import PipelinedGather, SlabLayout

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MM_BENCH_FORCE_DIST=1: take the N>1 code path (process group, pipelined gather) even with a
    # single rank -- a rehearsal of that path on a 1-GPU box
    if use_dist:
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    def make(name, rows_override=0):
        _, b, c, s, k, wm = WORKLOADS[name]
        if rows_override:
            b = rows_override
        cfg = MfccConfig(**k)
        n = int(s * cfg.sr)
        plan = MfccPlan(cfg)
        if a.generic:
            plan.force_generic(True)
        if a.variant:
            plan.set_variant(a.variant)
        if a.no_fuse_tail:
            plan.set_fuse_tail(False)
        audio = synth_batch(torch, dev, b * c, n, cfg.sr, seed0=1000 * rank)
        if c > 1:                      # [B, ch, n]: the rows the kernels see are the channels, stride n
            audio = audio.view(b, c, n)
        return cfg, plan, audio, n, cfg.num_frames(n)

    cfg, plan, audio, n, T = make(a.workload, a.clips)
    rows = audio.reshape(-1, n)
    R = rows.shape[0]

    # one flat output slab per rank so that a single gather per step moves everything; with N > 1
    # the slabs are double-buffered and the gather of step k runs on a side stream under the
    # kernels of step k+1 (every step's gather still completes inside the timed region)
    # N > 1, --gather mfcc: the modulation spectrum is a linear map of the MFCC trajectories, so only
    # the MFCC slab travels and the root runs ONE trajectory rFFT over the gathered block on the
    # gather's side stream (modulation_mfcc_amd/dist.py)
    n_mod = cfg.mod_fft_len(T) if with_mod else 0
    plan.workspace(R, n)

    def run_variant(gather_mode, steps, warmup):
        mod_on_root = use_dist and with_mod and gather_mode == "mfcc"
        lay = SlabLayout.make(cfg, R, n, with_mod and not mod_on_root)
        pg = PipelinedGather(lay.numel, dev) if use_dist else None
        slab1 = torch.empty(lay.numel, dtype=torch.float32, device=dev) if not use_dist else None
        mod_all = None
        if mod_on_root and rank == 0:
            mod_all = [torch.empty((world * R, cfg.n_mfcc, n_mod // 2 + 1), dtype=torch.complex64, device=dev)
                       for _ in range(pg.depth)]

        def root_modspec(i):        # runs inside the gather's side stream, root only
            got = pg.recv_block[i][:, :lay.mfcc_numel].reshape(world * R, cfg.n_mfcc, T)
            plan.modspec(got, out=mod_all[i])

        def step():
            slab = pg.acquire() if pg else slab1
            mfcc_out, mod_out = lay.views(slab)
            if with_mod and not mod_on_root:
                plan.mfcc_modspec(rows, out=mfcc_out, out_mod=mod_out)    # one launch where the plan can (fused tail)
            else:
                plan.mfcc(rows, out=mfcc_out)
            if pg:
                pg.submit(post=root_modspec if mod_on_root else None)

        def drain():
            if pg:
                pg.finish()
            if use_dist:
                dist.barrier()

        dt, stage = time_steps(torch, plan, step, steps, warmup, ["logmel"], sync=drain)
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if pg is not None and rank == 0 and os.environ.get("MM_BENCH_FORCE_DIST"):
            # rehearsal check: what arrived at the root is what the last steps produced
            last = (pg.k - 1) % pg.depth
            assert torch.equal(pg.received[last][0], pg.slabs[last]), "gathered slab differs"
        return dt, stage, mod_on_root

    def run_extras():
        """The bench's other sections -- stage-isolated rFFT, device copy, BASELINE configs[1] and configs[3] -- run
        BEFORE the headline region: they are ~100 ms of GPU work, after which the clocks are at their steady state
        (with the driver's 5 warmup steps = 2.5 ms alone, the 10 ms timed region falls into the clock ramp and reads
        15 % slow: 0.50 vs 0.43 ms per step on the same box)."""
        ex = {}
        # ---- the rows either side of the hot path, for the record (N = 1 only); FIRST: measured
        # right before the headline region their double-precision kernels left it 10 % slower (0.49 vs 0.44 ms) ----
        if world == 1:
            try:
                ex["next_rows"] = time_next_rows(torch, dev)
            except Exception as e:          # never let a side measurement take the bench down
                ex["next_rows"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # ---- stage-isolated batched rFFT (frames in -> complex bins out), same process ---------
        nrows = R * T
        frames_buf = torch.randn((nrows, cfg.n_fft), device=dev, dtype=torch.float32)
        spec = torch.empty((nrows, cfg.n_bins), dtype=torch.complex64, device=dev)
        for _ in range(3):
            plan.rfft(frames_buf, cfg.n_fft, out=spec)
        torch.cuda.synchronize()
        plan.timing_enable(True)
        for _ in range(10):
            plan.rfft(frames_buf, cfg.n_fft, out=spec)
        plan.timing_enable(False)
        ms, cnt = plan.timing_read()["rfft"]
        bpf = 4 * cfg.n_fft + 8 * cfg.n_bins
        ach = nrows * bpf / (ms / cnt * 1e-3) / 1e9
        ex["rfft_stage"] = {"kernel": "batched rFFT-%d, %d rows" % (cfg.n_fft, nrows), "bound": "hbm",
                             "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": bpf,
                             "frames_per_s": nrows / (ms / cnt * 1e-3), "avg_launch_ms": ms / cnt}
        # practical HBM ceiling of this device: float4 grid-stride copy kernel of the library (bytes
        # read + written), on the same stream, timed with events
        src_c = frames_buf.view(-1)[: (1 << 28)]              # 1 GiB
        dst_c = torch.empty_like(src_c)
        lib = _lib.load()
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(2):
            _lib.check(lib.mm_devcopy_f32(src_c.data_ptr(), dst_c.data_ptr(), src_c.numel(), st), "mm_devcopy_f32")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mm_devcopy_f32(src_c.data_ptr(), dst_c.data_ptr(), src_c.numel(), st)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src_c.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        ex["rfft_stage"]["device_copy_GBs"] = copy_gbs
        ex["rfft_stage"]["frac_of_device_copy"] = ach / copy_gbs
        ex["_copy_gbs"] = copy_gbs
        del frames_buf, spec, src_c, dst_c

        # ---- the other single-GPU configurations, same process (N = 1 only) --------------------
        if world == 1:
            for name in ("c2", "c4"):
                if name == a.workload:
                    continue
                c2, p2, audio2, n2, T2 = make(name)
                rows2 = audio2.reshape(-1, n2)
                R2 = rows2.shape[0]
                out2 = torch.empty((R2, c2.n_mfcc, T2), dtype=torch.float32, device=dev)
                p2.workspace(R2, n2)
                k = max(5, a.steps // 2)
                dt2, st2 = time_steps(torch, p2, lambda: p2.mfcc(rows2, out=out2), k, 2, ["logmel"])
                ps = {kk: {"avg_ms": v[0] / v[1], "launches": v[1]} for kk, v in st2.items()}
                key2 = {"radix16-w16s": "logmel512s_kernel<1", "radix16-m12": "logmel12m", "radix16-wpf": "logmel_wpf"}.get(p2.kernel_path)
                ex[name] = {"workload": workload_label(name, WORKLOADS[name][1], T2, c2, 0),
                             "metric": "MFCC frames/sec", "value": R2 * T2 * k / dt2, "unit": "frames/s",
                             "ms_per_step": 1e3 * dt2 / k, "steps": k, "kernel_path": p2.kernel_path,
                             "kernels_ms": {kk: round(v["avg_ms"], 4) for kk, v in ps.items()},
                             "roofline": roofline_of(c2, R2, T2, 0, False, ps, p2.fused_dct, key2)}
                del out2, audio2, rows2
                torch.cuda.empty_cache()


        return ex

    extras = run_extras() if not a.no_extra else {}
    dt, stage, mod_on_root = run_variant(a.gather, a.steps, a.warmup)
    frames_total = world * R * T * a.steps
    res = {
        "metric": "MFCC+mod-spectrum frames/sec" if with_mod else "MFCC frames/sec",
        "value": frames_total / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_label(a.workload, B, T, cfg, n_mod),
                   "frames_per_clip": T, "clips_total": world * B, "kernel_path": plan.kernel_path,
                   "parallelism": f"clips sharded x{world}" + (", one RCCL gather per step (overlapped with the next step's kernels)" if world > 1 else "")
                                  + ("; MFCC slab gathered, modulation spectrum of the gathered trajectories computed on the root" if mod_on_root else ""),
                   "gather": (a.gather if use_dist else None),
                   "hsa_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
    }
    if use_dist and with_mod:
        other = "full" if a.gather == "mfcc" else "mfcc"
        dt2, _, _ = run_variant(other, max(3, a.steps // 2), 2)
        k2 = max(3, a.steps // 2)
        if rank == 0:
            res["gather_other"] = {"gather": other, "value": world * R * T * k2 / dt2, "unit": "frames/s",
                                   "ms_per_step": 1e3 * dt2 / k2, "steps": k2,
                                   "note": "'full' = the literal north-star split: every rank computes its own modulation "
                                           "spectrum and MFCC + modulation spectrum travel in the one gather"}

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events recorded around every launch -----
        # the n_fft 512 tile kernels also apply the (unclamped) DCT: their launch stores the MFCC rows too
        fused = plan.fused_dct
        per_stage = {k: {"avg_ms": v[0] / v[1], "launches": v[1]} for k, v in stage.items()}
        res["kernels_ms"] = {k: round(v["avg_ms"], 4) for k, v in per_stage.items()}
        key = {"radix16-w16s": "logmel512s_kernel<1", "radix16-m12": "logmel12m", "radix16-wpf": "logmel_wpf"}.get(plan.kernel_path)
        ftail = bool(with_mod and not mod_on_root and plan.fused_tail(R, n))
        if ftail:
            key = "logmel512s_kernel<2"         # the clip-mode instantiation
        res["config"]["launches_per_step"] = 1 if ftail else (len(per_stage) if per_stage else None)
        res["config"]["fused_tail"] = ftail
        rl = roofline_of(cfg, R, T, n_mod, with_mod, per_stage, fused, key, fused_tail=ftail)
        if rl:
            res["roofline"] = rl

        res["config"]["sections_before_headline"] = sorted(k_ for k_ in extras if not k_.startswith("_"))
        for k_, v_ in extras.items():
            if k_ != "_copy_gbs":
                res[k_] = v_
        if "_copy_gbs" in extras and "roofline" in res:
            res["roofline"]["device_copy_GBs"] = extras["_copy_gbs"]

        if cpu is not None:
            res["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
