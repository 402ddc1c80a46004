"""dev: time mm_pcm_decode_f32 (GPU box): interleaved PCM -> planar float32, 256 clips x 10 s x 44.1 kHz stereo"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
n, ch = 256 * 441000, 2
for fmt, bps, name in ((2, 2, "s16"), (3, 3, "s24"), (5, 4, "f32")):
    raw = torch.randint(0, 255, (n * ch * bps,), dtype=torch.uint8, device=dev)
    out = torch.empty((ch, n), dtype=torch.float32, device=dev)
    for _ in range(2): lib.mm_pcm_decode_f32(raw.data_ptr(), fmt, ch, n, out.data_ptr(), n, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): lib.mm_pcm_decode_f32(raw.data_ptr(), fmt, ch, n, out.data_ptr(), n, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name} stereo, {n} frames: {dt*1e3:.3f} ms, {(raw.numel() + out.numel()*4)/dt/1e9:.0f} GB/s", flush=True)
