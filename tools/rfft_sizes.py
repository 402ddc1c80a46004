import sys; sys.path.insert(0,'.')
import torch, time
from modulation_mfcc_amd import MfccConfig, MfccPlan
plan = MfccPlan(MfccConfig(sr=48000,n_fft=2048,win_length=1200,hop_length=480,n_mels=80,n_mfcc=40,fmin=100.,fmax=10000.))
for n in (512, 1024, 2048, 4096):
    rows = (1<<29)//(n*4)//2
    x = torch.randn((rows, n), device='cuda')
    out = torch.empty((rows, n//2+1), dtype=torch.complex64, device='cuda')
    for _ in range(3): plan.rfft(x, n, out=out)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): plan.rfft(x, n, out=out)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    by = rows*(4*n + 8*(n//2+1))
    print(n, rows, round(dt*1e3,3), 'ms', round(by/dt/1e9), 'GB/s')
