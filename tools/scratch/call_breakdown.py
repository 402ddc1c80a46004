"""dev: where one get_MFCCS_change(path, 10000, ...) call spends its wall time (stages separated by synchronize)"""
import os, sys, time, tempfile, wave
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
import bench
import mfcc_oracle as O
from modulation_mfcc_amd import get_MFCCS_change
from modulation_mfcc_amd.audio_io import load_audio
y = O.synth_clip(424242, 441000, 44100, "am")
pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype("<i2")
td = tempfile.mkdtemp(); path = os.path.join(td, "clip.wav")
with wave.open(path, "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
kw = bench.UI_CALL
def t(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("whole call (path)      %.3f ms" % t(lambda: get_MFCCS_change(path, 10000, **kw)))
print("load_audio(path)       %.3f ms" % t(lambda: load_audio(path, 10000)))
x = load_audio(path, 10000)[0]
print("call on device array   %.3f ms" % t(lambda: get_MFCCS_change(x, 10000, **kw)))
xn = x.cpu().numpy()
print("call on numpy array    %.3f ms" % t(lambda: get_MFCCS_change(xn, 10000, **kw)))
print("file read only         %.3f ms" % t(lambda: open(path, "rb").read()))
