"""dev: dump the ISA of one kernel (mangled-name substring) and list spills / MFMAs / barriers with the loop labels"""
import subprocess, sys, collections, os
sub = sys.argv[1]
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modulation_mfcc_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                "-x", "hip", "mm_kernels.hip", "-o", "/tmp/k.s"] + sys.argv[2:], cwd=src, check=True, stderr=subprocess.DEVNULL)
s = open("/tmp/k.s").read()
import re
names = re.findall(r"^(_Z\w*" + re.escape(sub) + r"\w*):", s, re.M)
name = names[0]
a = s.index(name + ":"); b = s.index(".Lfunc_end", a)
body = s[a:b].split("\n")
open("/tmp/kernel.s", "w").write("\n".join(body))
cnt = collections.Counter(l.strip().split(" ")[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", ".")))
print(name, len(body), "lines")
for k in ("scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx2", "scratch_load_dwordx4", "v_readlane_b32", "v_writelane_b32",
          "v_mfma_f32_16x16x4_f32", "s_barrier", "global_load_lds_dwordx4", "ds_read_b128", "ds_read_b64", "ds_read2_b64", "ds_write_b32", "ds_read_b32", "s_waitcnt"):
    if cnt[k]: print(f"  {k}: {cnt[k]}")
lab = ""
for i, l in enumerate(body):
    if l.startswith(".LBB"): lab = l
    t = l.strip()
    if t.startswith("scratch_") or t.startswith("s_barrier"): print(i, lab, t[:70])
