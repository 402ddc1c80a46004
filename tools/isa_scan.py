"""dev: the ISA of one kernel, compiled with the product flags -- spills / MFMAs / barriers with their loop labels, and
the instruction MIX of the whole kernel or of the regions between marker comments.

    python tools/isa_scan.py <mangled-name substring> [--tu mm_logmel16s.hip] [--mix] [extra hipcc flags]

--mix classifies every instruction (static count; loop bodies counted once -- weigh with the trip counts yourself):
  arith   v_fma / v_mul / v_add / v_sub / v_mac / v_pk_* / v_max / v_min / v_log / v_exp / v_rcp / v_sqrt ... on floats
  mfma    v_mfma_*
  move    v_mov, v_accvgpr_*, v_swap
  select  v_cndmask, v_cmp* (vector compares feeding selects / exec masks)
  xlane   DPP-modified ops (row_mirror, quad_perm, ...), v_readlane / v_readfirstlane / v_writelane, ds_bpermute / ds_swizzle, v_permlane
  addr    integer vector arithmetic (v_add_u32, v_lshl*, v_and, v_or, v_mad_u32 / i32, v_mul_lo ...): addresses and indices
  cvt     v_cvt_*
  lds     ds_read* / ds_write* (not the cross-lane forms)   vmem  global_* / buffer_* / scratch_*
  salu    s_* except waits / barriers / branches             wait  s_waitcnt, s_barrier, s_nop, s_sleep     branch s_cbranch / s_branch
A region starts at a line `; MMREG <name>` (emit with asm volatile("; MMREG name") in a side build) and runs to the next marker.
"""
import collections
import os
import re
import subprocess
import sys

args = sys.argv[1:]
sub = args.pop(0)
tu = "mm_unity.hip"
mix = False
while args and args[0] in ("--tu", "--mix"):
    if args[0] == "--tu":
        tu = args[1]; args = args[2:]
    else:
        mix = True; args = args[1:]
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modulation_mfcc_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                "-x", "hip", tu, "-o", "/tmp/k.s"] + args, cwd=src, check=True, stderr=subprocess.DEVNULL)
s = open("/tmp/k.s").read()
names = re.findall(r"^(_Z\w*" + re.escape(sub) + r"\w*):", s, re.M)
if not names:
    raise SystemExit(f"no kernel matching {sub!r} in {tu}")
name = names[0]
a = s.index(name + ":"); b = s.index(".Lfunc_end", a)
body = s[a:b].split("\n")
open("/tmp/kernel.s", "w").write("\n".join(body))


def klass(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")) or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")):
        return "xlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio", "s_sethalt")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if not op.startswith("v_"):
        return "other"
    if " row_" in line or "quad_perm" in line or "wave_sh" in line or "row_bcast" in line or "_dpp" in op:
        return "xlane"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "move"
    if op.startswith(("v_cndmask", "v_cmp")):
        return "select"
    if op.startswith("v_cvt"):
        return "cvt"
    if re.match(r"v_(pk_)?(fma|mul|add|sub|mac|max|min|log|exp|rcp|rsq|sqrt|sin|cos|fmac|fract|floor|ldexp|med3|max3|min3|mad)(_legacy)?_f(16|32|64)", op) or op.startswith(("v_fmac", "v_fmaak", "v_fmamk", "v_dot")):
        return "arith"
    return "addr"


cnt = collections.Counter(l.strip().split(" ")[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", ".")))
print(name, len(body), "lines")
for k in ("scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx2", "scratch_load_dwordx4", "v_readlane_b32", "v_writelane_b32",
          "v_mfma_f32_16x16x4_f32", "s_barrier", "global_load_lds_dwordx4", "ds_read_b128", "ds_read_b64", "ds_read2_b64", "ds_write_b32", "ds_read_b32", "s_waitcnt"):
    if cnt[k]:
        print(f"  {k}: {cnt[k]}")
lab = ""
for i, l in enumerate(body):
    if l.startswith(".LBB"):
        lab = l
    t = l.strip()
    if t.startswith("scratch_") or t.startswith("s_barrier"):
        print(i, lab, t[:70])
if mix:
    regions = collections.OrderedDict()
    cur = "(kernel)"
    for l in body:
        t = l.strip()
        m = re.match(r";\s*MMREG\s+(\S+)", t)
        if m:
            cur = m.group(1)
            continue
        if not l.startswith("\t") or t.startswith((";", ".")) or not t:
            continue
        op = t.split(" ")[0]
        regions.setdefault(cur, collections.Counter())[klass(op, t)] += 1
    cols = ["arith", "mfma", "move", "select", "xlane", "addr", "cvt", "lds", "vmem", "salu", "wait", "branch", "other"]
    print("\n| region | " + " | ".join(cols) + " | VALU total | arith share of VALU |")
    print("|---|" + "---|" * (len(cols) + 2))
    for r, c in regions.items():
        valu = sum(c[k] for k in ("arith", "mfma", "move", "select", "xlane", "addr", "cvt")) - 0
        print(f"| {r} | " + " | ".join(str(c[k]) for k in cols) + f" | {valu} | {c['arith'] / max(valu, 1):.2f} |")
