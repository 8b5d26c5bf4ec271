#!/usr/bin/env python3
"""Registers, spills and occupancy of every kernel of the library, from hipcc's
-Rpass-analysis=kernel-resource-usage.  usage: kernel_resources.py [name substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "speech-recognition-hmm-continuous_amd")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
       "-I" + os.path.join(ROOT, "include"), "-munsafe-fp-atomics", "-ffp-contract=on", "-c",
       os.path.join(PKG, "csrc", "ghmm_hip.hip"), "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
want = sys.argv[1:]
pat = (r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)"
       r".*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)")
for m in re.finditer(pat, txt, re.S):
    n = m.group(1)
    try:
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.split("(")[0].strip()
    except OSError:
        pass
    n = n.replace("void ghmm::", "")
    if want and not any(w in n for w in want):
        continue
    print(f"{n:42s} sgpr {m.group(2):>3} vgpr {m.group(3):>3} agpr {m.group(4):>3} scratch {m.group(5):>4} "
          f"occ {m.group(6)} spill s{m.group(7)} v{m.group(8)}")
