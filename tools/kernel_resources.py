#!/usr/bin/env python3
"""VGPRs / occupancy / spills / LDS of every kernel of csrc/denoise.hip (or trace.hip), from hipcc's kernel-resource-usage remarks.
    python tools/kernel_resources.py [denoise.hip | trace.hip]      (build container: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys

f = sys.argv[1] if len(sys.argv) > 1 else "denoise.hip"
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ai_path_tracer_denoiser_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] + (["-ffp-contract=off"] if f == "trace.hip" else []) + \
      ["-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur, d = None, {}
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "Occupancy [waves/SIMD]": "occ", "SGPRs Spill": "sspill", "VGPRs Spill": "vspill",
        "LDS Size [bytes/block]": "lds"}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("void aipt::", "").replace("aipt::", "").split("(")[0]
        d = {}
    for k, short in keys.items():
        m = re.search(re.escape(k) + r": (\d+)", ln)
        if m and cur:
            d[short] = m.group(1)
    if "LDS Size" in ln and cur:
        print(f"{cur:62s} " + " ".join(f"{k} {d.get(k, '-'):>5s}" for k in keys.values()))
        cur = None
