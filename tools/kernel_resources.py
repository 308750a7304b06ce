#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py algebra_amd/csrc/msm_bls12_381_g1.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
                      "-Rpass-analysis=kernel-resource-usage", *sys.argv[3:], "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    if flt and flt not in name:
        continue
    short = re.sub(r"\(.*", "", name).replace("arkhip::", "").replace("void ", "")
    print("%-70s VGPR %4s  AGPR %3s  SGPR %3s  spilled VGPRs %3s  scratch %5s  LDS %6s  occ %s" % (
        short[:70], d.get("VGPRs"), d.get("AGPRs"), d.get("TotalSGPRs"), d.get("VGPRs Spill"), d.get("ScratchSize [bytes/lane]"),
        d.get("LDS Size [bytes/block]"), d.get("Occupancy [waves/SIMD]")))
