#!/usr/bin/env python3
"""One line per kernel: name, VGPRs, AGPRs, scratch bytes/lane, occupancy, LDS bytes.
usage: tools/kernel_resources.py ldm_tf2_amd/csrc/gemm.hip ..."""
import re
import subprocess
import sys

for f in sys.argv[1:]:
  r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude",
                      "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True)
  cur = {}
  for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", line)
    if not m:
      continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
      cur = {}
    cur[k] = v
    if k.startswith("LDS Size"):
      name = subprocess.run(["c++filt", cur["Function Name"]],
                            capture_output=True, text=True).stdout.strip()
      name = name.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
      print("%-64s vgpr=%s agpr=%s scratch=%s occ=%s lds=%s" % (
          name, cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"),
          cur.get("Occupancy [waves/SIMD]"), v))
