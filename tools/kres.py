#!/usr/bin/env python3
"""Register / scratch usage per kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: tools/kres.py conv3x3_glds.hip [filter-regex] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "crimac_classifiers_unet_amd", "csrc", sys.argv[1])
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else None
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
       "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" ")[0] + ("_spill" if "Spill" in k else "")] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    if flt and not flt.search(name):
        continue
    print(f"{name[:90]:90s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize','?'):>5s} "
          f"vspill {r.get('VGPRs_spill','?'):>4s} sspill {r.get('SGPRs_spill','?'):>4s} occ {r.get('Occupancy','?')}")
