"""Compiles one .hip file with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
name, VGPRs, AGPRs, scratch bytes per lane, occupancy.  usage: kernel_resources.py file.hip [name filter] [extra flags]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Iinclude", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:70]:70s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} scratch {v.get('ScratchSize')} occ {v.get('Occupancy')} LDS {v.get('LDS')}")
