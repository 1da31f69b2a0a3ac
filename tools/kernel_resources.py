#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels of one translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py mesh_to_sdf_amd/csrc/bvh.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c %s -o /tmp/_kr.o" % src
out = subprocess.run(cmd.split(), capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt in name and "m2s" in name:
        short = name.replace("m2s::(anonymous namespace)::", "").split("(")[0]
        print(f"{short[:70]:70s} VGPR {r.get('VGPRs', -1):3d} AGPR {r.get('AGPRs', 0):3d} SGPR {r.get('SGPRs', -1):3d} scratch {r.get('ScratchSize', 0):4d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', -1)}")
