#!/usr/bin/env python3
"""Print VGPR/AGPR/scratch/occupancy per kernel of one HIP source (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
        if "error" in line:
            print(line)
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:90]:90s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('TotalSGPRs')} "
              f"scratch={v.get('ScratchSize')} occ={v.get('Occupancy')} lds={v.get('LDS Size')}")


if __name__ == "__main__":
    main()
