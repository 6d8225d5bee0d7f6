#!/usr/bin/env python3
"""Kernel resource usage (VGPR / scratch / occupancy) of one csrc/*.hip file, compactly."""
import re, subprocess, sys
src = sys.argv[1]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17",
                    "-I/root/repo/include", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
cur = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
            if key == "lds":
                n = re.sub(r"\(anonymous namespace\)::|void ", "", cur.get("name", "?"))
                n = re.sub(r"\(.*", "", n)
                print(f"{n[:70]:70s} vgpr={cur.get('vgpr')}+a{cur.get('agpr')} scratch={cur.get('scratch')} occ={cur.get('occ')} lds={cur.get('lds')}")
if r.returncode:
    print(r.stderr[-2000:])
