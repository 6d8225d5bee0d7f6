#!/usr/bin/env python3
"""Markdown table of per-kernel PMC figures from rocprofv3 counter_collection + kernel_trace CSVs
(one or more passes of tools/run_forward.py): duration, wave occupancy, VALU / MFMA busy, wave time parked,
LDS bank-conflict share.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_* count cycles (MI355X_MICROARCH.md)."""
import collections, csv, glob, re, subprocess, sys

SIMDS, CUS = 1024, 256


def dem(n):
    if n.startswith("_Z"):
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("__hip_bfloat16", "bf16")


ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[dem(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[dem(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for k, c in ctr.items():
    if k.startswith("at::") or not dur.get(k):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n = max(len(v) for v in c.values())
    us = sum(dur[k]) / len(dur[k]) / 1e3
    busy = m.get("SQ_BUSY_CYCLES", 0) / 32.0                      # summed over 32 shader engines
    if busy <= 0:
        continue
    simd_cycles = busy * SIMDS
    rows.append((us * n, k, n, us,
                 4 * m.get("SQ_WAVE_CYCLES", 0) / simd_cycles,
                 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles,
                 m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles,
                 m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                 m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                 m.get("SQ_LDS_IDX_ACTIVE", 0) / (busy * CUS)))
rows.sort(reverse=True)
print("| kernel | launches | avg us | waves/SIMD | VALU busy | MFMA busy | wave time parked | LDS active | LDS conflict share |")
print("|---|---|---|---|---|---|---|---|---|")
for _, k, n, us, occ, valu, mfma, wait, ldsc, ldsa in rows[:int(__import__("os").environ.get("PMC_ROWS", "24"))]:
    print(f"| `{k[:64]}` | {n} | {us:.1f} | {occ:.1f} | {valu:.0%} | {mfma:.0%} | {wait:.0%} | {ldsa:.0%} | {ldsc:.0%} |")
