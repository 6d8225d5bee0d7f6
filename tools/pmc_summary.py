#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch."""
import csv, glob, re, subprocess, sys, collections
def dem(n):
    if n.startswith('_Z'):
        n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    return re.sub(r'\(.*', '', n)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in agg.items():
    rows.append((dem(k), {c: (sum(v) / len(v), len(v)) for c, v in cs.items()}))
ctrs = sorted({c for _, cs in rows for c in cs})
print("kernel".ljust(52), *[c[:14].rjust(15) for c in ctrs], "n".rjust(5))
for k, cs in sorted(rows, key=lambda r: -max(v[0] * v[1] for v in r[1].values())):
    print(k[:52].ljust(52), *[f"{cs.get(c, (0, 0))[0]:15.4g}" for c in ctrs], f"{max(v[1] for v in cs.values()):5d}")
