#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
tools/run_forward.py --n 2 --batch 8).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB
and gfx950's FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section).
usage: traffic.py <FETCH_SIZE dir> <WRITE_SIZE dir> <out.json>   |   traffic.py --resummarize <traffic.json>  (the family
figures again from the file's own per_kernel table, e.g. after a kernel's template signature changed)"""
import collections, csv, glob, json, re, subprocess, sys
RESUM = sys.argv[1] == "--resummarize"
if RESUM:
    out = sys.argv[2]
    fetch_dir = write_dir = None
else:
    fetch_dir, write_dir, out = sys.argv[1:4]


def dem(n):
    if n.startswith("_Z"):
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    return re.sub(r"\(.*", "", n)


def load(d, ctr):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[dem(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


old = json.load(open(out)) if RESUM else None
fe, wr = ({}, {}) if RESUM else (load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE"))
per = dict(old["per_kernel"]) if RESUM else {}
for k in fe:
    if k not in wr or k.startswith("at::") or "elementwise" in k:
        continue
    f, w = sum(fe[k]) / len(fe[k]), sum(wr[k]) / len(wr[k])
    per[k] = {"n": len(fe[k]), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
per = dict(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["n"]))


def fam(prefix):
    ks = [k for k in per if k.startswith(prefix)]
    n = sum(per[k]["n"] for k in ks)
    return round(sum(per[k]["hbm_bytes_per_launch"] * per[k]["n"] for k in ks) / n) if n else None


def fam_re(pat):
    ks = [k for k in per if re.match(pat, k)]
    n = sum(per[k]["n"] for k in ks)
    return round(sum(per[k]["hbm_bytes_per_launch"] * per[k]["n"] for k in ks) / n) if n else None


import glob as _g, hashlib, os
_h = hashlib.sha256()
for _f in sorted(_g.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "founddiff_amd", "csrc", "*"))):
    _h.update(open(_f, "rb").read())
nfwd = int(os.environ.get("TRAFFIC_FORWARDS", "2"))          # tools/run_forward.py --n
res = {"note": __doc__.split("usage:")[0].strip(),
       "csrc_sha": old["csrc_sha"] if RESUM else _h.hexdigest()[:16], "batch": old["batch"] if RESUM else int(os.environ.get("TRAFFIC_BATCH", "8")),
       "total_hbm_bytes_per_forward": round(sum(v["hbm_bytes_per_launch"] * v["n"] for v in per.values()) / nfwd),
       "pwdw_gram_hbm_bytes_per_launch": fam("pwdw_gram_kernel"),
       "pwdw_hbm_bytes_per_launch": fam("pwdw_kernel"),
       "pwdw64_hbm_bytes_per_launch": fam("pwdw_kernel<64, false>"),
       "pwdw128_hbm_bytes_per_launch": fam("pwdw_kernel<128, false>"),
       "pwdw_proj_hbm_bytes_per_launch": fam("pwdw_kernel<64, true>"),
       "gemm_rows_zre_hbm_bytes_per_launch": fam("gemm_rows_zre_kernel"),
       "down_fused_hbm_bytes_per_launch": fam("down_fused_kernel"),
       "conv3x3_rw_hbm_bytes_per_launch": fam("conv3x3_rw_kernel"),
       "dwconv3x3_bf16_hbm_bytes_per_launch": fam("dwconv3x3_bf16_kernel"),
       "conv3x3_halo_hbm_bytes_per_launch": fam("conv3x3_halo_kernel"),
       # (template <BN, TH, F8, UP, SPL>)
       "conv3x3_halo128_hbm_bytes_per_launch": fam_re(r"conv3x3_halo_kernel<128, 8, false, false(, false)?>"),
       "conv3x3_halo64_hbm_bytes_per_launch": fam_re(r"conv3x3_halo_kernel<64, \d+, false, false(, false)?>"),
       "conv3x3_up_hbm_bytes_per_launch": fam_re(r"conv3x3_halo_kernel<\d+, \d+, false, true(, false)?>"),
       "per_kernel": per}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}))
