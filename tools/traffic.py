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


def scan_name(n):
    """scan kernels by their template arguments, from the mangled name (this image's c++filt does not know the bf16 / f16 type codes
    DF16b / DF16_ and leaves them mangled or garbles them): scan_chunk_kernel<T, N, R, FINAL, ODD, CPL>, scan_seq_kernel<T, N, R, ODD>"""
    m = re.search(r"(scan_chunk_kernel|scan_seq_kernel)I(DF16b|DF16_|f)((?:L[ib]\d+E)+)E", n)
    if not m:
        return None
    t = {"DF16b": "bf16", "DF16_": "f16", "f": "f32"}[m.group(2)]
    args = ", ".join(re.findall(r"L[ib](\d+)E", m.group(3)))
    return f"{m.group(1)}<{t}, {args}>"


def dem(n):
    if scan_name(n):
        return scan_name(n)
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


def scan_groups():
    """HBM bytes per API launch (fd_selective_scan[_xproj] = phase A + carry + phase C, or the single-pass kernel) of the three scan
    groups bench.py's roofline leg reports, keyed the way it looks them up.  The carry kernels are instantiated by segment count, not
    by level: the level-0 group gets two of the per-forward launches of the largest instantiation, levels 1-2 the rest."""
    nf = int(os.environ.get("TRAFFIC_FORWARDS", "2"))
    tot = {"l0": 0.0, "l12": 0.0, "seq": 0.0}
    cnt = {"l0": 2 * nf, "l12": 4 * nf, "seq": 0}
    carries = []
    for k, v in per.items():
        kk = scan_name(k) or k
        b = v["hbm_bytes_per_launch"] * v["n"]
        m = re.match(r"scan_chunk_kernel<\w+, (\d+), (\d+), (\d), (\d), (\d)>", kk)
        if m:
            tot["l0" if (int(m.group(1)) <= 4 and m.group(5) == "2") else "l12"] += b
        elif kk.startswith("scan_chunk_kernel<bool _Accum"):      # (c++filt's garbled form in a file written before scan_name existed)
            tot["l12"] += b
        elif kk.startswith("scan_seq_kernel"):
            tot["seq"] += b
            cnt["seq"] += v["n"]
        elif kk.startswith("scan_carry_kernel"):
            carries.append((v["hbm_bytes_per_launch"], v["n"]))
    carries.sort(reverse=True)
    if carries:
        big = carries[0][0]
        tot["l0"] += big * 2 * nf
        tot["l12"] += sum(b * n for b, n in carries) - big * 2 * nf
    return {"scan_seq_kernel (single pass": round(tot["seq"] / cnt["seq"]) if cnt["seq"] else None,
            "scan_chunk_kernel x2 + scan_carry_kernel|l0": round(tot["l0"] / cnt["l0"]) if tot["l0"] else None,
            "scan_chunk_kernel x2 + scan_carry_kernel|l12": round(tot["l12"] / cnt["l12"]) if tot["l12"] else None}


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
       "scan_hbm_bytes_per_launch": None,
       "per_kernel": per}
res["scan_hbm_bytes_per_launch"] = scan_groups()
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}))
