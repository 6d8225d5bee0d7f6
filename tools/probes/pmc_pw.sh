set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_pw; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o k --output-format csv -- python3 tools/kbench.py pw > $OUT/$c.out 2> $OUT/$c.err
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, m in c.items():
    if "conv_igemm" in k:
        f = m.get("FETCH_SIZE", [0]); w = m.get("WRITE_SIZE", [0])
        # per-shape: launches come in groups of 23 (3 warm + 20 timed) per shape
        n = 23
        for i in range(0, len(f), n):
            fi = sum(f[i:i+n])/len(f[i:i+n]); wi = sum(w[i:i+n])/max(len(w[i:i+n]),1)
            print(f"{k[:50]} shape#{i//n}: FETCH_SIZE {fi/1e3:.1f} MB(KB-units x1)  x2 = {2*fi/1e3:.1f} MB   WRITE_SIZE {wi/1e3:.1f} MB")
PY
