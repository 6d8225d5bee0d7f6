mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -x -q > gpurun_out/r4c/pytest.txt 2>&1; tail -4 gpurun_out/r4c/pytest.txt
for lib in base new; do
  if [ $lib = base ]; then export FOUNDDIFF_LIB=$PWD/founddiff_amd/lib/libfd_r4base.so; else unset FOUNDDIFF_LIB; fi
  python tools/stage_times.py --batches 8 --detail > gpurun_out/r4c/stages_$lib.md 2> gpurun_out/r4c/stages_$lib.err
  python tools/kbench.py conv3 > gpurun_out/r4c/conv3_$lib.txt 2>&1
  python tools/kbench.py pwdw > gpurun_out/r4c/pwdw_$lib.txt 2>&1
done
unset FOUNDDIFF_LIB
paste gpurun_out/r4c/conv3_base.txt gpurun_out/r4c/conv3_new.txt | awk -F'\t' '{print $1; print "   NEW: " $2}' | cut -c1-150
cat gpurun_out/r4c/pwdw_base.txt gpurun_out/r4c/pwdw_new.txt
python - <<'PY'
def rd(f):
    d={}
    for l in open(f):
        p=[x.strip() for x in l.strip().strip('|').split('|')]
        if len(p)==2:
            try: d[p[0]]=float(p[1].strip('*'))
            except: pass
    return d
a=rd('gpurun_out/r4c/stages_base.md'); b=rd('gpurun_out/r4c/stages_new.md')
for k in a:
    if k in b and (abs(a[k]-b[k])>0.0008 or 'total' in k): print(f"{k:14s} {a[k]:.4f} -> {b[k]:.4f}  {100*(b[k]/a[k]-1):+.1f}%")
PY
