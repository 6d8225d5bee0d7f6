#!/usr/bin/env python3
"""Kernel resource usage (VGPR / scratch / occupancy) of one csrc/*.hip file, compactly."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from founddiff_amd import build as _build          # the library's own compiler and flags (HIPCC from the environment, include path from __file__)
HIPCC = _build.HIPCC
CFLAGS = [f for f in _build.FLAGS if not f.startswith("-Rpass") and f != "-fPIC"]
src = sys.argv[1]
if src == "--check-drain":          # (the ISA checks at the end of this file)
    src = None
r = subprocess.run([HIPCC] + CFLAGS + ["-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True) if src else subprocess.CompletedProcess([], 0, "", "")
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


# ---- ISA check of the inline-asm MFMA drains: `python tools/kres.py --check-drain file.hip [...]`
# Kernels that issue their MFMAs through inline asm protect the first VALU read of the accumulators with a block of s_nop
# (FD_MFMA_ASM_DRAIN, fd_common.h); the accumulators are tied to that block as asm operands, so the compiler cannot schedule
# a consumer above it.  This check reads the ISA anyway: between the last v_mfma in front of a drain block and the block itself
# no VALU instruction may read a register any of the preceding MFMAs writes.
def isa_lines(src):
    asm = "/tmp/kres_%s.s" % os.path.basename(src)
    r = subprocess.run([HIPCC] + CFLAGS + ["-S", "--cuda-device-only", src, "-o", asm], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    return [ln.strip() for ln in open(asm)]


def vregs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_drain(src):
    lines = isa_lines(src)

    def regs(tok):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
            if m.group(3) is not None:
                out.add(int(m.group(3)))
            else:
                out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        return out
    def dst_src(ins):
        ops = ins.split(None, 1)
        if len(ops) < 2:
            return ops[0], set(), set()
        parts = ops[1].split(",", 1)
        is_store = ops[0].startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "flat_store"))
        return ops[0], (set() if is_store else regs(parts[0])), regs(ops[1] if is_store else (parts[1] if len(parts) > 1 else ""))
    # (a label may carry a comment: ".LBB1_30:      ; in Loop: Header=BB1_32 Depth=1")
    is_label = lambda t: re.match(r"^[.\w$@]+:(\s*;.*)?$", t) is not None
    label_of = lambda t: t.split(":")[0]
    is_ins = lambda t: t and not t.startswith((";", ".")) and not is_label(t) and ":" not in t.split()[0]
    # per function (register numbers mean nothing across kernels): [start, end) line ranges and their MFMA destinations
    bounds = [k for k, t in enumerate(lines) if t.startswith(".type") and "@function" in t] + [len(lines)]
    fn_of = lambda k: max(b for b in bounds[:-1] if b <= k) if any(b <= k for b in bounds[:-1]) else 0
    fn_dst = {}
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        d = set()
        for t in lines[lo:hi]:
            if t.startswith("v_mfma"):
                d |= regs(t.split(",")[0])
        fn_dst[lo] = d

    def tail_ok(end, what):
        """the straight-line run of instructions in front of line `end`, back to the closest v_mfma or to the block's label:
        no VALU instruction in it may read an MFMA destination that the run itself has not redefined"""
        j = end - 1
        run = []
        while j >= 0 and not lines[j].startswith("v_mfma") and not is_label(lines[j]):
            if is_ins(lines[j]):
                run.append(lines[j])
            j -= 1
        redef, n = set(), 0
        mfma_dst = fn_dst.get(fn_of(end), set())
        if j >= 0 and is_label(lines[j]) and not lines[j].startswith(".L"):
            # the run is the function's ENTRY block (its label is the kernel symbol, e.g. the path on which a K loop with a
            # runtime trip count is skipped): no MFMA of this kernel has been issued on it, whatever its registers hold
            return 0, None
        for ins in reversed(run):
            op, d, sr = dst_src(ins)
            if op.startswith("v_") and (sr & mfma_dst) - redef:
                n += 1
                print(f"{src}: `{ins}` reads an MFMA result in front of {what}")
            redef |= d
        return n, (label_of(lines[j]) if j >= 0 and is_label(lines[j]) else None)

    blocks = bad = 0
    for i, ln in enumerate(lines):
        if not (ln.startswith("s_nop 15") and i > 0 and lines[i - 1].startswith(";;#ASMSTART")):
            continue
        blocks += 1
        n, label = tail_ok(i - 1, f"the drain block (line {i})")
        bad += n
        if label:                                   # the drain starts a basic block: also the tails of the blocks that branch to it
            for k, t in enumerate(lines):
                if t.startswith(("s_branch", "s_cbranch")) and t.split()[-1] == label:
                    bad += tail_ok(k, f"a branch to the drain block (line {k})")[0]
    print(f"{os.path.basename(src)}: {blocks} drain block(s), {bad} early read(s)")
    return blocks, bad


if len(sys.argv) > 2 and sys.argv[1] == "--check-drain":
    tot = [check_drain(f) for f in sys.argv[2:]]
    sys.exit(1 if any(b for _, b in tot) or not all(n for n, _ in tot) else 0)
