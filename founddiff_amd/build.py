"""Build founddiff_amd/lib/libfounddiff_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree)."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfounddiff_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the library (round 6).  With them the
# library was NOT run-to-run repeatable once two kernels of different kinds shared the chip -- the default two-stream sample():
# the element in the EVEN register of a packed result came out wrong in lanes 48..63 of a wave, sporadically (a level-0 scan
# output in ~1 of 10 concurrent forwards; up to 0.14 on a [0, 1] image over a 50-step loop; one stream: always bit-stable).  Found
# with tools/probes/determinism_matrix.py + race_hunt.py, isolated by this flag: every configuration repeatable with it, none
# without (profiles/r06/packed_fp32_nondeterminism.md).  Cost: 14.20 -> 14.07 slices/s alternated.  The f32x2 vector types in the
# sources stay; the compiler splits them into scalar instructions.  (-fno-slp-vectorize, round 2, was the first sighting of the
# same thing: SLP-packed v_pk_add_f32 feeding ds_bpermute gave run-to-run differences.)
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# -DFD_RELEASE: the development switches of csrc/fd_common.h (FD_DEV_SWITCHES) are compiled to their defaults -- the shipped library
# reads no environment variable.  FOUNDDIFF_DEV_BUILD=1 builds the form that reads them (A/B experiments, tools/probes/).
RELEASE = [] if os.environ.get("FOUNDDIFF_DEV_BUILD") == "1" else ["-DFD_RELEASE"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", *NO_PACKED_F32, *RELEASE,
         "-Rpass-analysis=kernel-resource-usage",       # -> lib/obj/<file>.resources.txt (resources() below)
         "-I" + os.path.join(HERE, "..", "include")]


# -fno-slp-vectorize is the default (packed-f32 + ds_bpermute run-to-run differences);
# files listed here contain no cross-lane traffic fed by packed math and profit from v_pk_*_f32.
SLP_OK = set()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# The binary16 build without v_fma_mix_f32.  With f16 values in sight hipcc selects v_fma_mix_f32 (VOP3P) for plain fp32 fmas too --
# 84 of the 1355 VALU instructions of pwdw_kernel<64>, one for one against v_fma_f32 / v_fmac_f32, every operand fp32 -- and the fused
# LN -> 1x1 -> depthwise kernels ran 8-13 % slower than their bfloat16 twins with the SAME instruction count (PMC: VALU busy 71 %
# against 79 %, 44 % of wave time parked against 38 %: the instruction issues at the v_fma_f32 rate, tools/probes/valu_rate.hip, but
# dependent instructions wait longer for it).  Without the feature the same fmas are v_fma_f32 again, bit for bit the same results,
# and half of that gap closes (launch 6 of the forward: 279-286 -> 257-265 us, bf16 239-245; profiles/r06/fp16_mode.md).
NO_FMA_MIX = ["-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]
# the second form of the library: the same sources with IEEE binary16 as the 16-bit storage / MFMA operand type (csrc/fd_common.h)
LIB_F16 = os.path.join(LIBDIR, "libfounddiff_hip_f16.so")


def build(force=False, verbose=False, half="bf16"):
    """half='bf16': lib/libfounddiff_hip.so; half='fp16': lib/libfounddiff_hip_f16.so (-DFD_HALF_F16, objects under lib/obj_f16)."""
    if half == "fp16":
        return _build(force, verbose, LIB_F16, "obj_f16", ["-DFD_HALF_F16", *NO_FMA_MIX])
    return _build(force, verbose, LIB, "obj", [])


def build_all(force=False, verbose=False):
    return build(force, verbose), build(force, verbose, half="fp16")


def _build(force, verbose, LIB, objname, extra):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, objname)
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or not os.path.exists(o[:-2] + ".resources.txt"):
            flags = [f for f in FLAGS if not (f == "-fno-slp-vectorize" and os.path.basename(s) in SLP_OK)] + extra
            jobs.append([HIPCC] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        err = r.stderr
        if "-c" in cmd:          # the per-kernel register / scratch / occupancy remarks of this object, kept beside it
            rem = [ln for ln in err.splitlines() if "remark:" in ln]
            err = "\n".join(ln for ln in err.splitlines() if "remark:" not in ln)
            with open(cmd[cmd.index("-o") + 1][:-2] + ".resources.txt", "w") as f:
                f.write("\n".join(rem) + "\n")
        if verbose and err.strip():
            print(err)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def resources(half="bf16"):
    """{source file: {mangled kernel name: {"vgpr", "agpr", "scratch", "occupancy", "lds"}}} of the objects build(half=...) compiled
    (hipcc's kernel-resource-usage remarks): tests/test_host_cpu.py holds the hot kernels to a scratch ledger, so that a
    register-allocation accident (round 5: an early return in fd_softplus_fast cost the fp32 scans 450-820 bytes of scratch and a
    factor 3-5) fails a test instead of waiting for a profile."""
    out = {}
    for f in sorted(glob.glob(os.path.join(LIBDIR, "obj_f16" if half == "fp16" else "obj", "*.resources.txt"))):
        cur, tab = None, {}
        for ln in open(f):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = tab.setdefault(m.group(1), {})
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, ln)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
        out[os.path.basename(f)[:-len(".resources.txt")]] = tab
    return out


if __name__ == "__main__":
    for lib in build_all(force="--force" in sys.argv, verbose="--quiet" not in sys.argv):
        print(lib)
