"""ORACLE (test infrastructure): build oracle/_build/libfd_oracle.so from oracle/csrc with gcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libfd_oracle.so")
SRC = [os.path.join(HERE, "csrc", "scan_oracle.c")]


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(s) for s in SRC):
        return OUT
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", OUT] + SRC + ["-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
