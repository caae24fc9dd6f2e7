"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Compiles the C restatement of the RNN-T loss (oracle/rnnt_loss_ref.c) into
oracle/_build/librnnt_oracle.so with plain gcc.  Called by __graft_entry__.build(), by the tests'
conftest and by bench.py's cpu_baseline leg.  Building the checker is not using it.

The reference is pure Python (no native sources, no setup.py), so there is nothing to
compile into oracle/_ref/: that directory stays empty by construction.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "librnnt_oracle.so")
SRC = os.path.join(HERE, "rnnt_loss_ref.c")


def build(force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
