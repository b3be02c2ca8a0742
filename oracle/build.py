"""ORACLE -- test infrastructure only.  Builds oracle/admm_port.c with gcc into
oracle/_build/ (git-ignored; travels to the GPU box with the snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libadmm_port.so")
SRC = os.path.join(HERE, "admm_port.c")


def build_oracle(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


if __name__ == "__main__":
    print(build_oracle(force=True))
