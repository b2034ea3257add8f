"""Builds tests/emu/libemu_kernels.so: the device kernel headers compiled for the host against the
lockstep emulator (hip_emu.h + the host model of gfx950_prims.h).  TEST INFRASTRUCTURE ONLY."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "visualrwkv_amd", "csrc")
SO = os.path.join(HERE, "libemu_kernels.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emu(force=False):
    srcs = sorted(glob.glob(os.path.join(HERE, "emu_*.cpp")))
    EXP = os.path.join(ROOT, "benchmarks", "experiments")      # A/B partners kept lane-exact too (wkv7_bwd_v7.h, the J-wave tail)
    deps = srcs + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(EXP, "*.h"))
    if not force and os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    cmd = [cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-I", HERE, "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", EXP, *srcs, "-o", SO]
    subprocess.run(cmd, check=True)
    return SO


if __name__ == "__main__":
    print(build_emu(force=True))
