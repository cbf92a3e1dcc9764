"""Build-time check (CPU container, no GPU): every kernel a translation unit can LAUNCH is in that unit's device code.

A launch goes through a host stub registered against the fat binary of the stub's own translation unit; a template
instance that the host side of a unit references but its device side never instantiated is found only at run time
("Cannot find Symbol ..." inside hipLaunchKernel, which aborts the process -- round 2, when the deferred-update pass was
split over two files).  For each object: host stubs (nm: __device_stub__<kernel>) must be a subset of the kernel
descriptors (<kernel>.kd) of the gfx950 code object bundled in the same file.

    python tools/check_kernels.py linear-programming_amd/csrc/*.o
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def host_stubs(obj):
    out = subprocess.check_output(["nm", obj], text=True)
    names = set()
    for line in out.splitlines():
        sym = line.split()[-1]
        m = re.match(r"_Z(\d+)__device_stub__(.*)$", sym)
        if m:
            n = int(m.group(1)) - len("__device_stub__")
            names.add(f"_Z{n}{m.group(2)}")
    return names


def device_kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.check_output([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, stderr=subprocess.STDOUT)
        kernels = set()
        for f in os.listdir(tmp):
            if "amdgcn" in f:
                syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "--symbols", "--wide", os.path.join(tmp, f)], text=True)
                for line in syms.splitlines():
                    parts = line.split()
                    if parts and parts[-1].endswith(".kd"):
                        kernels.add(parts[-1][:-3])
        return kernels


def main(objs):
    bad = 0
    total = 0
    for obj in objs:
        stubs, kernels = host_stubs(obj), device_kernels(obj)
        missing = sorted(stubs - kernels)
        total += len(stubs)
        for k in missing:
            print(f"{obj}: host stub without device code: {k}")
        bad += len(missing)
    print(f"check_kernels: {total} launchable kernels in {len(objs)} objects, {bad} without device code")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
