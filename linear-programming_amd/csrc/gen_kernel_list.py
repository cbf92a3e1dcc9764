"""nm output of the library's objects (stdin) -> lpx_kernel_list.inc: one X(index, "mangled handle") line per launchable
kernel whose handle has external linkage (file-local kernels cannot lose their device code to another unit)."""
import re
import sys

types = {}
stubs = set()
for line in sys.stdin:
    parts = line.split()
    if len(parts) < 2:
        continue
    sym, typ = parts[-1], parts[-2]
    m = re.match(r"_Z(\d+)__device_stub__(.*)$", sym)
    if m:
        stubs.add(f"_Z{int(m.group(1)) - 15}{m.group(2)}")
    elif typ.isupper() and typ != "U":
        types[sym] = typ
for k, name in enumerate(sorted(s for s in stubs if s in types)):
    print(f'X({k}, "{name}")')
