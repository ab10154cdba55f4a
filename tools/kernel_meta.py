#!/usr/bin/env python3
"""Per-kernel resource usage of the built device library, read from the gfx950 code object's own metadata
(llvm-readelf --notes on the unbundled .hip_fatbin): VGPRs, SGPRs, spills, scratch, LDS.

    python tools/kernel_meta.py [path/to/librtamd.so]      one line per kernel

tests/test_code_object.py asserts on it (the production kernels keep no scratch and spill no VGPR)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(REPO, "raytracer_amd", "lib", "librtamd.so")
FIELDS = ("private_segment_fixed_size", "group_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count")


def kernel_meta(lib=DEFAULT_LIB):
    """{demangled kernel name: {field: int}} for the gfx950 code object embedded in `lib`."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.elf")
        subprocess.run([f"{LLVM_BIN}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "unused.so")], check=True)
        subprocess.run([f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM_BIN}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    # one YAML map per kernel; the fields of a kernel follow its ".name:" up to the next one
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"^\s*\.name:\s*(\S+)", block, re.M)
        if not name:
            continue
        rec = {}
        for f in FIELDS:
            m = re.search(rf"^\s*\.{f}:\s*(\d+)", block, re.M)
            rec[f] = int(m.group(1)) if m else None
        out[name.group(1)] = rec
    names = list(out)
    dem = subprocess.run(["c++filt"] + names, check=True, capture_output=True, text=True).stdout.splitlines()
    return {d: out[n] for d, n in zip(dem, names)}


if __name__ == "__main__":
    for k, v in kernel_meta(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB).items():
        print(f"{k.split('(')[0]:72s} " + " ".join(f"{f.replace('_count', '').replace('_fixed_size', '')}={v[f]}" for f in FIELDS))
