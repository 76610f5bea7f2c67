#!/usr/bin/env python3
"""Build guard: parse hipcc's -Rpass-analysis=kernel-resource-usage remarks (csrc/build.sh writes them to
<build dir>/<source>.remarks) and FAIL when a kernel that keeps asm-issued loads in C++ variables reports scratch or VGPR
spills: in the fp32-activation path of linear_tiled.hip a spilled `xr` register would be copied before its data landed.
usage: check_resources.py <build dir> [--table]"""
import glob
import os
import re
import subprocess
import sys

NO_SCRATCH = ("linear_tiled", "shared_attn_fwd_w128")     # mangled-name substrings: scratch / spills are a build error for these kernels


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True, timeout=10).stdout.strip() or name
    except Exception:
        return name


def parse(path):
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass-analysis", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    return kernels


def main():
    bdir = sys.argv[1]
    table = "--table" in sys.argv
    bad = []
    for f in sorted(glob.glob(os.path.join(bdir, "*.remarks"))):
        for k in parse(f):
            scratch = int(k.get("ScratchSize [bytes/lane]", "0") or 0)
            spill = int(k.get("VGPRs Spill", "0") or 0)
            if table:
                print("%-28s vgpr %4s agpr %3s spill %3d scratch %4d occ %s  %s" % (os.path.basename(f)[:-8], k.get("VGPRs", "?"), k.get("AGPRs", "?"), spill,
                                                                                 scratch, k.get("Occupancy [waves/SIMD]", "?"), demangle(k["name"])[:150]))
            if any(s in k["name"] for s in NO_SCRATCH) and (scratch or spill):
                bad.append((k["name"], spill, scratch))
    for name, spill, scratch in bad:
        print("check_resources: %s: %d VGPRs spilled, %d bytes of scratch per lane - this kernel holds asm-issued loads in C++ variables "
              "and must not spill (instantrestore_amd/csrc/linear_tiled.hip)" % (demangle(name), spill, scratch), file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
