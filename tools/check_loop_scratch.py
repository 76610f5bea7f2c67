#!/usr/bin/env python3
"""Build guard (round 4): no scratch traffic inside a loop that issues MFMAs.

A spilled register that is reloaded inside a main loop costs more than the reload: scratch loads count in vmcnt, hipcc
waits for them with s_waitcnt vmcnt(0), and that drains every store / LDS-DMA transfer the loop deliberately keeps in
flight.  The K = 320 X-stationary GEMM carried one such reload per chunk from round 2 to round 4 (bias variant; the plain
variant from the statistics tail on) - tools/check_resources.py could not see it: it reads totals, not where they land.
This tool compiles a source to ISA (-S), finds every loop (backward branch) per kernel and lists scratch_* instructions
inside loops that contain MFMAs.  Exit status 1 when a kernel matching --fail has any.
usage: check_loop_scratch.py <file.hip> [--fail REGEX] [--flags "..."] [-v]"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HERE = os.path.dirname(os.path.abspath(__file__))
INC = os.path.join(HERE, "..", "include")


def compile_to_isa(src, flags):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + INC, "-S", "--cuda-device-only", "-o", out] + flags + [src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise SystemExit("check_loop_scratch: compilation of %s failed" % src)
    text = open(out).read()
    os.unlink(out)
    return text.split("\n")


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S*: ", l) or re.match(r"^_Z\S*:\s*;", l)]
    starts.append((len(lines), None))
    for (s, name), (e, _) in zip(starts, starts[1:]):
        yield name, s, e


def loops_with_scratch(lines, s, e):
    """blocks by loop, from the loop annotations hipcc writes on block labels ("=>This Inner Loop Header", "in Loop: Header=BBn_m"):
    backward branches alone also catch cold blocks the layout placed above their predecessors"""
    loops, cur = {}, None
    for i in range(s, e):
        m = re.match(r"^\.(LBB\d+_\d+):\s*;(.*)$", lines[i])
        if m:
            name, note = m.group(1), m.group(2)
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
            if "Loop Header" in note:
                cur = name
            elif h:
                cur = "L" + h.group(1)
            else:
                cur = None
            continue
        if re.match(r"^\.LBB\d+_\d+:", lines[i]):
            cur = None
            continue
        if cur is not None:
            loops.setdefault(cur, []).append(lines[i])
    found = {}
    for h, body in loops.items():
        n_mfma = sum("v_mfma" in b for b in body)
        scr = [b.strip() for b in body if "scratch_" in b]
        if n_mfma and scr:
            found[(0, "." + h)] = (n_mfma, scr)
    return found


def main():
    args = sys.argv[1:]
    if not args:
        raise SystemExit(__doc__)
    src = args[0]
    fail_re = args[args.index("--fail") + 1] if "--fail" in args else None
    flags = args[args.index("--flags") + 1].split() if "--flags" in args else []
    verbose = "-v" in args
    lines = compile_to_isa(src, flags)
    bad = 0
    for name, s, e in kernels(lines):
        found = loops_with_scratch(lines, s, e)
        if not found:
            continue
        innermost = min(found.items(), key=lambda kv: kv[1][0])   # the loop with the fewest MFMAs that still has scratch traffic
        (line, label), (n_mfma, scr) = innermost
        failing = bool(fail_re and re.search(fail_re, name))
        bad += failing
        print("%s %s: loop %s (%d MFMAs) holds %d scratch instruction(s)" % ("FAIL" if failing else "note", name, label, n_mfma, len(scr)))
        if verbose or failing:
            for b in scr[:8]:
                print("      " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
