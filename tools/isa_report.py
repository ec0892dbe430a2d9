#!/usr/bin/env python3
"""Where does the scratch traffic of the -hist kernel fall?  Compiles merfin_amd/csrc/mfx_kernels.hip to gfx950 assembly (device pass
only), takes every instance of mfx_hist_kernel apart at its loops -- the per-TILE loop (depth 1) and, inside it, the per-BATCH loop
(depth 2: extract, canonicalise, place, probe, K*, bin for BT k-mers per lane; the steady state) -- and counts the scratch, lane-spill
(v_readlane / v_writelane of spilled SGPRs), global and LDS instructions inside and outside the batch loop.
   python tools/isa_report.py [> profiles/rNN_isa_hot_loop.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "merfin_amd", "csrc", "mfx_kernels.hip")


def main():
    out = os.path.join(tempfile.mkdtemp(prefix="mfx_isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S", SRC, "-o", out]
    cmd += [a for a in sys.argv[1:] if a.startswith("-D")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    print("# %s" % " ".join(cmd[:-2]))
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z15mfx_hist_kernelI\S+):", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        body = lines[i:j + 1]
        i = j + 1
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        # the first depth-2 loop header inside the depth-1 tile loop
        hdr = None
        for n, l in enumerate(body):
            if re.match(r"^\.LBB\d+_\d+:", l) and n + 1 < len(body) and "This Loop Header: Depth=2" in body[n + 1]:
                hdr = (n, l.split(":")[0])
                break
        if hdr is None:
            for n, l in enumerate(body):
                if re.match(r"^\.LBB\d+_\d+:", l) and "Loop Header: Depth=2" in "".join(body[n:n + 3]):
                    hdr = (n, l.split(":")[0])
                    break
        if hdr is None:
            print("%s: no depth-2 loop found" % demangled)
            continue
        last = max(n for n, l in enumerate(body) if re.search(r"s_c?branch\S*\s+%s\b" % re.escape(hdr[1]), l))

        def count(rng, pat):
            return sum(1 for l in rng if re.search(pat, l) and not l.lstrip().startswith(";"))
        inside, outside = body[hdr[0]:last + 1], body[:hdr[0]] + body[last + 1:]
        isn = lambda l: l.startswith("\t") and not l.lstrip().startswith((";", "."))
        print("%s" % demangled)
        print("   instructions %5d   per-batch loop %s .. line %d: %5d" % (sum(map(isn, body)), hdr[1], last, sum(map(isn, inside))))
        for what, pat in (("scratch_load / scratch_store", r"\bscratch_"), ("v_readlane / v_writelane (SGPR spills)", r"\bv_(read|write)lane"),
                          ("global_load", r"\bglobal_load"), ("global_atomic", r"\bglobal_atomic"), ("ds_ (LDS)", r"\bds_")):
            print("   %-40s inside the batch loop %4d   outside %4d" % (what, count(inside, pat), count(outside, pat)))
        # the straight-line blocks of the loop that hold its global loads of 16 bytes (the probe): lane spills there are on the hot path
        blocks, cur = [], None
        for l in inside:
            if re.match(r"^\.LBB\d+_\d+:", l) or cur is None:
                cur = {"label": l.split(":")[0], "n": 0, "lanes": 0, "x4": 0, "scratch": 0}
                blocks.append(cur)
            if isn(l):
                cur["n"] += 1
                cur["lanes"] += bool(re.search(r"\bv_(read|write)lane", l))
                cur["x4"] += "global_load_dwordx4" in l
                cur["scratch"] += "scratch_" in l
        hot = [b for b in blocks if b["x4"] and b["n"] > 100]
        for b in hot:
            print("   block %-12s %4d instructions, %d x global_load_dwordx4 (the probe), lane spills %d, scratch %d" % (b["label"], b["n"], b["x4"], b["lanes"], b["scratch"]))


if __name__ == "__main__":
    main()
