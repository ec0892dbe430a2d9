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
        # Blocks and the loops they belong to, from the compiler's own annotations ("in Loop: Header=BBf_n Depth=d", "Parent Loop BBf_n
        # Depth=d", "This Loop Header: Depth=d").  The per-batch loop is the depth-2 loop with the most instructions; a block is inside it
        # if it is its header, names it as its header, or names it as a parent loop.
        isn = lambda l: l.startswith("\t") and not l.lstrip().startswith((";", "."))
        blocks, cur = [], None
        for n, l in enumerate(body):
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
                cur = {"label": l.split(":")[0].replace("; %", ""), "note": l, "lines": []}
                blocks.append(cur)
            elif cur is not None and l.lstrip().startswith(";") and not isn(l) and not cur["lines"]:
                cur["note"] += l
            elif cur is not None:
                cur["lines"].append(l)
        heads = [b_ for b_ in blocks if "Loop Header: Depth=2" in b_["note"]]
        if not heads:
            print("%s: no depth-2 loop found" % demangled)
            continue

        def members(h):
            name = h["label"].lstrip(".L")
            return [b_ for b_ in blocks if b_ is h or re.search(r"(Header=|Parent Loop )%s\b" % re.escape(name), b_["note"])]
        hd = max(heads, key=lambda h: sum(sum(map(isn, b_["lines"])) for b_ in members(h)))
        mem = members(hd)
        inside = [l for b_ in mem for l in b_["lines"]]
        outside = [l for b_ in blocks if b_ not in mem for l in b_["lines"]] + body[:body.index(blocks[0]["note"].split("\n")[0]) if blocks else 0]
        hdr, last = (0, hd["label"]), len(mem)

        def count(rng, pat):
            return sum(1 for l in rng if re.search(pat, l) and not l.lstrip().startswith(";"))
        print("%s" % demangled)
        print("   instructions %5d   per-batch loop %s (%d blocks): %5d" % (sum(map(isn, body)), hdr[1], last, sum(map(isn, inside))))
        for what, pat in (("scratch_load / scratch_store", r"\bscratch_"), ("v_readlane / v_writelane (SGPR spills)", r"\bv_(read|write)lane"),
                          ("global_load", r"\bglobal_load"), ("global_atomic", r"\bglobal_atomic"), ("ds_ (LDS)", r"\bds_")):
            print("   %-40s inside the batch loop %4d   outside %4d" % (what, count(inside, pat), count(outside, pat)))
        # the straight-line blocks of the loop that hold its global loads of 16 bytes (the probe): lane spills there are on the hot path
        hot = []
        for b_ in mem:
            ins = [l for l in b_["lines"] if isn(l)]
            x4 = sum("global_load_dwordx4" in l for l in ins)
            if x4 and len(ins) > 100:
                hot.append({"label": b_["label"], "n": len(ins), "x4": x4, "lanes": sum(bool(re.search(r"\bv_(read|write)lane", l)) for l in ins),
                            "scratch": sum("scratch_" in l for l in ins)})
        for b in hot:
            print("   block %-12s %4d instructions, %d x global_load_dwordx4 (the probe), lane spills %d, scratch %d" % (b["label"], b["n"], b["x4"], b["lanes"], b["scratch"]))


if __name__ == "__main__":
    main()
