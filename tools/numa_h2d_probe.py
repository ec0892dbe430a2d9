"""H2D rate from pinned host memory allocated (and first touched) on each NUMA node of the host, and where the GPU hangs:
   python tools/numa_h2d_probe.py        (on the GPU box)"""
import glob
import os
import time

import torch


def cpus_of(node):
    s = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def main():
    for f in sorted(glob.glob("/sys/bus/pci/devices/*/vendor")):
        try:
            if open(f).read().strip() == "0x1002":
                d = os.path.dirname(f)
                cls = open(d + "/class").read().strip()
                if cls.startswith("0x03") or cls.startswith("0x12"):
                    print("AMD device %s class %s numa_node %s" % (os.path.basename(d), cls, open(d + "/numa_node").read().strip()))
        except OSError:
            pass
    nodes = sorted(int(os.path.basename(p)[4:]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
    all_cpus = sorted(os.sched_getaffinity(0))
    dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):
        for node in nodes:
            cp = [c for c in cpus_of(node) if c in all_cpus]
            if not cp:
                continue
            os.sched_setaffinity(0, cp)
            h = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
            h.fill_(7)
            rates = []
            for _ in range(5):
                torch.cuda.synchronize()
                t = time.perf_counter()
                dev.copy_(h, non_blocking=True)
                torch.cuda.synchronize()
                rates.append((1 << 30) / (time.perf_counter() - t) / 1e9)
            back = []
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                h.copy_(dev, non_blocking=True)
                torch.cuda.synchronize()
                back.append((1 << 30) / (time.perf_counter() - t) / 1e9)
            print("pinned memory of node %d (%d cpus): H2D %s GB/s   D2H %s GB/s" % (node, len(cp), " ".join("%.1f" % r for r in rates), " ".join("%.1f" % r for r in back)), flush=True)
            del h
            os.sched_setaffinity(0, all_cpus)


if __name__ == "__main__":
    main()
