import sys, time
sys.path.insert(0, "/root/repo")
import merfin_amd as m
for gb in (8, 96, 96, 96):
    t0 = time.time()
    r = m.gather_rate(int(gb * 2**30))
    print("table %d GiB: %.2f G lines/s (%.2f s)" % (gb, r / 1e9, time.time() - t0), flush=True)
time.sleep(20)
print("after 20 s idle: %.2f" % (m.gather_rate(96 * 2**30) / 1e9))
print("again: %.2f" % (m.gather_rate(96 * 2**30) / 1e9))
