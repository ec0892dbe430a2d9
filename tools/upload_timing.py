import sys, time, numpy as np
sys.path.insert(0, ".")
import merfin_amd as m
r = np.random.default_rng(1)
seq = r.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1 << 30).tobytes()
for rep in range(3):
    t = time.perf_counter(); s = m.Sequences([seq]); print("upload %.3f s" % (time.perf_counter() - t), flush=True); del s
