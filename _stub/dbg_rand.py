import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from oracle import pyoracle as po
from tests import synth
import merfin_amd as m
from tests.test_gpu_parity import build_index, oracle_hist, _trim
seed = int(sys.argv[1])
r = np.random.default_rng(1000 + seed)
k = int(r.integers(3, 32))
peak = float(r.choice([0.37, 1.0, 2.5, 9.0, 17.3, 26.0, 333.3, 1e6]))
os.environ["MFX_LOAD_FACTOR"] = str(r.choice([0.3, 0.5, 0.7, 0.9]))
os.environ["MFX_MZ_W"] = str(int(r.integers(1, 6)))
if r.random() < 0.2:
    os.environ["MFX_HOME_MODE"] = "plain"
sizes = tuple(int(x) for x in r.choice([0, 1, k - 1, k, k + 1, 37, 500, 4095, 4096, 4097, 9000, 20000], size=int(r.integers(1, 9))))
contigs, read, asm = synth.world(k=k, peak=max(peak, 1.0) if peak < 1e5 else 20.0, seed=2000 + seed, sizes=sizes, err_kmers=int(r.integers(0, 3000)) if k > 8 else 0)
rk, rv = read
rv = rv.astype(np.uint64)
if len(rv):
    big = r.random(len(rv)) < 0.02
    rv[big] = r.choice([1, 2, 1023, 1024, 1025, 65535, 10**6, 2**32 - 1], size=int(big.sum()))
rv = rv.astype(np.uint32)
read = (rk, rv)
probK = probP = None
if r.random() < 0.6:
    n = int(r.choice([1, 8, 184, 1500]))
    probK = r.integers(0, 12, size=n).astype(np.uint32)
    probK[r.random(n) < 0.3] = 0
    probP = np.round(r.random(n), 4)
lo, hi = 0, 2**64 - 1
if r.random() < 0.4:
    lo, hi = int(r.integers(0, 5)), int(r.choice([30, 1000, 2**31]))
print("seed", seed, "k", k, "peak", peak, "env", os.environ["MFX_LOAD_FACTOR"], os.environ["MFX_MZ_W"], os.environ.get("MFX_HOME_MODE"), "sizes", sizes, "prob", None if probK is None else len(probK), "minmax", lo, hi, flush=True)
t = time.time()
p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP, lo, hi)
print("oracle %.1fs kasm %d kmissing %d undrMax %d overMax %d" % (time.time() - t, g.kasm, g.kmissing, len(_trim(g.undr())), len(_trim(g.over()))), flush=True)
t = time.time()
ix = build_index(m, k, read, asm, lo, hi)
ev = m.Evaluator(ix, m.KParams(peak, probK, probP))
seqs = m.Sequences(contigs)
print("build %.1fs" % (time.time() - t), flush=True)
t = time.time()
res = ev.hist(seqs)
print("hist %.1fs kasm %d kmissing %d" % (time.time() - t, res.kasm, res.kmissing), flush=True)
t = time.time()
u, o = res.undr(), res.over()
print("arrays %.1fs undr %d over %d  sum %d %d  oracle sums %d %d" % (time.time() - t, len(_trim(u)), len(_trim(o)), int(u.sum()), int(o.sum()), int(g.undr().sum()), int(g.over().sum())), flush=True)
print("undrMax overMax", res.c.undrMax, res.c.overMax)
import torch
counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
ev.take_overflow()
t = time.time()
ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover)
torch.cuda.synchronize()
print("launch %.2fs" % (time.time() - t), flush=True)
h = counts.cpu().numpy().view(np.uint64)
nb = ev.nbins
print("image: undr sum", int(h[:nb].sum()), "over sum", int(h[nb:2 * nb].sum()), "kasm", int(h[2 * nb]), "kmis", int(h[2 * nb + 1]), "novf", int(h[2 * nb + 2]), "over[:5]", h[nb:nb + 5])
t = time.time()
rec = ev.take_overflow()
print("take_overflow %.2fs n=%d" % (time.time() - t, len(rec)), [hex(int(x)) for x in rec[:8]], flush=True)
t = time.time()
r2 = ev.result_from_counts(h.copy(), float(kover.item()), seqs.ncontigs)
print("from_counts %.2fs over sum %d" % (time.time() - t, int(r2.over().sum())), flush=True)
t = time.time()
r2.add_overflow(rec)
print("add_overflow %.2fs over sum %d overMax %d" % (time.time() - t, int(r2.over().sum()), r2.c.overMax), flush=True)
