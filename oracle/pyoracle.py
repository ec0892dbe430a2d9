"""ctypes binding of the CPU oracle (oracle/merfin_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke(); never from merfin_amd/.  Parity status of the
oracle itself: see merfin_oracle.h ("parity unpinned" at the meryl boundary).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmerfin_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("merfin_oracle.c", "merfin_oracle.h", "merfin_oracle_variants.cpp", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [("k", C.c_int), ("peak", C.c_double), ("n_prob", C.c_uint32),
                ("probK", C.POINTER(C.c_uint32)), ("probP", C.POINTER(C.c_double))]


class _Hist(C.Structure):
    _fields_ = [("kasm", C.c_uint64), ("kmissing", C.c_uint64), ("koverCpy", C.c_double),
                ("undrMax", C.c_uint32), ("overMax", C.c_uint32),
                ("undr", C.POINTER(C.c_uint64)), ("over", C.POINTER(C.c_uint64))]


class _KIter(C.Structure):
    _fields_ = [("k", C.c_int), ("bases", C.c_char_p), ("len", C.c_uint64), ("pos", C.c_uint64),
                ("fmer", C.c_uint64), ("rmer", C.c_uint64), ("mask", C.c_uint64), ("run", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u64p, u32p, f64p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L.orc_base_code.restype = C.c_int
    L.orc_base_code.argtypes = [C.c_ubyte]
    L.orc_revcomp.restype = C.c_uint64
    L.orc_revcomp.argtypes = [C.c_uint64, C.c_int]
    L.orc_canonical.restype = C.c_uint64
    L.orc_canonical.argtypes = [C.c_uint64, C.c_int]
    L.orc_encode.restype = C.c_uint64
    L.orc_encode.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_kiter_init.argtypes = [C.POINTER(_KIter), C.c_int, C.c_char_p, C.c_uint64]
    L.orc_kiter_next_base.restype = C.c_int
    L.orc_kiter_next_base.argtypes = [C.POINTER(_KIter)]
    L.orc_kiter_is_valid.restype = C.c_int
    L.orc_kiter_is_valid.argtypes = [C.POINTER(_KIter)]
    L.orc_kiter_position.restype = C.c_uint64
    L.orc_kiter_position.argtypes = [C.POINTER(_KIter)]
    L.orc_lookup_build.restype = C.c_void_p
    L.orc_lookup_build.argtypes = [C.c_int, u64p, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.orc_lookup_value.restype = C.c_uint32
    L.orc_lookup_value.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_lookup_size.restype = C.c_uint64
    L.orc_lookup_size.argtypes = [C.c_void_p]
    L.orc_lookup_free.argtypes = [C.c_void_p]
    L.orc_lookup_export.restype = C.c_uint64
    L.orc_lookup_export.argtypes = [C.c_void_p, u64p, u32p]
    L.orc_counter_new.restype = C.c_void_p
    L.orc_counter_new.argtypes = [C.c_int]
    L.orc_counter_add.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.orc_counter_finish.restype = C.c_uint64
    L.orc_counter_finish.argtypes = [C.c_void_p, C.POINTER(u64p), C.POINTER(u32p)]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_getK_values.argtypes = [C.POINTER(_Params), C.c_uint32, C.c_uint32, f64p, f64p, f64p]
    L.orc_getK_kmers.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, f64p, f64p, f64p]
    L.orc_getKmetric.restype = C.c_double
    L.orc_getKmetric.argtypes = [C.c_double, C.c_double]
    L.orc_histoQV.restype = C.c_double
    L.orc_histoQV.argtypes = [C.c_double, C.c_double, C.c_int]
    L.orc_load_kmetric.restype = C.c_int
    L.orc_load_kmetric.argtypes = [C.c_char_p, C.POINTER(u32p), C.POINTER(f64p)]
    L.orc_hist_init.argtypes = [C.POINTER(_Hist)]
    L.orc_hist_free.argtypes = [C.POINTER(_Hist)]
    L.orc_process_histogram.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(_Hist)]
    L.orc_output_histogram.restype = C.c_double
    L.orc_output_histogram.argtypes = [C.POINTER(_Params), C.POINTER(_Hist), C.POINTER(_Hist)]
    L.orc_report_histogram.argtypes = [C.POINTER(_Params), C.POINTER(_Hist), C.c_void_p, C.c_void_p]
    L.orc_hist_run.restype = C.c_double
    L.orc_hist_run.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), u64p, C.c_uint32,
                               C.c_int, C.c_int, C.c_uint64, C.POINTER(_Hist), u64p, u64p]
    L.orc_process_dump.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64,
                                   f64p, f64p, f64p, u64p, u64p]
    L.orc_output_dump.restype = C.c_uint64
    L.orc_output_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, f64p, f64p, f64p]
    L.orc_completeness_piece.argtypes = [C.POINTER(_Params), u64p, u32p, C.c_uint64, u64p, u32p, C.c_uint64, f64p, f64p]
    L.orc_variants_run.restype = C.c_long
    L.orc_variants_run.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p,
                                   C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
    L.orc_variants_run_cb.restype = C.c_long
    L.orc_variants_run_cb.argtypes = [C.POINTER(_Params), GETK_TEXT_FN, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p,
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
    _lib = L
    return L


GETK_TEXT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Params:
    """merfinGlobal's K* parameters: k, -peak, -prob table."""

    def __init__(self, k, peak, probK=None, probP=None):
        self.k = int(k)
        self.peak = float(peak)
        self.probK = np.ascontiguousarray(probK if probK is not None else [], dtype=np.uint32)
        self.probP = np.ascontiguousarray(probP if probP is not None else [], dtype=np.float64)
        assert len(self.probK) == len(self.probP)
        self.c = _Params(self.k, self.peak, len(self.probK), _u32(self.probK), _f64(self.probP))

    def ref(self):
        return C.byref(self.c)


def load_kmetric(path):
    K = C.POINTER(C.c_uint32)()
    P = C.POINTER(C.c_double)()
    n = lib().orc_load_kmetric(path.encode(), C.byref(K), C.byref(P))
    if n < 0:
        raise FileNotFoundError(path)
    k = np.array([K[i] for i in range(n)], dtype=np.uint32)
    p = np.array([P[i] for i in range(n)], dtype=np.float64)
    lib().orc_free(K)
    lib().orc_free(P)
    return k, p


class Lookup:
    """merylExactLookup restatement."""

    def __init__(self, k, kmers, values, minV=0, maxV=2**64 - 1, prefix_bits=0):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        values = np.ascontiguousarray(values, dtype=np.uint32)
        assert kmers.shape == values.shape
        self.k = k
        self.h = lib().orc_lookup_build(k, _u64(kmers), _u32(values), len(kmers), minV, maxV, prefix_bits)

    def value(self, kmer):
        return lib().orc_lookup_value(self.h, int(kmer))

    def __len__(self):
        return lib().orc_lookup_size(self.h)

    def export(self):
        n = len(self)
        k = np.zeros(n, dtype=np.uint64)
        v = np.zeros(n, dtype=np.uint32)
        lib().orc_lookup_export(self.h, _u64(k), _u32(v))
        return k, v

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_lookup_free(self.h)
            self.h = None


def count_kmers(k, seqs):
    """Canonical k-mer counts of a list of byte strings (`meryl count`)."""
    c = lib().orc_counter_new(k)
    for s in seqs:
        lib().orc_counter_add(c, s, len(s))
    K = C.POINTER(C.c_uint64)()
    V = C.POINTER(C.c_uint32)()
    n = lib().orc_counter_finish(c, C.byref(K), C.byref(V))
    kk = np.ctypeslib.as_array(K, shape=(max(n, 1),))[:n].copy()
    vv = np.ctypeslib.as_array(V, shape=(max(n, 1),))[:n].copy()
    lib().orc_free(K)
    lib().orc_free(V)
    return kk, vv


class Hist:
    def __init__(self):
        self.c = _Hist()
        lib().orc_hist_init(C.byref(self.c))

    @property
    def kasm(self):
        return self.c.kasm

    @property
    def kmissing(self):
        return self.c.kmissing

    @property
    def koverCpy(self):
        return self.c.koverCpy

    def undr(self):
        return np.array([self.c.undr[i] for i in range(self.c.undrMax)], dtype=np.uint64)

    def over(self):
        return np.array([self.c.over[i] for i in range(self.c.overMax)], dtype=np.uint64)

    def __del__(self):
        lib().orc_hist_free(C.byref(self.c))


def getK_values(p, readV, asmV):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().orc_getK_values(p.ref(), readV, asmV, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def getKmetric(readK, asmK):
    return lib().orc_getKmetric(readK, asmK)


def histoQV(kval, ktot, k):
    return lib().orc_histoQV(kval, ktot, k)


def process_histogram(p, R, A, bases):
    h = Hist()
    lib().orc_process_histogram(p.ref(), R.h, A.h, bases, len(bases), C.byref(h.c))
    return h


def hist_run(p, R, A, contigs, threads=1, mode=0, tile=0):
    """Whole-assembly -hist.  Returns (global Hist, per-contig kasm, kmissing, seconds)."""
    n = len(contigs)
    arr = (C.c_char_p * n)(*contigs)
    lens = np.array([len(c) for c in contigs], dtype=np.uint64)
    g = Hist()
    ka = np.zeros(n, dtype=np.uint64)
    km = np.zeros(n, dtype=np.uint64)
    t = lib().orc_hist_run(p.ref(), R.h, A.h, arr, _u64(lens), n, threads, mode, tile, C.byref(g.c), _u64(ka), _u64(km))
    return g, ka, km, t


def report_histogram(p, g, hist_path=None, summary_path=None):
    fh = _libc.fopen(hist_path.encode(), b"w") if hist_path else None
    fs = _libc.fopen(summary_path.encode(), b"w") if summary_path else None
    lib().orc_report_histogram(p.ref(), C.byref(g.c), fh, fs)
    if fh:
        _libc.fclose(fh)
    if fs:
        _libc.fclose(fs)


def process_dump(p, R, A, bases):
    n = len(bases)
    rk = np.zeros(n + 1)
    ak = np.zeros(n + 1)
    km = np.zeros(n + 1)
    kasm, kmiss = C.c_uint64(0), C.c_uint64(0)
    lib().orc_process_dump(p.ref(), R.h, A.h, bases, n, _f64(rk), _f64(ak), _f64(km), C.byref(kasm), C.byref(kmiss))
    return rk, ak, km, kasm.value, kmiss.value


def output_dump(path, name, rk, ak, km, append=False):
    f = _libc.fopen(path.encode(), b"a" if append else b"w")
    n = lib().orc_output_dump(f, name.encode(), len(rk) - 1, _f64(rk), _f64(ak), _f64(km))
    _libc.fclose(f)
    return n


def completeness_piece(p, rk, rv, ak, av):
    rk = np.ascontiguousarray(rk, dtype=np.uint64)
    rv = np.ascontiguousarray(rv, dtype=np.uint32)
    ak = np.ascontiguousarray(ak, dtype=np.uint64)
    av = np.ascontiguousarray(av, dtype=np.uint32)
    t, u = C.c_double(), C.c_double()
    lib().orc_completeness_piece(p.ref(), _u64(rk), _u32(rv), len(rk), _u64(ak), _u32(av), len(ak), C.byref(t), C.byref(u))
    return t.value, u.value


def kiter(k, bases):
    """Yield (position, fmer, rmer) for every valid k-mer, as kmerIterator does."""
    it = _KIter()
    L = lib()
    L.orc_kiter_init(C.byref(it), k, bases, len(bases))
    while L.orc_kiter_next_base(C.byref(it)):
        if L.orc_kiter_is_valid(C.byref(it)):
            yield L.orc_kiter_position(C.byref(it)), it.fmer, it.rmer


VARIANT_MODES = {"filter": 4, "polish": 5, "better": 6, "strict": 7, "loose": 8}


def variants_run(p, R, A, mode, vcf_path, names, contigs, out_path, comb=15, nosplit=False, debug_path=None, log_path=None):
    """processVariants + outputVariants (merfin-variants.C) over all contigs, input order."""
    n = len(contigs)
    nm = (C.c_char_p * n)(*[x.encode() for x in names])
    arr = (C.c_char_p * n)(*contigs)
    lens = np.array([len(c) for c in contigs], dtype=np.uint64)
    rc = lib().orc_variants_run(p.ref(), R.h, A.h, VARIANT_MODES[mode], comb, 1 if nosplit else 0, vcf_path.encode(),
                                nm, arr, _u64(lens), n, out_path.encode(),
                                debug_path.encode() if debug_path else None, log_path.encode() if log_path else None)
    if rc < 0:
        raise RuntimeError("orc_variants_run failed: %d" % rc)
    return rc


def variants_run_text(k, getk, mode, vcf_path, names, contigs, out_path, comb=15, nosplit=False, debug_path=None, log_path=None):
    """the same with the lookups supplied as a Python function getk(kmer_text) -> (readK, asmK, prob) of the k-mer's text
    (k bases, ACGT): k-agnostic -- the oracle of the variant modes at 32 <= k <= 64 (oracle/plain.py holds the k-mers as
    Python integers).  Everything above the lookups is the C++ restatement unchanged."""
    n = len(contigs)
    nm = (C.c_char_p * n)(*[x.encode() for x in names])
    arr = (C.c_char_p * n)(*contigs)
    lens = np.array([len(c) for c in contigs], dtype=np.uint64)

    def cb(_ctx, text, kk, rk, ak, pr):
        a, b, c = getk(C.string_at(text, kk).decode())
        rk[0], ak[0], pr[0] = a, b, c

    fn = GETK_TEXT_FN(cb)
    p = Params(k, 1.0)                         # only p.k matters on this path
    rc = lib().orc_variants_run_cb(p.ref(), fn, None, VARIANT_MODES[mode], comb, 1 if nosplit else 0, vcf_path.encode(),
                                   nm, arr, _u64(lens), n, out_path.encode(),
                                   debug_path.encode() if debug_path else None, log_path.encode() if log_path else None)
    if rc < 0:
        raise RuntimeError("orc_variants_run_cb failed: %d" % rc)
    return rc
