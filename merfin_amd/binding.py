"""ctypes binding of include/merfin_amd.h.  Names follow the reference's
objects: Index <-> the merylExactLookup pair (merfin-globals.H:217,220),
Sequences <-> the dnaSeq records loadSequence produces (merfin.C:30-53),
Evaluator <-> merfinGlobal's K* parameters + process*/output* callbacks."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TILE = 4096

MFX_ERRORS = {-1: "INVAL", -2: "NOMEM", -3: "HIP", -4: "FULL", -5: "OVERFLOW", -6: "IO", -7: "FORMAT", -8: "NODEVICE", -9: "NONCANON"}
E_NONCANON = -9


class MfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("merfin_amd error %s (%d): %s" % (MFX_ERRORS.get(code, "?"), code, msg))
        self.code = code


def lib_path():
    # MFX_LIB: another build of the same sources (kernel A/B experiments: tools/ab_build.sh); never a different ABI
    return os.environ.get("MFX_LIB") or os.path.join(_HERE, "libmerfin_amd.so")


class _KP(C.Structure):
    _fields_ = [("peak", C.c_double), ("n_prob", C.c_uint32),
                ("probK", C.POINTER(C.c_uint32)), ("probP", C.POINTER(C.c_double))]


class _Info(C.Structure):
    _fields_ = [("k", C.c_int), ("canonical", C.c_int), ("capacity", C.c_uint64),
                ("distinct", C.c_uint64), ("bytes", C.c_uint64), ("seq_only", C.c_int), ("compact", C.c_int),
                ("dropped", C.c_uint64)]


class _DbInfo(C.Structure):
    _fields_ = [("k", C.c_int), ("format", C.c_int), ("n_kmers", C.c_uint64), ("placed", C.c_int)]


class _VarOpts(C.Structure):
    _fields_ = [("mode", C.c_int), ("comb", C.c_uint32), ("nosplit", C.c_int), ("debug_path", C.c_char_p)]


INDEX_HEADER_BYTES = 128          # MFX_INDEX_HEADER_BYTES

VARIANT_MODES = {"filter": 4, "polish": 5, "better": 6, "strict": 7, "loose": 8}


class _HistResult(C.Structure):
    _fields_ = [("kasm", C.c_uint64), ("kmissing", C.c_uint64), ("koverCpy", C.c_double),
                ("undrMax", C.c_uint32), ("overMax", C.c_uint32),
                ("undr", C.POINTER(C.c_uint64)), ("over", C.POINTER(C.c_uint64)),
                ("ncontigs", C.c_uint32),
                ("contig_kasm", C.POINTER(C.c_uint64)), ("contig_kmissing", C.POINTER(C.c_uint64))]


_lib = None

# every symbol include/merfin_amd.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "mfx_last_error", "mfx_last_error_code", "mfx_version", "mfx_device_count", "mfx_device_warm", "mfx_device_memory",
    "mfx_index_create", "mfx_index_free", "mfx_index_estimate_gb", "mfx_index_add_read", "mfx_index_add_asm",
    "mfx_index_count_asm", "mfx_index_build_for_hist", "mfx_index_count_claimed", "mfx_hist_run_parts", "mfx_index_create_for_seq", "mfx_index_create_for_seq_lf", "mfx_index_create_lf", "mfx_db_stage_begin", "mfx_index_build_for_hist_staged", "mfx_index_load_db_staged", "mfx_db_stage_free", "mfx_db_stage_boost", "mfx_index_estimate_gb_for_seq", "mfx_index_claim_seq", "mfx_index_value", "mfx_index_get_info", "mfx_index_export",
    "mfx_db_probe", "mfx_index_load_db", "mfx_index_load_db_multi", "mfx_db_write_flat", "mfx_db_convert", "mfx_db_convert_placed", "mfx_db_write_flat_placed", "mfx_db_place_keys", "mfx_index_save", "mfx_index_load",
    "mfx_index_set_fingerprint", "mfx_index_get_origin",
    "mfx_host_alloc", "mfx_host_free", "mfx_seq_create", "mfx_hist_run_streamed",
    "mfx_hist_run_streamed_multi", "mfx_hist_run_streamed_range", "mfx_hist_stream_share",
    "mfx_index_replicate", "mfx_seq_replicate", "mfx_hist_run_multi", "mfx_hist_run_sharded",
    "mfx_comm_unique_id", "mfx_comm_create", "mfx_comm_free", "mfx_comm_rank", "mfx_comm_size", "mfx_comm_barrier", "mfx_comm_exchange_counts", "mfx_comm_alltoallv",
    "mfx_index_replicate_many", "mfx_seq_replicate_many", "mfx_seq_pack",
    "mfx_hist_allreduce", "mfx_hist_allgather_overflow", "mfx_hist_result_add_overflow",
    "mfx_index_image_header", "mfx_index_create_from_header", "mfx_index_device_image", "mfx_index_commit",
    "mfx_seq_upload", "mfx_seq_from_device", "mfx_seq_free", "mfx_seq_num_contigs", "mfx_seq_num_bases",
    "mfx_seq_num_tiles",
    "mfx_diag_gather_rate",
    "mfx_eval_create", "mfx_eval_free", "mfx_eval_nbins", "mfx_eval_debug_enable", "mfx_eval_debug_counters", "mfx_getK", "mfx_getKmetric", "mfx_histoQV",
    "mfx_hist_run", "mfx_hist_result_free", "mfx_hist_launch", "mfx_hist_launch_cyclic", "mfx_hist_result_from_counts",
    "mfx_hist_take_overflow", "mfx_hist_report", "mfx_diag_stream_rates",
    "mfx_pack_bases", "mfx_host_threads_share", "mfx_dump_values", "mfx_dump_contig", "mfx_dump_values_sharded", "mfx_dump_contig_sharded", "mfx_variants_run_sharded", "mfx_vcf_load", "mfx_vcf_free", "mfx_variants_run_vcf", "mfx_vcf_prepare", "mfx_vcf_path_bound", "mfx_index_claim_paths", "mfx_vcf_prepare_path_index", "mfx_completeness", "mfx_completeness_pieces", "mfx_variants_run",
    "mfx_index_set_shard", "mfx_router_create", "mfx_router_free", "mfx_route_tiles", "mfx_hist_keys_launch",
]


def _share_hip_runtime_with_torch():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their
    own libamdhip64.so; if libmerfin_amd.so pulled in /opt/rocm's copy first, a
    later `import torch` would start a second runtime that sees no GPU.  So when
    torch is installed, map its copy first (same SONAME: ours then binds to it)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """Load libmerfin_amd.so.  Fails loudly when the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("merfin_amd: %s is missing -- build it with `make -C merfin_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % p)
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    u64p, u32p, f64p, vp = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.c_void_p
    L.mfx_last_error.restype = C.c_char_p
    L.mfx_version.restype = C.c_char_p
    L.mfx_last_error_code.restype = C.c_int
    L.mfx_device_count.restype = C.c_int
    L.mfx_device_warm.argtypes = [C.c_int]
    L.mfx_device_memory.argtypes = [C.c_int, u64p, u64p]
    L.mfx_index_create.restype = vp
    L.mfx_index_create.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_int]
    L.mfx_index_free.argtypes = [vp]
    L.mfx_index_estimate_gb.restype = C.c_double
    L.mfx_index_estimate_gb.argtypes = [C.c_int, C.c_uint64]
    L.mfx_index_add_read.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.mfx_index_add_asm.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
    L.mfx_index_count_asm.argtypes = [vp, vp, vp]
    L.mfx_index_count_claimed.argtypes = [vp, vp, vp]
    L.mfx_index_build_for_hist.argtypes = [vp, vp, C.c_char_p, C.c_uint64, C.c_uint64]
    L.mfx_hist_run_parts.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint32)), C.c_uint32, C.c_uint32, vp]
    L.mfx_index_create_for_seq.restype = vp
    L.mfx_index_create_for_seq.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_int]
    L.mfx_index_create_lf.restype = vp
    L.mfx_index_create_lf.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_int, C.c_double]
    L.mfx_index_create_for_seq_lf.restype = vp
    L.mfx_index_create_for_seq_lf.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_int, C.c_double]
    L.mfx_db_stage_begin.restype = vp
    L.mfx_db_stage_begin.argtypes = [C.c_char_p, C.c_int]
    L.mfx_index_build_for_hist_staged.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
    L.mfx_index_load_db_staged.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64]
    L.mfx_db_stage_free.restype = None
    L.mfx_db_stage_free.argtypes = [vp]
    L.mfx_db_stage_boost.restype = None
    L.mfx_db_stage_boost.argtypes = [vp]
    L.mfx_index_estimate_gb_for_seq.restype = C.c_double
    L.mfx_index_estimate_gb_for_seq.argtypes = [C.c_int, C.c_uint64]
    L.mfx_index_claim_seq.argtypes = [vp, vp, vp]
    L.mfx_index_value.argtypes = [vp, u64p, C.c_uint64, u32p, u32p]
    L.mfx_index_get_info.argtypes = [vp, C.POINTER(_Info)]
    L.mfx_index_export.argtypes = [vp, u64p, u32p, u32p, u64p]
    L.mfx_db_probe.argtypes = [C.c_char_p, C.POINTER(_DbInfo)]
    L.mfx_index_load_db.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint64, C.c_uint64]
    L.mfx_db_write_flat.argtypes = [C.c_char_p, C.c_int, u64p, u32p, C.c_uint64]
    L.mfx_db_convert.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]
    L.mfx_db_convert_placed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]
    L.mfx_db_write_flat_placed.argtypes = [C.c_char_p, C.c_int, u64p, u32p, C.c_uint64]
    L.mfx_db_place_keys.argtypes = [C.c_int, vp, C.c_uint64, vp, C.c_int, C.c_int]
    L.mfx_index_load_db_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.c_char_p, C.c_int, C.c_uint64, C.c_uint64]
    L.mfx_index_save.argtypes = [vp, C.c_char_p]
    L.mfx_index_load.restype = vp
    L.mfx_index_load.argtypes = [C.c_char_p, C.c_double, C.c_int]
    L.mfx_index_image_header.argtypes = [vp, vp]
    L.mfx_index_create_from_header.restype = vp
    L.mfx_index_create_from_header.argtypes = [vp, C.c_double, C.c_int]
    L.mfx_index_device_image.argtypes = [vp, C.POINTER(vp), u64p, C.POINTER(vp), u64p]
    L.mfx_index_commit.argtypes = [vp]
    L.mfx_seq_upload.restype = vp
    L.mfx_seq_upload.argtypes = [C.c_int, C.POINTER(C.c_char_p), u64p, C.c_uint32]
    L.mfx_seq_from_device.restype = vp
    L.mfx_seq_from_device.argtypes = [C.c_int, C.POINTER(vp), u64p, C.c_uint32, vp]
    L.mfx_seq_free.argtypes = [vp]
    L.mfx_seq_num_contigs.restype = C.c_uint32
    L.mfx_seq_num_contigs.argtypes = [vp]
    L.mfx_seq_num_bases.restype = C.c_uint64
    L.mfx_seq_num_bases.argtypes = [vp]
    L.mfx_seq_num_tiles.restype = C.c_uint64
    L.mfx_seq_num_tiles.argtypes = [vp]
    L.mfx_eval_create.restype = vp
    L.mfx_eval_create.argtypes = [vp, C.POINTER(_KP), C.c_uint32]
    L.mfx_eval_free.argtypes = [vp]
    L.mfx_eval_nbins.restype = C.c_uint32
    L.mfx_eval_nbins.argtypes = [vp]
    L.mfx_diag_gather_rate.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    L.mfx_eval_debug_enable.argtypes = [vp, C.c_int]
    L.mfx_eval_debug_counters.argtypes = [vp, u64p]
    L.mfx_getK.argtypes = [C.POINTER(_KP), C.c_uint32, C.c_uint32, f64p, f64p, f64p]
    L.mfx_getKmetric.restype = C.c_double
    L.mfx_getKmetric.argtypes = [C.c_double, C.c_double]
    L.mfx_histoQV.restype = C.c_double
    L.mfx_histoQV.argtypes = [C.c_double, C.c_double, C.c_int]
    L.mfx_hist_run.argtypes = [vp, vp, C.POINTER(_HistResult)]
    L.mfx_hist_result_free.argtypes = [C.POINTER(_HistResult)]
    L.mfx_hist_launch.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp]
    L.mfx_hist_launch_cyclic.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.mfx_hist_result_from_counts.argtypes = [C.c_uint32, u64p, C.c_double, C.c_uint32, C.POINTER(_HistResult)]
    L.mfx_hist_take_overflow.argtypes = [vp, u64p, C.c_uint64, u64p]
    L.mfx_diag_stream_rates.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mfx_hist_report.argtypes = [C.POINTER(_HistResult), C.c_int, C.c_char_p, C.c_char_p]
    L.mfx_dump_values.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, u32p, u32p, u64p, u64p]
    L.mfx_dump_contig.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, u64p, u64p]
    L.mfx_dump_values_sharded.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u32p, u32p, u64p, u64p]
    L.mfx_dump_contig_sharded.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, u64p, u64p]
    L.mfx_variants_run_sharded.argtypes = [vp, C.c_uint32, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32,
                                           C.POINTER(_VarOpts), C.c_char_p, C.c_char_p, u64p]
    L.mfx_completeness.argtypes = [vp, f64p, f64p]
    L.mfx_completeness_pieces.argtypes = [vp, f64p, f64p]
    L.mfx_index_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mfx_router_create.restype = vp
    L.mfx_router_create.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mfx_router_free.argtypes = [vp]
    L.mfx_route_tiles.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp, u64p, vp]
    L.mfx_hist_keys_launch.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]
    L.mfx_variants_run.argtypes = [vp, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32,
                                   C.POINTER(_VarOpts), C.c_char_p, C.c_char_p, u64p]
    L.mfx_vcf_load.restype = vp
    L.mfx_vcf_load.argtypes = [C.c_char_p]
    L.mfx_vcf_free.restype = None
    L.mfx_vcf_free.argtypes = [vp]
    L.mfx_variants_run_vcf.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32,
                                       C.POINTER(_VarOpts), C.c_char_p, C.c_char_p, u64p]
    L.mfx_vcf_prepare.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32, C.POINTER(_VarOpts)]
    L.mfx_vcf_path_bound.argtypes = [vp, u64p]
    L.mfx_index_claim_paths.argtypes = [vp, vp, u64p]
    L.mfx_vcf_prepare_path_index.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_uint32, C.POINTER(_VarOpts), C.c_double, C.c_int, C.c_double,
                                             C.POINTER(vp)]
    L.mfx_index_set_fingerprint.argtypes = [vp, C.c_uint64]
    L.mfx_index_get_origin.argtypes = [vp, u64p, u64p, u64p]
    L.mfx_host_threads_share.restype = None
    L.mfx_host_threads_share.argtypes = [C.c_uint]
    L.mfx_pack_bases.restype = None
    L.mfx_pack_bases.argtypes = [vp, C.c_uint64, vp, vp]
    L.mfx_host_alloc.restype = vp
    L.mfx_host_alloc.argtypes = [C.c_size_t]
    L.mfx_host_free.restype = None
    L.mfx_host_free.argtypes = [vp]
    L.mfx_seq_create.restype = vp
    L.mfx_seq_create.argtypes = [C.c_int, u64p, C.c_uint32]
    L.mfx_hist_run_streamed.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(_HistResult)]
    L.mfx_index_replicate.restype = vp
    L.mfx_index_replicate.argtypes = [vp, C.c_int]
    L.mfx_seq_replicate.restype = vp
    L.mfx_seq_replicate.argtypes = [vp, C.c_int]
    L.mfx_hist_run_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(_HistResult)]
    L.mfx_hist_run_streamed_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(vp), C.POINTER(_HistResult)]
    L.mfx_hist_run_streamed_range.argtypes = [vp, vp, C.POINTER(vp), C.c_uint64, C.c_uint64, vp, vp]
    L.mfx_hist_stream_share.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mfx_hist_run_sharded.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(_HistResult)]
    L.mfx_comm_unique_id.argtypes = [vp]
    L.mfx_comm_create.restype = vp
    L.mfx_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.mfx_comm_free.restype = None
    L.mfx_comm_free.argtypes = [vp]
    L.mfx_comm_rank.argtypes = [vp]
    L.mfx_comm_size.argtypes = [vp]
    L.mfx_comm_barrier.argtypes = [vp, vp]
    L.mfx_comm_exchange_counts.argtypes = [vp, u64p, u64p, vp]
    L.mfx_comm_alltoallv.argtypes = [vp, vp, u64p, vp, u64p, C.c_uint32, vp]
    L.mfx_index_replicate_many.argtypes = [vp, C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.mfx_seq_replicate_many.argtypes = [vp, C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.mfx_seq_pack.argtypes = [vp]
    L.mfx_hist_allreduce.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.mfx_hist_allgather_overflow.argtypes = [vp, vp, u64p, C.c_uint64, u64p, vp]
    L.mfx_hist_result_add_overflow.argtypes = [C.POINTER(_HistResult), u64p, C.c_uint64]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MfxError(rc, load_library().mfx_last_error().decode())


def _need(ptr):
    if not ptr:
        L = load_library()
        raise MfxError(L.mfx_last_error_code() or -3, L.mfx_last_error().decode())
    return ptr


def device_count():
    return load_library().mfx_device_count()


def device_warm(device=0):
    """context, code object and pinned-memory path of the device brought up now (mfx_device_warm)"""
    _check(load_library().mfx_device_warm(device))


def device_memory(device=0):
    """(free, total) bytes of the device's memory (mfx_device_memory)"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(load_library().mfx_device_memory(device, C.byref(f), C.byref(t)))
    return f.value, t.value


def hist_words(nbins, ncontigs):
    return 2 * nbins + 3 + 2 * ncontigs


class KParams:
    """-peak and the -prob table (merfin-globals.H:226-227,239)."""

    def __init__(self, peak, probK=None, probP=None):
        self.peak = float(peak)
        self.probK = np.ascontiguousarray(probK if probK is not None else [], dtype=np.uint32)
        self.probP = np.ascontiguousarray(probP if probP is not None else [], dtype=np.float64)
        assert len(self.probK) == len(self.probP)
        self.c = _KP(self.peak, len(self.probK),
                     self.probK.ctypes.data_as(C.POINTER(C.c_uint32)), self.probP.ctypes.data_as(C.POINTER(C.c_double)))

    @staticmethod
    def from_file(peak, path):
        """load_Kmetric (merfin-globals.C:21-62): lines with exactly two comma-separated fields."""
        K, P = [], []
        with open(path) as f:
            for line in f:
                w = [x for x in line.rstrip("\r\n").split(",") if x != ""]
                if len(w) == 2:
                    K.append(int(w[0]))
                    P.append(float(w[1]))
        return KParams(peak, K, P)


def getK(kp, readV, asmV):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    load_library().mfx_getK(C.byref(kp.c), readV, asmV, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def getKmetric(readK, asmK):
    return load_library().mfx_getKmetric(readK, asmK)


def histoQV(kval, ktot, k):
    return load_library().mfx_histoQV(kval, ktot, k)


def db_probe(path):
    """merylFileReader(path): (k, format, n_kmers) of a k-mer database on disk"""
    i = _DbInfo()
    _check(load_library().mfx_db_probe(path.encode(), C.byref(i)))
    d = {"k": i.k, "format": {1: "meryl", 2: "text", 3: "flat"}.get(i.format), "n_kmers": i.n_kmers}
    if i.placed:
        d["placed"] = True                                      # (only a placed flat file says so: mfx_db_convert_placed)
    return d


def load_db_multi(indexes, path, side, minV=0, maxV=2**64 - 1):
    """one pass over a k-mer database into several tables (the shards of one process)"""
    n = len(indexes)
    arr = (C.c_void_p * n)(*[ix.h for ix in indexes])
    _check(load_library().mfx_index_load_db_multi(arr, n, path.encode(), side, minV, maxV))


def db_convert(in_path, out_path):
    """any accepted database -> the flat form (sorted: delta-coded blocks); returns the number of k-mers (mfx_db_convert, host only)"""
    n = C.c_uint64(0)
    _check(load_library().mfx_db_convert(in_path.encode(), out_path.encode(), C.byref(n)))
    return n.value


def db_convert_placed(in_path, out_path):
    """any accepted canonical database (13 <= k <= 30) -> the PLACED flat form: records sorted by their place in the compact table
    (mfx_db_convert_placed, host only); returns the number of k-mers"""
    n = C.c_uint64(0)
    _check(load_library().mfx_db_convert_placed(in_path.encode(), out_path.encode(), C.byref(n)))
    return n.value


def db_place_keys(k, kmers, out=None, device=0):
    """the placement numbers P (csrc/mfx_place.h) of k-mers: numpy uint64 arrays on the host, or torch int64 CUDA tensors on the device"""
    if isinstance(kmers, np.ndarray):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.empty_like(kmers) if out is None else out
        _check(load_library().mfx_db_place_keys(k, C.c_void_p(kmers.ctypes.data), len(kmers), C.c_void_p(out.ctypes.data), 0, device))
        return out
    import torch
    out = torch.empty_like(kmers) if out is None else out
    _check(load_library().mfx_db_place_keys(k, C.c_void_p(kmers.data_ptr()), kmers.numel(), C.c_void_p(out.data_ptr()), 1, device))
    return out


def db_write_flat_placed(path, k, pkeys, values):
    """ascending placement numbers (db_place_keys of the k-mers, sorted) and their counts -> a placed flat file"""
    pkeys = np.ascontiguousarray(pkeys, dtype=np.uint64)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    _check(load_library().mfx_db_write_flat_placed(path.encode(), k, pkeys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   values.ctypes.data_as(C.POINTER(C.c_uint32)), len(pkeys)))


def db_write_flat(path, k, kmers, values):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    _check(load_library().mfx_db_write_flat(path.encode(), k, kmers.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            values.ctypes.data_as(C.POINTER(C.c_uint32)), len(kmers)))


def _ptr(x):
    """host numpy array or device pointer (int / torch tensor) -> (void*, on_device, keepalive)"""
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data), 0, x
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr()), (1 if x.is_cuda else 0), x
    return C.c_void_p(int(x)), 1, None


class DbStage:
    """A delta-coded flat database on its way into device memory (mfx_db_stage_begin); None-like (ok == False) when it cannot be staged."""

    def __init__(self, path, device=0):
        self.h = load_library().mfx_db_stage_begin(path.encode(), device)
        self.ok = bool(self.h)
        self.why = None if self.ok else load_library().mfx_last_error().decode()

    def boost(self):
        if self.h:
            load_library().mfx_db_stage_boost(self.h)

    def close(self):
        if self.h:
            load_library().mfx_db_stage_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """Joint read+assembly k-mer count table resident in HBM."""

    def __init__(self, k, capacity_kmers, max_gb=0.0, device=0, _handle=None, seq_only=False, load_factor=0.0):
        L = load_library()
        self.k = k
        self.device = device
        if _handle is not None:
            self.h = _handle
        elif seq_only and load_factor:
            self.h = _need(L.mfx_index_create_for_seq_lf(k, int(capacity_kmers), float(max_gb), device, float(load_factor)))
        elif seq_only:
            self.h = _need(L.mfx_index_create_for_seq(k, int(capacity_kmers), float(max_gb), device))
        elif load_factor:
            self.h = _need(L.mfx_index_create_lf(k, int(capacity_kmers), float(max_gb), device, float(load_factor)))
        else:
            self.h = _need(L.mfx_index_create(k, int(capacity_kmers), float(max_gb), device))

    @staticmethod
    def for_seq(k, capacity_kmers, max_gb=0.0, device=0, load_factor=0.0):
        """a SEQUENCE-ONLY index (mfx_index_create_for_seq): claim the k-mers of the sequence first (count_asm or
        claim_seq), then add / load -- those only update the claimed k-mers.  For -hist and -dump."""
        return Index(k, capacity_kmers, max_gb=max_gb, device=device, seq_only=True, load_factor=load_factor)

    def count_claimed(self, seqs, stream=None):
        """asmV += 1 per occurrence, in `seqs`, of a k-mer claimed before (claim_seq); nothing is claimed"""
        _check(load_library().mfx_index_count_claimed(self.h, seqs.h, C.c_void_p(stream or 0)))

    def build_for_hist(self, seqs, read_db_path, minV=0, maxV=2**64 - 1):
        """count_asm(seqs) + load_db(read_db_path, 0, minV, maxV) in one call: the database crosses PCIe while the sequence's k-mers are claimed"""
        _check(load_library().mfx_index_build_for_hist(self.h, seqs.h, read_db_path.encode(), minV, maxV))

    def build_for_hist_staged(self, seqs, stage, minV=0, maxV=2**64 - 1):
        """the same from a DbStage (mfx_db_stage_begin): the database has been on its way into device memory since the stage was made"""
        _check(load_library().mfx_index_build_for_hist_staged(self.h, seqs.h, stage.h, minV, maxV))

    def load_db_staged(self, stage, side, minV=0, maxV=2**64 - 1):
        """load_db of a DbStage: only the decode + insert kernels are left to run (side 0 -readmers, 1 -seqmers)"""
        _check(load_library().mfx_index_load_db_staged(self.h, stage.h, int(side), minV, maxV))

    def claim_seq(self, seqs, stream=None):
        _check(load_library().mfx_index_claim_seq(self.h, seqs.h, C.c_void_p(stream or 0)))

    def save(self, path):
        """write the built table as a device-format image"""
        _check(load_library().mfx_index_save(self.h, path.encode()))

    @staticmethod
    def load(path, max_gb=0.0, device=0):
        h = _need(load_library().mfx_index_load(path.encode(), float(max_gb), device))
        ix = Index(0, 0, device=device, _handle=h)
        ix.k = ix.info()["k"]
        return ix

    def image_header(self):
        """geometry + filter of the built table (INDEX_HEADER_BYTES bytes), see Index.from_header"""
        buf = np.zeros(INDEX_HEADER_BYTES, dtype=np.uint8)
        _check(load_library().mfx_index_image_header(self.h, C.c_void_p(buf.ctypes.data)))
        return buf

    @staticmethod
    def from_header(header, max_gb=0.0, device=0):
        """an empty index of exactly that geometry; fill device_image() (e.g. by a broadcast), then commit()"""
        header = np.ascontiguousarray(header, dtype=np.uint8)
        h = _need(load_library().mfx_index_create_from_header(C.c_void_p(header.ctypes.data), float(max_gb), device))
        ix = Index(0, 0, device=device, _handle=h)
        ix.k = ix.info()["k"]
        return ix

    def device_image(self):
        """(lines device pointer, bytes, meta device pointer, bytes) of the table in HBM"""
        pl, pm = C.c_void_p(), C.c_void_p()
        nl, nm = C.c_uint64(), C.c_uint64()
        _check(load_library().mfx_index_device_image(self.h, C.byref(pl), C.byref(nl), C.byref(pm), C.byref(nm)))
        return pl.value, nl.value, pm.value, nm.value

    def commit(self):
        _check(load_library().mfx_index_commit(self.h))

    def replicate(self, device):
        """a copy of the built table on another device of the node (peer copy over xGMI), no second build"""
        h = _need(load_library().mfx_index_replicate(self.h, device))
        ix = Index(0, 0, device=device, _handle=h)
        ix.k = self.k
        return ix

    def replicate_many(self, devices):
        """copies on several devices at once: a doubling tree over xGMI (every device that holds the table feeds another one
        in each round, all copies of a round in flight together)"""
        devs = (C.c_int * len(devices))(*devices)
        out = (C.c_void_p * len(devices))()
        _check(load_library().mfx_index_replicate_many(self.h, devs, len(devices), out))
        res = []
        for d, h in zip(devices, out):
            ix = Index(0, 0, device=d, _handle=h)
            ix.k = self.k
            res.append(ix)
        return res

    def set_fingerprint(self, fp):
        _check(load_library().mfx_index_set_fingerprint(self.h, int(fp)))

    def origin(self):
        """(fingerprint, minV, maxV) stored with the table"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load_library().mfx_index_get_origin(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def add_read(self, kmers, values, minV=0, maxV=2**64 - 1):
        if isinstance(kmers, np.ndarray):
            kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
            values = np.ascontiguousarray(values, dtype=np.uint32)
        pk, dev, _k1 = _ptr(kmers)
        pv, _, _k2 = _ptr(values)
        _check(load_library().mfx_index_add_read(self.h, pk, pv, len(kmers), minV, maxV, dev))

    def add_asm(self, kmers, values):
        if isinstance(kmers, np.ndarray):
            kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
            values = np.ascontiguousarray(values, dtype=np.uint32)
        pk, dev, _k1 = _ptr(kmers)
        pv, _, _k2 = _ptr(values)
        _check(load_library().mfx_index_add_asm(self.h, pk, pv, len(kmers), dev))

    def set_shard(self, rank, nranks):
        """Sharded index: keep only the k-mers owned by `rank` of `nranks` (call before loading)."""
        _check(load_library().mfx_index_set_shard(self.h, rank, nranks))

    def load_db(self, path, side, minV=0, maxV=2**64 - 1):
        """side 0: read DB (-min/-max), side 1: assembly DB"""
        _check(load_library().mfx_index_load_db(self.h, path.encode(), side, minV, maxV))

    def count_asm(self, seqs, stream=None):
        _check(load_library().mfx_index_count_asm(self.h, seqs.h, C.c_void_p(stream or 0)))

    def value(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        r = np.zeros(len(kmers), dtype=np.uint32)
        a = np.zeros(len(kmers), dtype=np.uint32)
        _check(load_library().mfx_index_value(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers),
                                              r.ctypes.data_as(C.POINTER(C.c_uint32)), a.ctypes.data_as(C.POINTER(C.c_uint32))))
        return r, a

    def info(self):
        i = _Info()
        _check(load_library().mfx_index_get_info(self.h, C.byref(i)))
        return {"k": i.k, "canonical": bool(i.canonical), "capacity": i.capacity, "distinct": i.distinct, "bytes": i.bytes,
                "seq_only": bool(i.seq_only), "compact": bool(i.compact), "dropped": i.dropped}

    def export(self, sort=True):
        """every stored (k-mer, readV, asmV), sorted by k-mer (sort=False: table order).  k > 31: k-mers are rows
        [low 64 bits, high bits]"""
        n = self.info()["distinct"]
        if self.k > 31:
            k = np.zeros((max(n, 1), 2), dtype=np.uint64)
            r = np.zeros(max(n, 1), dtype=np.uint32)
            a = np.zeros(max(n, 1), dtype=np.uint32)
            cnt = C.c_uint64(0)
            _check(load_library().mfx_index_export(self.h, k.ctypes.data_as(C.POINTER(C.c_uint64)), r.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                   a.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cnt)))
            k, r, a = k[:cnt.value], r[:cnt.value], a[:cnt.value]
            if not sort:
                return k, r, a
            o = np.lexsort((k[:, 0], k[:, 1]))
            return k[o], r[o], a[o]
        k = np.zeros(max(n, 1), dtype=np.uint64)
        r = np.zeros(max(n, 1), dtype=np.uint32)
        a = np.zeros(max(n, 1), dtype=np.uint32)
        cnt = C.c_uint64(0)
        _check(load_library().mfx_index_export(self.h, k.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               r.ctypes.data_as(C.POINTER(C.c_uint32)), a.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               C.byref(cnt)))
        if not sort:
            return k[:cnt.value], r[:cnt.value], a[:cnt.value]
        o = np.argsort(k[:cnt.value])
        return k[:cnt.value][o], r[:cnt.value][o], a[:cnt.value][o]

    def close(self):
        if getattr(self, "h", None):
            load_library().mfx_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Sequences:
    """All contigs of the assembly packed into one HBM buffer."""

    def __init__(self, contigs=None, device=0, names=None, _handle=None):
        L = load_library()
        self.device = device
        if _handle is not None:
            self.h = _handle
        else:
            n = len(contigs)
            arr = (C.c_char_p * n)(*contigs)
            lens = np.array([len(c) for c in contigs], dtype=np.uint64)
            self.h = _need(L.mfx_seq_upload(device, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n))
        self.names = names

    @staticmethod
    def from_device(ptrs, lens, device=0, stream=None, names=None):
        L = load_library()
        n = len(ptrs)
        arr = (C.c_void_p * n)(*[int(p) for p in ptrs])
        ln = np.array(lens, dtype=np.uint64)
        h = _need(L.mfx_seq_from_device(device, arr, ln.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.c_void_p(stream or 0)))
        return Sequences(device=device, names=names, _handle=h)

    @staticmethod
    def create(lens, device=0, names=None):
        """layout + device buffers of an assembly whose bases arrive later (Evaluator.hist_streamed)"""
        ln = np.array(lens, dtype=np.uint64)
        h = _need(load_library().mfx_seq_create(device, ln.ctypes.data_as(C.POINTER(C.c_uint64)), len(ln)))
        return Sequences(device=device, names=names, _handle=h)

    def replicate(self, device):
        """a copy of the packed assembly on another device of the node"""
        return Sequences(device=device, names=self.names, _handle=_need(load_library().mfx_seq_replicate(self.h, device)))

    def replicate_many(self, devices):
        """copies on several devices at once (doubling tree over xGMI); the assembly travels as its packed planes"""
        devs = (C.c_int * len(devices))(*devices)
        out = (C.c_void_p * len(devices))()
        _check(load_library().mfx_seq_replicate_many(self.h, devs, len(devices), out))
        return [Sequences(device=d, names=self.names, _handle=h) for d, h in zip(devices, out)]

    def pack(self):
        """build the packed planes (2-bit codes + validity bits) from the resident bases, on the device"""
        _check(load_library().mfx_seq_pack(self.h))

    @property
    def ncontigs(self):
        return load_library().mfx_seq_num_contigs(self.h)

    @property
    def nbases(self):
        return load_library().mfx_seq_num_bases(self.h)

    @property
    def ntiles(self):
        return load_library().mfx_seq_num_tiles(self.h)

    def close(self):
        if getattr(self, "h", None):
            load_library().mfx_seq_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class HistResult:
    """merfinGlobal's histogram accumulators (merfin-globals.H:184-192)."""

    def __init__(self):
        self.c = _HistResult()

    kasm = property(lambda s: s.c.kasm)
    kmissing = property(lambda s: s.c.kmissing)
    koverCpy = property(lambda s: s.c.koverCpy)

    def undr(self):
        return np.ctypeslib.as_array(self.c.undr, shape=(self.c.undrMax,)).copy()

    def over(self):
        return np.ctypeslib.as_array(self.c.over, shape=(self.c.overMax,)).copy()

    def contig_kasm(self):
        return np.ctypeslib.as_array(self.c.contig_kasm, shape=(max(self.c.ncontigs, 1),))[:self.c.ncontigs].copy()

    def contig_kmissing(self):
        return np.ctypeslib.as_array(self.c.contig_kmissing, shape=(max(self.c.ncontigs, 1),))[:self.c.ncontigs].copy()

    def add_overflow(self, records):
        """fold K* bins >= nbins (Evaluator.take_overflow / Comm.allgather_overflow: {key, occurrences} pairs, shape (n, 2)) into this result"""
        rec = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1)
        if len(rec):
            _check(load_library().mfx_hist_result_add_overflow(C.byref(self.c), rec.ctypes.data_as(C.POINTER(C.c_uint64)), len(rec) // 2))
        return self

    def report(self, k, hist_path=None, summary_path=None):
        _check(load_library().mfx_hist_report(C.byref(self.c), k, hist_path.encode() if hist_path else None,
                                              summary_path.encode() if summary_path else None))

    def __del__(self):
        try:
            load_library().mfx_hist_result_free(C.byref(self.c))
        except Exception:
            pass


def result_from_counts(nbins, h_counts, kover, ncontigs):
    """(all-reduced) counts image -> HistResult; host-only, needs no device"""
    h_counts = np.ascontiguousarray(h_counts, dtype=np.uint64)
    assert len(h_counts) >= hist_words(nbins, ncontigs)
    r = HistResult()
    _check(load_library().mfx_hist_result_from_counts(nbins, h_counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      float(kover), ncontigs, C.byref(r.c)))
    return r


class Router:
    """Source side of the sharded -hist: groups the k-mers of sequence tiles by owner rank."""

    def __init__(self, index, nranks, max_tiles):
        self.index = index
        self.nranks = nranks
        self.max_tiles = max_tiles
        self.h = _need(load_library().mfx_router_create(index.h, nranks, max_tiles))

    def route(self, seqs, tile_begin, tile_end, nbins, d_counts, d_keys_out, d_contigs_out, stream=None):
        p = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        dest = np.zeros(self.nranks, dtype=np.uint64)
        _check(load_library().mfx_route_tiles(self.h, seqs.h, tile_begin, tile_end, nbins, p(d_counts), p(d_keys_out),
                                              p(d_contigs_out), dest.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_void_p(stream or 0)))
        return dest

    def close(self):
        if getattr(self, "h", None):
            load_library().mfx_router_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class PinnedBuffer:
    """page-locked host memory from the library (mfx_host_alloc): what a loader should read the assembly into, so that
    the streamed upload DMAs it in place.  .array is a uint8 numpy view."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.p = _need(load_library().mfx_host_alloc(self.nbytes))
        self.array = np.ctypeslib.as_array(C.cast(self.p, C.POINTER(C.c_uint8)), shape=(max(self.nbytes, 1),))[:self.nbytes]

    def close(self):
        if getattr(self, "p", None):
            self.array = None
            load_library().mfx_host_free(self.p)
            self.p = None

    def __del__(self):
        self.close()


def hist_multi(evaluators, sequences):
    """-hist over several devices driven by this one process (mfx_hist_run_multi): evaluators[d] / sequences[d] are
    replicas on device d; slot d evaluates the block-cyclic share d of N of the tiles, all at once."""
    n = len(evaluators)
    assert n == len(sequences) and n >= 1
    ev = (C.c_void_p * n)(*[e.h for e in evaluators])
    sq = (C.c_void_p * n)(*[s.h for s in sequences])
    r = HistResult()
    _check(load_library().mfx_hist_run_multi(ev, sq, n, C.byref(r.c)))
    return r


def _host_ptrs(host_contigs):
    """(ctypes void* array, objects to keep alive) of bytes / numpy uint8 arrays / PinnedBuffer views, one per contig"""
    n = len(host_contigs)
    keep = []
    ptrs = (C.c_void_p * max(n, 1))()
    for i, c in enumerate(host_contigs):
        if isinstance(c, (bytes, bytearray)):
            b = C.c_char_p(bytes(c))
            keep.append(b)
            ptrs[i] = C.cast(b, C.c_void_p).value
        else:
            a = np.ascontiguousarray(c, dtype=np.uint8)
            keep.append(a)
            ptrs[i] = a.ctypes.data
    return ptrs, keep


def hist_streamed_multi(evaluators, sequences, host_contigs):
    """SURVEY 8(d)'s evaluate phase over several devices driven by this one process (mfx_hist_run_streamed_multi): slot d gets,
    through its own copy stream, only the packed planes of its contiguous share of the tiles and evaluates them as they land.
    sequences[d] = Sequences.create(lens, device=d): one object per slot (it holds the slot's part afterwards)."""
    n = len(evaluators)
    assert n == len(sequences) and n >= 1
    ev = (C.c_void_p * n)(*[e.h for e in evaluators])
    sq = (C.c_void_p * n)(*[s.h for s in sequences])
    ptrs, keep = _host_ptrs(host_contigs)
    r = HistResult()
    _check(load_library().mfx_hist_run_streamed_multi(ev, sq, n, ptrs, C.byref(r.c)))
    return r


def stream_share(ntiles, rank, nranks):
    """the tiles [begin, end) rank `rank` of `nranks` streams (mfx_hist_stream_share)"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    _check(load_library().mfx_hist_stream_share(ntiles, rank, nranks, C.byref(a), C.byref(b)))
    return a.value, b.value


def hist_sharded(evaluators, routers, sequences):
    """-hist over an index sharded across the slots of ONE process (mfx_hist_run_sharded): slot d = evaluator + router on
    shard d of N, plus the packed assembly on that shard's device"""
    n = len(evaluators)
    assert n == len(routers) == len(sequences) and n >= 1
    ev = (C.c_void_p * n)(*[e.h for e in evaluators])
    ro = (C.c_void_p * n)(*[r.h for r in routers])
    sq = (C.c_void_p * n)(*[s.h for s in sequences])
    r = HistResult()
    _check(load_library().mfx_hist_run_sharded(ev, ro, sq, n, C.byref(r.c)))
    return r


def pack_bases(seq):
    """host-side encoding of the packed sequence transport: bytes -> (codes uint64[ceil(n/32)], valid uint32[ceil(n/32)])"""
    src = np.frombuffer(bytes(seq), dtype=np.uint8)
    nw = (len(src) + 31) // 32
    codes = np.zeros(max(nw, 1), dtype=np.uint64)
    valid = np.zeros(max(nw, 1), dtype=np.uint32)
    load_library().mfx_pack_bases(C.c_void_p(src.ctypes.data), len(src), C.c_void_p(codes.ctypes.data), C.c_void_p(valid.ctypes.data))
    return codes[:nw], valid[:nw]


def dump_values_sharded(evaluators, sequences, contig, pos_begin, pos_end):
    """(readV, asmV) per k-mer start over an index sharded across the slots (mfx_dump_values_sharded)"""
    ns = len(evaluators)
    assert ns == len(sequences) and ns >= 1
    ev = (C.c_void_p * ns)(*[e.h for e in evaluators])
    sq = (C.c_void_p * ns)(*[s.h for s in sequences])
    n = pos_end - pos_begin
    r = np.zeros(max(n, 1), dtype=np.uint32)
    a = np.zeros(max(n, 1), dtype=np.uint32)
    ka, km = C.c_uint64(0), C.c_uint64(0)
    _check(load_library().mfx_dump_values_sharded(ev, sq, ns, contig, pos_begin, pos_end, r.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                  a.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ka), C.byref(km)))
    return r[:n], a[:n], ka.value, km.value


def dump_contig_sharded(evaluators, sequences, contig, name, path, append=False):
    ns = len(evaluators)
    ev = (C.c_void_p * ns)(*[e.h for e in evaluators])
    sq = (C.c_void_p * ns)(*[s.h for s in sequences])
    ka, km = C.c_uint64(0), C.c_uint64(0)
    _check(load_library().mfx_dump_contig_sharded(ev, sq, ns, contig, name.encode(), path.encode(), 1 if append else 0,
                                                  C.byref(ka), C.byref(km)))
    return ka.value, km.value


def variants_sharded(evaluators, mode, vcf_path, names, contigs, out_path, comb=15, nosplit=False, debug_path=None, log_path=None):
    """the variant modes over an index sharded across the slots (mfx_variants_run_sharded); returns clusters evaluated"""
    ns = len(evaluators)
    ev = (C.c_void_p * ns)(*[e.h for e in evaluators])
    n = len(contigs)
    nm = (C.c_char_p * n)(*[x.encode() for x in names])
    arr = (C.c_char_p * n)(*contigs)
    lens = np.array([len(c) for c in contigs], dtype=np.uint64)
    o = _VarOpts(VARIANT_MODES[mode], comb, 1 if nosplit else 0, debug_path.encode() if debug_path else None)
    ncl = C.c_uint64(0)
    _check(load_library().mfx_variants_run_sharded(ev, ns, vcf_path.encode(), nm, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                                   C.byref(o), out_path.encode(), log_path.encode() if log_path else None, C.byref(ncl)))
    return ncl.value


COMM_ID_BYTES = 128          # MFX_COMM_ID_BYTES


class Comm:
    """One rank of the multi-process -hist: RCCL communicator + the path's collective (csrc/mfx_comm.cpp)."""

    @staticmethod
    def unique_id():
        buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        _check(load_library().mfx_comm_unique_id(C.c_void_p(buf.ctypes.data)))
        return buf

    def __init__(self, uid, rank, nranks, device=0):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        assert len(uid) == COMM_ID_BYTES
        self.rank, self.nranks, self.device = rank, nranks, device
        self.h = _need(load_library().mfx_comm_create(C.c_void_p(uid.ctypes.data), rank, nranks, device))

    def hist_allreduce(self, ev, d_counts, d_kover, ncontigs, stream=None):
        """counts image + koverCpy of every rank -> the global ones, in place on every rank (async on `stream`)"""
        p = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _check(load_library().mfx_hist_allreduce(self.h, p(d_counts), p(d_kover), ev.nbins, ncontigs, C.c_void_p(stream or 0)))

    def barrier(self, stream=None):
        """every rank has arrived and `stream` has drained"""
        _check(load_library().mfx_comm_barrier(self.h, C.c_void_p(stream or 0)))

    def allgather_overflow(self, ev, cap=1 << 20, stream=None):
        """the far K* bins of ALL ranks as {key, occurrences} pairs, shape (n, 2) (collective; call on every rank when the reduced
        image's novf word is non-zero)"""
        rec = np.zeros(2 * cap, dtype=np.uint64)
        n = C.c_uint64(0)
        _check(load_library().mfx_hist_allgather_overflow(self.h, ev.h, rec.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n),
                                                          C.c_void_p(stream or 0)))
        return rec[:2 * n.value].reshape(-1, 2)

    def exchange_counts(self, send_counts, stream=None):
        """what every rank will send to this one (all-gather of the count rows; synchronises `stream`)"""
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        assert len(sc) == self.nranks
        rc = np.zeros(self.nranks, dtype=np.uint64)
        _check(load_library().mfx_comm_exchange_counts(self.h, sc.ctypes.data_as(C.POINTER(C.c_uint64)), rc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       C.c_void_p(stream or 0)))
        return rc

    def alltoallv(self, d_send, send_counts, d_recv, recv_counts, elem_bytes, stream=None):
        """group r of d_send -> rank r, received groups in source-rank order (device buffers; async on `stream`)"""
        p = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        _check(load_library().mfx_comm_alltoallv(self.h, p(d_send), sc.ctypes.data_as(C.POINTER(C.c_uint64)), p(d_recv),
                                                 rc.ctypes.data_as(C.POINTER(C.c_uint64)), elem_bytes, C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "h", None):
            load_library().mfx_comm_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def hist_parts(evaluators, sequences, contig_ids, ncontigs_total):
    """mfx_hist_run_parts: slot d evaluates its contigs (sequences[d], numbered contig_ids[d] in the whole assembly) on its own index"""
    n = len(evaluators)
    evs = (C.c_void_p * n)(*[e.h for e in evaluators])
    sqs = (C.c_void_p * n)(*[s.h for s in sequences])
    keep = [np.ascontiguousarray(ids, dtype=np.uint32) for ids in contig_ids]
    idp = (C.POINTER(C.c_uint32) * n)(*[k.ctypes.data_as(C.POINTER(C.c_uint32)) for k in keep])
    r = HistResult()
    _check(load_library().mfx_hist_run_parts(evs, sqs, idp, n, int(ncontigs_total), C.byref(r.c)))
    return r


def stream_rates(src, threads, device=0):
    """(GB/s `threads` host threads encode `src` -- a PinnedBuffer view or any contiguous uint8 array -- at, GB/s the device's link moves it at: 0 when
    src is not pinned): the two rates a streamed run's transport is chosen from (mfx_diag_stream_rates)"""
    a = np.ascontiguousarray(src)
    enc, link = C.c_double(0), C.c_double(0)
    _check(load_library().mfx_diag_stream_rates(device, C.c_void_p(a.ctypes.data), a.nbytes, threads, C.byref(enc), C.byref(link)))
    return enc.value, link.value


def gather_rate(table_bytes, device=0):
    """diagnostic: random 128-byte lines per second the device's HBM delivers over a table of that size (mfx_diag_gather_rate)"""
    out = C.c_double(0.0)
    _check(load_library().mfx_diag_gather_rate(device, int(table_bytes), C.byref(out)))
    return out.value


class LoadedVcf:
    """a VCF read and parsed ahead of its run (mfx_vcf_load): no device involved"""

    def __init__(self, path):
        self.h = _need(load_library().mfx_vcf_load(path.encode()))

    def prepare(self, k, mode, names, contigs, comb=15, nosplit=False, debug_path=None):
        """the clusters merged and their allele combinations enumerated and packed ahead of the run (mfx_vcf_prepare: host work, no
        index, no device); Evaluator.variants_loaded on this handle with the same k / comb / nosplit / contigs then starts at the lookups"""
        n = len(contigs)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        arr = (C.c_char_p * n)(*contigs)
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        o = _VarOpts(VARIANT_MODES[mode], comb, 1 if nosplit else 0, debug_path.encode() if debug_path else None)
        _check(load_library().mfx_vcf_prepare(self.h, int(k), nm, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(o)))

    def prepare_path_index(self, k, mode, names, contigs, comb=15, nosplit=False, debug_path=None, max_gb=0.0, device=0, load_factor=0.0):
        """prepare + the path-only index in one pass (mfx_vcf_prepare_path_index): the claimed Index (load the databases next), or None when
        this call set cannot have one (the handle is prepared either way)"""
        n = len(contigs)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        arr = (C.c_char_p * n)(*contigs)
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        o = _VarOpts(VARIANT_MODES[mode], comb, 1 if nosplit else 0, debug_path.encode() if debug_path else None)
        h = C.c_void_p(None)
        _check(load_library().mfx_vcf_prepare_path_index(self.h, int(k), nm, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(o), float(max_gb), int(device),
                                                         float(load_factor), C.byref(h)))
        if not h.value:
            return None
        ix = Index(0, 0, device=device, _handle=h.value)
        ix.k = int(k)
        return ix

    def path_bound(self):
        """k-mer positions of all path text of the prepared call set: the capacity of its path-only index (mfx_vcf_path_bound)"""
        n = C.c_uint64(0)
        _check(load_library().mfx_vcf_path_bound(self.h, C.byref(n)))
        return n.value

    def claim_paths(self, index):
        """the k-mers of every path claimed on a sequence-only index (Index.for_sequence): the PATH-ONLY index of the variant modes
        (mfx_index_claim_paths); the databases then update those k-mers only and Evaluator.variants_loaded on this handle runs on it"""
        n = C.c_uint64(0)
        _check(load_library().mfx_index_claim_paths(index.h, self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            load_library().mfx_vcf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Evaluator:
    """K* parameters bound to an Index; runs -hist / -dump / -completeness."""

    def __init__(self, index, kparams, nbins=0):
        self.index = index
        self.kp = kparams
        self.h = _need(load_library().mfx_eval_create(index.h, C.byref(kparams.c), nbins))

    @property
    def nbins(self):
        return load_library().mfx_eval_nbins(self.h)

    def debug(self, on=True):
        """test hook: -hist launches run the DEBUG instance of the kernel (probe path counters) where one exists"""
        _check(load_library().mfx_eval_debug_enable(self.h, 1 if on else 0))

    def debug_counters(self):
        """how the probe's queries that left the one-load path ended (cleared): first_pass / second_pass (cooperative passes over the home
        line / the next candidate line), side_table (a saturated count field), line_scans (further candidate lines: listed for
        mfx_hist_rest_kernel), side_not_in_two_slots (saturated, not in slot 0 / 1 of its side-table line: listed), ended_per_lane (no
        list, or the list was full: scanned for by the lane itself)"""
        out = np.zeros(8, dtype=np.uint64)
        _check(load_library().mfx_eval_debug_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return {"first_pass": int(out[0]), "second_pass": int(out[1]), "side_table": int(out[2]), "line_scans": int(out[3]),
                "side_not_in_two_slots": int(out[4]), "ended_per_lane": int(out[5])}

    def hist(self, seqs):
        r = HistResult()
        _check(load_library().mfx_hist_run(self.h, seqs.h, C.byref(r.c)))
        return r

    def hist_streamed(self, seqs, host_contigs):
        """-hist with the upload inside (SURVEY 8d's evaluate phase): `seqs` = Sequences.create(lens); host_contigs =
        bytes / numpy uint8 arrays / PinnedBuffer views, one per contig.  Chunked H2D overlapped with the kernel."""
        ptrs, keep = _host_ptrs(host_contigs)
        r = HistResult()
        _check(load_library().mfx_hist_run_streamed(self.h, seqs.h, ptrs, C.byref(r.c)))
        return r

    def hist_streamed_range(self, seqs, host_contigs, tile_begin, tile_end, d_counts, d_kover):
        """one rank's share of a streamed -hist (mfx_hist_run_streamed_range): tiles [tile_begin, tile_end) encoded, uploaded and
        evaluated, ADDED to the caller's device image / koverCpy; returns when the device is done"""
        ptrs, keep = _host_ptrs(host_contigs)
        pc = d_counts.data_ptr() if hasattr(d_counts, "data_ptr") else int(d_counts)
        pk = d_kover.data_ptr() if hasattr(d_kover, "data_ptr") else int(d_kover)
        _check(load_library().mfx_hist_run_streamed_range(self.h, seqs.h, ptrs, tile_begin, tile_end, C.c_void_p(pc), C.c_void_p(pk)))

    def hist_launch(self, seqs, tile_begin, tile_end, d_counts, d_kover, stream=None):
        """Asynchronous accumulate into caller-owned device buffers (torch tensors or raw pointers)."""
        pc = d_counts.data_ptr() if hasattr(d_counts, "data_ptr") else int(d_counts)
        pk = d_kover.data_ptr() if hasattr(d_kover, "data_ptr") else int(d_kover)
        _check(load_library().mfx_hist_launch(self.h, seqs.h, tile_begin, tile_end, C.c_void_p(pc), C.c_void_p(pk),
                                              C.c_void_p(stream or 0)))

    def hist_launch_cyclic(self, seqs, rank, nranks, d_counts, d_kover, block_tiles=256, stream=None):
        """the block-cyclic share of `rank`: blocks of `block_tiles` tiles dealt round-robin to the ranks"""
        p = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _check(load_library().mfx_hist_launch_cyclic(self.h, seqs.h, rank, nranks, block_tiles, p(d_counts), p(d_kover),
                                                     C.c_void_p(stream or 0)))

    def result_from_counts(self, h_counts, kover, ncontigs):
        return result_from_counts(self.nbins, h_counts, kover, ncontigs)

    def take_overflow(self, cap=1 << 20):
        """the K* bins beyond the dense image seen by this evaluator's launches since the last call, as {key (bit 63: `over`; low
        bits: bin index), occurrences} pairs, shape (n, 2), sorted by key; the evaluator's table is emptied"""
        rec = np.zeros(2 * cap, dtype=np.uint64)
        n = C.c_uint64(0)
        _check(load_library().mfx_hist_take_overflow(self.h, rec.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
        return rec[:2 * n.value].reshape(-1, 2)

    def hist_keys_launch(self, d_keys, d_contigs, n, ncontigs, d_counts, d_kover, stream=None):
        """Owner side of the sharded -hist: probe/K*/bin n received canonical k-mers (device buffers)."""
        p = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _check(load_library().mfx_hist_keys_launch(self.h, p(d_keys), p(d_contigs), n, ncontigs, p(d_counts), p(d_kover),
                                                   C.c_void_p(stream or 0)))

    def dump_values(self, seqs, contig, pos_begin, pos_end):
        n = pos_end - pos_begin
        r = np.zeros(max(n, 1), dtype=np.uint32)
        a = np.zeros(max(n, 1), dtype=np.uint32)
        ka, km = C.c_uint64(0), C.c_uint64(0)
        _check(load_library().mfx_dump_values(self.h, seqs.h, contig, pos_begin, pos_end,
                                              r.ctypes.data_as(C.POINTER(C.c_uint32)), a.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              C.byref(ka), C.byref(km)))
        return r[:n], a[:n], ka.value, km.value

    def dump_contig(self, seqs, contig, name, path, append=False):
        ka, km = C.c_uint64(0), C.c_uint64(0)
        _check(load_library().mfx_dump_contig(self.h, seqs.h, contig, name.encode(), path.encode(), 1 if append else 0,
                                              C.byref(ka), C.byref(km)))
        return ka.value, km.value

    def variants(self, mode, vcf_path, names, contigs, out_path, comb=15, nosplit=False, debug_path=None, log_path=None):
        """-filter/-polish/-better/-strict/-loose over all contigs; returns clusters evaluated"""
        n = len(contigs)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        arr = (C.c_char_p * n)(*contigs)
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        o = _VarOpts(VARIANT_MODES[mode], comb, 1 if nosplit else 0, debug_path.encode() if debug_path else None)
        ncl = C.c_uint64(0)
        _check(load_library().mfx_variants_run(self.h, vcf_path.encode(), nm, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                               C.byref(o), out_path.encode(), log_path.encode() if log_path else None, C.byref(ncl)))
        return ncl.value

    def variants_loaded(self, mode, vcf, names, contigs, out_path, comb=15, nosplit=False, debug_path=None, log_path=None):
        """the same on a VCF read ahead of the run (LoadedVcf: mfx_vcf_load, host work only); the handle serves one run"""
        n = len(contigs)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        arr = (C.c_char_p * n)(*contigs)
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        o = _VarOpts(VARIANT_MODES[mode], comb, 1 if nosplit else 0, debug_path.encode() if debug_path else None)
        ncl = C.c_uint64(0)
        _check(load_library().mfx_variants_run_vcf(self.h, vcf.h, nm, arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                                   C.byref(o), out_path.encode(), log_path.encode() if log_path else None, C.byref(ncl)))
        return ncl.value

    def completeness_pieces(self):
        t = np.zeros(64)
        u = np.zeros(64)
        _check(load_library().mfx_completeness_pieces(self.h, t.ctypes.data_as(C.POINTER(C.c_double)), u.ctypes.data_as(C.POINTER(C.c_double))))
        return t, u

    def completeness(self):
        t, u = C.c_double(), C.c_double()
        _check(load_library().mfx_completeness(self.h, C.byref(t), C.byref(u)))
        return t.value, u.value

    def close(self):
        if getattr(self, "h", None):
            load_library().mfx_eval_free(self.h)
            self.h = None

    def __del__(self):
        self.close()
