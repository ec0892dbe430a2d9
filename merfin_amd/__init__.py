"""merfin_amd -- MI355X-native k-mer multiplicity evaluator (merfin's -hist /
-dump / -completeness evaluation path on gfx950).

This package is a thin ctypes binding of the C ABI in include/merfin_amd.h
(libmerfin_amd.so, hand-written HIP).  It exists for the parity tests and
bench.py; the product host program is the C++ `merfin` CLI in merfin_amd/cli.
There is no CPU fallback: importing works anywhere, but creating an index
without the HIP library or without a GPU raises.
"""
from .binding import (  # noqa: F401
    MfxError, E_NONCANON, Index, Sequences, Evaluator, Router, HistResult, KParams,
    lib_path, load_library, device_count, device_warm, getK, getKmetric, histoQV, hist_words, result_from_counts,
    TILE, db_probe, db_write_flat, db_convert, PinnedBuffer, Comm, hist_multi, hist_sharded, load_db_multi, dump_values_sharded, dump_contig_sharded, variants_sharded, pack_bases, gather_rate, stream_rates, hist_parts, LoadedVcf, hist_streamed_multi, stream_share, DbStage, db_convert_placed, db_place_keys, db_write_flat_placed,
)

__all__ = ["MfxError", "E_NONCANON", "Index", "Sequences", "Evaluator", "Router", "HistResult", "KParams", "lib_path",
           "load_library", "device_count", "device_warm", "getK", "getKmetric", "histoQV", "hist_words", "result_from_counts", "TILE", "db_probe", "db_write_flat", "db_convert",
           "PinnedBuffer", "Comm", "hist_multi", "hist_sharded", "load_db_multi", "dump_values_sharded", "dump_contig_sharded", "variants_sharded", "pack_bases", "gather_rate", "stream_rates", "hist_parts", "LoadedVcf", "hist_streamed_multi", "stream_share", "DbStage", "db_convert_placed", "db_place_keys", "db_write_flat_placed"]
