"""Multi-GPU `merfin -hist` / `-completeness`: one process per GPU, launched with torchrun.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      -m merfin_amd.mgpu -sequence asm.fasta.gz -readmers reads.meryl [-seqmers asm.meryl] \\
      -peak 26 [-prob lookup_table.txt] -output out.hist [-sharded]
  ... -m merfin_amd.mgpu -completeness -readmers reads.meryl -seqmers asm.meryl -peak 26 [-sharded]
  ... -m merfin_amd.mgpu -dump   -sequence asm.fasta -readmers reads.meryl -peak 26 -output out.dump
  ... -m merfin_amd.mgpu -polish -sequence asm.fasta -readmers reads.meryl -peak 26 -vcf calls.vcf -output out
      (-filter / -better / -strict / -loose likewise; -comb N, -nosplit as in merfin)

-dump and the variant modes have per-contig, ordered output: the contigs are split into one contiguous run per
rank (balanced by bases / by VCF records), every rank writes its part, rank 0 concatenates them in rank order.

Every rank reads the inputs, builds the index on its own GPU and evaluates its
share of the sequence tiles; the only collective is the all-reduce of the counts
image (RCCL over xGMI with the default backend "nccl").  With -sharded each rank
keeps only the k-mers it owns (read DBs larger than one GPU, BASELINE config 5)
and the k-mers are exchanged with one all-to-all per chunk of tiles.  Rank 0
writes the histogram and prints the summary, byte-identical to the single-GPU
`merfin -hist` (merfin-histogram.C:140-176).  The reference's multi-process story
is SLURM contig sharding + concatenation (scripts/parallel1/merfin.sh:68-85).

MFX_MGPU_BACKEND=gloo MFX_MGPU_SHARE_GPU=1 rehearse the multi-rank flow on a
single GPU (collectives through host memory; RCCL refuses two ranks per device).
"""
import argparse
import gzip
import os
import sys

import numpy as np


def read_sequences(path):
    """FASTA/FASTQ (plain or .gz): [(ident, bases)], ident = first header token (merfin.C:38, merfin-variants.C:141)"""
    op = gzip.open if path.endswith(".gz") else open
    names, seqs = [], []
    with op(path, "rb") as f:
        data = f.read()
    if not data:
        return names, seqs
    if data[:1] == b">":
        for rec in data.split(b"\n>"):
            head, _, body = rec.partition(b"\n")
            names.append(head.lstrip(b">").split()[0].decode() if head.strip(b">").strip() else "")
            seqs.append(body.replace(b"\n", b"").replace(b"\r", b""))
    else:                                                  # FASTQ, 4-line records
        lines = data.split(b"\n")
        for i in range(0, len(lines) - 1, 4):
            if lines[i].startswith(b"@"):
                names.append(lines[i][1:].split()[0].decode())
                seqs.append(lines[i + 1].strip())
    return names, seqs


VARIANT_MODES = ("filter", "polish", "better", "strict", "loose")


def concat_parts(out_path, parts, skip_header):
    """rank-ordered concatenation; `skip_header`: drop the '#' lines of every part but the first (VCF)"""
    with open(out_path, "wb") as o:
        for i, p in enumerate(parts):
            with open(p, "rb") as f:
                if skip_header and i:
                    for line in f:
                        if not line.startswith(b"#"):
                            o.write(line)
                            break
                    else:
                        continue
                while True:
                    blk = f.read(1 << 24)
                    if not blk:
                        break
                    o.write(blk)
            os.remove(p)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="merfin_amd.mgpu", add_help=True)
    for flag in ("-sequence", "-readmers", "-seqmers", "-prob", "-output"):
        ap.add_argument(flag)
    ap.add_argument("-peak", type=float, default=0.0)
    ap.add_argument("-min", type=int, default=0)
    ap.add_argument("-max", type=int, default=2**64 - 1)
    ap.add_argument("-sharded", action="store_true")
    ap.add_argument("-parts", action="store_true",
                    help="-hist / -dump: every rank indexes and evaluates its own run of contigs on a sequence-only index; "
                         "no k-mer is exchanged, each rank keeps the 1/N of the read database its contigs hit")
    ap.add_argument("-broadcast-index", dest="broadcast_index", action="store_true",
                    help="replicated index: rank 0 reads the k-mer databases and builds, the table is broadcast to the other GPUs")
    ap.add_argument("-completeness", action="store_true")
    ap.add_argument("-dump", action="store_true")
    for mode in VARIANT_MODES:
        ap.add_argument("-" + mode, action="store_true")
    ap.add_argument("-vcf")
    ap.add_argument("-comb", type=int, default=15)
    ap.add_argument("-nosplit", action="store_true")
    ap.add_argument("-chunk-tiles", type=int, default=16384)
    a = ap.parse_args(argv)
    if a.completeness:
        if not (a.readmers and a.peak and (a.seqmers or a.sequence)):
            ap.error("-completeness needs -readmers, -peak and -seqmers (or -sequence)")
    elif not (a.sequence and a.readmers and a.output and a.peak):
        ap.error("-sequence, -readmers, -peak and -output are required")
    vmode = [x for x in VARIANT_MODES if getattr(a, x)]
    if len(vmode) > 1 or (vmode and (a.dump or a.completeness)):
        ap.error("one report type at a time")
    if vmode and not a.vcf:
        ap.error("No variant call input (-vcf) supplied.")
    if (vmode or a.dump) and a.sharded:
        ap.error("-sharded applies to -hist and -completeness")
    if a.parts and (vmode or a.completeness or a.sharded or a.broadcast_index or not a.sequence):
        ap.error("-parts applies to -hist and -dump over -sequence")

    import torch
    import torch.distributed as dist
    import merfin_amd as m
    from merfin_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("MFX_MGPU_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MFX_MGPU_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    log = (lambda *x: print(*x, file=sys.stderr, flush=True)) if rank == 0 else (lambda *x: None)
    # the data-path collectives (all-reduce of the counts image, the all-to-all of a sharded index) are the library's own,
    # on RCCL (csrc/mfx_comm.cpp); torch.distributed stays the control plane (rendezvous, barriers, object gathers)
    comm = None
    if world > 1 and backend == "nccl":
        uid = [m.Comm.unique_id().tobytes() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = m.Comm(np.frombuffer(uid[0], dtype=np.uint8), rank, world, device=local)

    names, seqs = read_sequences(a.sequence) if a.sequence else ([], [])
    rdb = m.db_probe(a.readmers)
    k = rdb["k"]
    n_asm = m.db_probe(a.seqmers)["n_kmers"] if a.seqmers else sum(len(s) for s in seqs)
    cap = rdb["n_kmers"] + n_asm + 1024
    if a.sharded and world > 1:
        cap = int(cap / world * 1.15) + 1024               # owners are hash-balanced
    sq = m.Sequences(seqs, device=local, names=names)
    bcast = a.broadcast_index and world > 1 and not a.sharded
    ix = None
    own_lo = own_hi = 0
    sq_own = None
    if a.parts:
        # the one-process-per-GPU form of the CLI's run_parts (cli/merfin.cpp): a contiguous run of contigs per rank, a
        # sequence-only index of exactly their k-mers (claim), the assembly count of those k-mers over EVERY contig
        # (merfin-globals.C:182-186 counts the whole assembly), the read database update-only.  Nothing is exchanged but
        # the counts image at the end.
        if k > 31:
            ap.error("-parts needs k <= 31 (the sequence-only index)")
        own_lo, own_hi = D.contig_partition([len(s) for s in seqs], world)[rank]
        nb = sum(len(s) for s in seqs[own_lo:own_hi])
        print("-- Part %d of %d: contigs [%d,%d), %d bases." % (rank, world, own_lo, own_hi, nb), file=sys.stderr, flush=True)
        ix = m.Index.for_seq(k, nb + 1024, device=local)
        if own_hi > own_lo:
            sq_own = m.Sequences(seqs[own_lo:own_hi], device=local, names=names[own_lo:own_hi])
            ix.claim_seq(sq_own)
            if a.seqmers:
                ix.load_db(a.seqmers, 1)
            else:
                ix.count_claimed(sq)
            log("-- Loading kmers from '%s' into lookup table." % a.readmers)
            ix.load_db(a.readmers, 0, a.min, a.max)
    elif not bcast or rank == 0:
        ix = m.Index(k, cap, device=local)
        if a.sharded and world > 1:
            ix.set_shard(rank, world)
        log("-- Loading kmers from '%s' into lookup table." % a.readmers)
        ix.load_db(a.readmers, 0, a.min, a.max)
        if a.seqmers:
            log("-- Loading kmers from '%s' into lookup table." % a.seqmers)
            ix.load_db(a.seqmers, 1)
        else:
            ix.count_asm(sq)
    if bcast:
        log("-- Broadcasting the built table to %d GPUs." % world)
        ix = D.broadcast_index(ix, src=0, device=local)
    kp = m.KParams.from_file(a.peak, a.prob) if a.prob else m.KParams(a.peak)
    ev = m.Evaluator(ix, kp)
    if a.completeness:
        # every rank sums over the k-mers it holds; with a replicated index only rank 0 contributes
        log("-- Compute completeness on %d GPU(s)%s." % (world, " (sharded index)" if a.sharded else ""))
        own = (a.sharded and world > 1) or rank == 0
        t64, u64 = ev.completeness_pieces() if own else (np.zeros(64), np.zeros(64))
        total, undr, t64, u64 = D.reduce_completeness(t64, u64, device="cuda" if backend == "nccl" else None)
        if rank == 0:
            for piece in range(64):                        # merfin-completeness.C:119-120, in piece order
                c = 1.0 - u64[piece] / t64[piece] if t64[piece] else float("nan")
                print("thread %2d total %12.2f underc %15.5f completeness %0.8f" % (piece, t64[piece], u64[piece], c), file=sys.stderr)
            print("", file=sys.stderr)
            print("TOTAL readK:   %15.2f" % total, file=sys.stderr)
            print("TOTAL undrcpy:    %15.5f" % undr, file=sys.stderr)
            print("COMPLETENESS:             %0.5f" % (1.0 - undr / total if total else float("nan")), file=sys.stderr)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.dump or vmode:
        # per-contig ordered output: contiguous run of contigs per rank, parts concatenated by rank 0
        if vmode:
            per = dict.fromkeys(names, 0)
            with (gzip.open if a.vcf.endswith(".gz") else open)(a.vcf, "rb") as f:
                for line in f:
                    if line[:1] != b"#":
                        c = line.split(b"\t", 1)[0].decode()
                        if c in per:
                            per[c] += 1
            weights = [per[n] + 1e-9 * len(s) for n, s in zip(names, seqs)]
            out = a.output + (".polish.vcf" if vmode[0] == "polish" else ".filter.vcf")      # merfin-variants.C:324-327
        else:
            weights = [len(s) for s in seqs]
            out = a.output
        lo, hi = D.contig_partition(weights, world)[rank]
        assert not a.parts or (lo, hi) == (own_lo, own_hi)
        part = "%s.part%04d" % (out, rank) if world > 1 else out
        log("-- %s on %d GPU(s): rank 0 takes contigs [%d,%d) of %d." % ("-dump" if a.dump else "-" + vmode[0], world, lo, hi, len(names)))
        if a.dump:
            open(part, "wb").close()
            tot_a = tot_m = 0
            for c in range(lo, hi):
                ka, km = (ev.dump_contig(sq_own, c - lo, names[c], part, append=True) if a.parts else
                          ev.dump_contig(sq, c, names[c], part, append=True))
                tot_a += ka
                tot_m += km
            cnt = torch.tensor([tot_a, tot_m], dtype=torch.int64)
            if world > 1:
                cnt = cnt.cuda() if backend == "nccl" else cnt
                dist.all_reduce(cnt)
            log("K-mers found in the assembly: %d, missing: %d" % (int(cnt[0]), int(cnt[1])))
        else:
            ev.variants(vmode[0], a.vcf, names[lo:hi], seqs[lo:hi], part, comb=a.comb, nosplit=a.nosplit)
        if world > 1:
            dist.barrier()
            if rank == 0:
                concat_parts(out, ["%s.part%04d" % (out, r) for r in range(world)], skip_header=bool(vmode))
            dist.barrier()
            dist.destroy_process_group()
        return
    counts = torch.zeros(m.hist_words(ev.nbins, sq.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ev.take_overflow()                                     # start from an empty overflow list

    def host_exchange(keys, contigs_, send):               # gloo rehearsal only
        sc = torch.tensor([int(x) for x in send], dtype=torch.int64)
        rc = torch.empty(world, dtype=torch.int64)
        dist.all_to_all_single(rc, sc)
        rcl, scl = [int(x) for x in rc.tolist()], [int(x) for x in send]
        n_in = sum(scl)
        rk = torch.empty(sum(rcl), dtype=torch.int64)
        rg = torch.empty(sum(rcl), dtype=torch.int32)
        dist.all_to_all_single(rk, keys[:n_in].cpu(), output_split_sizes=rcl, input_split_sizes=scl)
        dist.all_to_all_single(rg, contigs_[:n_in].cpu(), output_split_sizes=rcl, input_split_sizes=scl)
        return rk.cuda(), rg.cuda()

    def reduce_all():
        if world == 1:
            return
        if comm is not None:
            comm.hist_allreduce(ev, counts, kover, sq.ncontigs, stream=stream)
        elif backend == "nccl":
            D.all_reduce_hist(counts, kover)
        else:
            c, kv = counts.cpu(), kover.cpu()
            D.all_reduce_hist(c, kv)
            counts.copy_(c)
            kover.copy_(kv)

    log("-- Generate histogram of the k* metric to '%s' on %d GPU(s)%s." % (a.output, world, " (sharded index)" if a.sharded else ""))
    if a.parts:
        n_own = own_hi - own_lo
        if n_own:
            mine = torch.zeros(m.hist_words(ev.nbins, n_own), dtype=torch.int64, device="cuda")
            ev.hist_launch(sq_own, 0, sq_own.ntiles, mine, kover, stream=stream)
            torch.cuda.synchronize()
            # this rank's image into the image of the whole assembly: the bins and totals as they are, the per-contig
            # words at the contigs' numbers in the input
            fixed = 2 * ev.nbins + 3
            counts[:fixed] = mine[:fixed]
            counts[fixed + own_lo:fixed + own_hi] = mine[fixed:fixed + n_own]
            counts[fixed + sq.ncontigs + own_lo:fixed + sq.ncontigs + own_hi] = mine[fixed + n_own:fixed + 2 * n_own]
        reduce_all()
    elif a.sharded and world > 1:
        router = m.Router(ix, world, min(a.chunk_tiles, max(1, sq.ntiles)))
        if backend == "nccl":
            D.sharded_hist(ev, router, sq, rank, world, counts, kover, stream=stream, comm=comm)
        else:
            T = sq.ntiles
            lo, hi = D.shard(T, rank, world)
            per = router.max_tiles
            keys = torch.empty(per * m.TILE, dtype=torch.int64, device="cuda")
            ctg = torch.empty(per * m.TILE, dtype=torch.int32, device="cuda")
            for r in range((-(-T // world) + per - 1) // per):
                tb = min(hi, lo + r * per)
                te = min(hi, tb + per)
                send = router.route(sq, tb, te, ev.nbins, counts, keys, ctg, stream=stream)
                rk, rg = host_exchange(keys, ctg, send)
                if rk.numel():
                    ev.hist_keys_launch(rk, rg, rk.numel(), sq.ncontigs, counts, kover, stream=stream)
                torch.cuda.synchronize()
            reduce_all()
    else:
        ev.hist_launch_cyclic(sq, rank, world, counts, kover, stream=stream)      # block-cyclic share of the tiles
        torch.cuda.synchronize()
        reduce_all()
    torch.cuda.synchronize()

    # K* bins beyond the dense image (the reference's arrays are unbounded, merfin-histogram.C:74,87) stay in each
    # evaluator's overflow list: the reduced novf word is the same on every rank, so all ranks agree on gathering them
    h = counts.cpu().numpy().view(np.uint64)
    overflow = np.zeros(0, dtype=np.uint64)
    if int(h[2 * ev.nbins + 2]):
        overflow = ev.take_overflow()
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, overflow)
            overflow = np.concatenate(parts)
    if rank == 0:
        res = m.result_from_counts(ev.nbins, h, float(kover.item()), sq.ncontigs).add_overflow(overflow)
        cum = 0
        for c, nm in enumerate(names):                     # outputHistogram's per-sequence line, input order
            cum += int(res.contig_kmissing()[c])
            print("%s\t%d\t%d\t%d\t%.2f" % (nm, res.contig_kmissing()[c], cum, res.contig_kasm()[c],
                                            m.histoQV(float(res.contig_kmissing()[c]), float(res.contig_kasm()[c]), k)), file=sys.stderr)
        res.report(k, a.output, "-")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
