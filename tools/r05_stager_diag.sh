#!/bin/bash
# round 5: the host's threads against the CPU quota of the box (cpu.max = 16 cores on the measured ones): `merfin -hist` at 3 Gb from the placed database,
# reader / FASTA / stager thread counts varied, with the cgroup's throttling counters around every run
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs10.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0, placed=True)
print("inputs written", inp["db_bytes"], inp["placed_db_bytes"], inp["read_kmers"], inp["write_s"])
PY
CMDA="$ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist -readmers $DIR/read.placed.mfxk"
stat_() { awk '/nr_throttled|throttled_usec|usage_usec/ {printf "%s ", $2}' /sys/fs/cgroup/cpu.stat; }
sleep 10
{
echo "# cpu.max: $(cat /sys/fs/cgroup/cpu.max)   columns after the wall: usage_usec / nr_throttled / throttled_usec of the cgroup, differences over the run"
tail -1 $OUT/r05_inputs10.log
for spec in "MFX_X=1" "MFX_CLI_STAGE_FIRST=0"; do
  for rep in 1 2 3; do
    sleep 4
    a=$(stat_)
    s=$(date +%s.%N)
    env $spec MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 MFX_CLI_SEQ_TIMING=1 MFX_UPLOAD_TIMING=1 $CMDA 2> $DIR/err.txt
    e=$(date +%s.%N)
    b=$(stat_)
    echo "$spec rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  cpu.stat $(python3 -c "
a='$a'.split(); b='$b'.split()
print(' '.join(str(int(y)-int(x)) for x,y in zip(a,b)))")  $(grep -h -- '-- timing:' $DIR/err.txt | tr '\n' ' ' | cut -c1-200)  md5 $(md5sum < $DIR/o.hist | cut -c1-8)"
    grep -h 'staged build\|-- stager\|read_fasta_parallel\|-- upload\|-- device warm' $DIR/err.txt | cut -c1-330 | sed 's/^/      /'
  done
done
} > $OUT/r05_stager_diag.txt 2>&1
rm -rf $DIR
cat $OUT/r05_stager_diag.txt | cut -c1-420
