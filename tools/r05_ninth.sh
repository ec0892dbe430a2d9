#!/bin/bash
# round 5: how the process leaves (MFX_CLI_QUICK_EXIT) at 3 Gb, and config 4 through the CLI without the record-by-record free of the VCF
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs11.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0, placed=True)
print("inputs written", inp["db_bytes"], inp["placed_db_bytes"], inp["read_kmers"], inp["write_s"])
PY
CMDA="$ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist -readmers $DIR/read.placed.mfxk"
sleep 8
{
echo "# merfin -hist at 3 Gb (placed database): MFX_CLI_QUICK_EXIT 0 / 1, runs spaced by 4 s; MFX_CLI_TIMING=3 stamps: main / devices / end (epoch seconds) next to the parent's spawn and reap times"
for spec in "MFX_CLI_QUICK_EXIT=0" "MFX_CLI_QUICK_EXIT=1" "MFX_CLI_QUICK_EXIT=0" "MFX_CLI_QUICK_EXIT=1"; do
  for rep in 1 2 3; do
    sleep 4
    python3 - "$spec" $CMDA <<'PY'
import os, subprocess, sys, time
spec = sys.argv[1]; cmd = sys.argv[2:]
env = dict(os.environ, MFX_CLI_TIMING="3"); k, v = spec.split("="); env[k] = v
t0 = time.time(); r = subprocess.run(cmd, env=env, capture_output=True, text=True); t1 = time.time()
st = [l for l in r.stderr.splitlines() if "stamps" in l]
end = float(st[0].split("end")[1]) if st else 0.0
main = float(st[0].split("main")[1].split()[0]) if st else 0.0
print("%s wall %.3f s = spawn -> main %.3f + main -> end %.3f + end -> reaped %.3f   rc %d" % (spec, t1 - t0, main - t0, end - main, t1 - end, r.returncode))
PY
  done
done
} > $OUT/r05_exit_ab.txt 2>&1
rm -rf $DIR
cat $OUT/r05_exit_ab.txt
( MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_nofree.txt
rm -rf /dev/shm/mfx_cfg4
grep "SLOTS=1\|mfx_variants\]\|timing:" $OUT/r05_cfg4_cli_nofree.txt | tail -12
