#!/bin/bash
# round 5: page cache -> HBM microbenchmark; the prepared variant run (tests, config 4 through the CLI); the e2e leg with the bench process holding no HBM
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/r05_hostreg.sh
( timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_cli.py -x -q 2>&1 | tail -5 ) > $OUT/r05_eighth_tests.txt
tail -3 $OUT/r05_eighth_tests.txt
( MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1 MFX_CFG4_AHEAD_AB=1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_ahead.txt
rm -rf /dev/shm/mfx_cfg4
grep "wall=\|mfx_variants\]\|timing:" $OUT/r05_cfg4_cli_ahead.txt | tail -24
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-k31 --no-full-index --no-streamed > $OUT/r05_bench_e2e_only.json 2> $OUT/r05_bench_e2e_only.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_e2e_only.json"))
e = d["e2e"]
for k in ("walls_s", "walls_back_to_back_s", "create_table_back_to_back_s", "bench_process_hbm_reserved_gb", "placed"):
    print(k, e.get(k))
PY
