#!/bin/bash
# per-kernel times of one `merfin -hist` run at 1 Gb (inputs written by tools/cfg2_cli_timing.py into /tmp/mfx_cfg2):
#   tools/cli_kernel_stats.sh <tag>   ->  gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-cli}
python tools/cfg2_cli_timing.py 1e9 > gpurun_out/${TAG}_timing.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o cli -- $R/merfin_amd/bin/merfin -hist -sequence /tmp/mfx_cfg2/asm.fasta -readmers /tmp/mfx_cfg2/read.mfxk \
  -peak 26 -prob $R/tests/golden/example_lookup_table.txt -output /tmp/mfx_cfg2/outp.hist > $R/gpurun_out/${TAG}_rocprof.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
cut -c1-160 $R/gpurun_out/${TAG}_kernel_stats.csv | head -12
