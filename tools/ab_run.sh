#!/bin/bash
# usage: tools/ab_run.sh <label>=<lib path or "default"> ...   -- 3 Gb -hist bench per library, same box, back to back (twice)
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$PWD/$lib; fi
  python bench.py --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-streamed --no-e2e --no-full-index > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$label.json"))
print("$label rep$rep", "%.2f G k-mers/s" % (d["value"]/1e9), "%.3f ms" % d["ms_per_step"], "kernel %.3f ms" % d["roofline"]["kernel_ms"], d["config"]["kmissing"], d["config"]["koverCpy"], d["config"]["hist_sum_check"])
PY
done
done
