#!/bin/bash
# A/B PMC comparison of the hist kernel under different index placements.
# usage: pmc_ab.sh <tag> <bases> <env assignments...>
TAG=$1; BASES=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PASS in "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $PASS | cut -d' ' -f1)
  env "$@" timeout 600 rocprofv3 --pmc $PASS --kernel-include-regex "mfx_hist" --output-format csv -d $OUT/$N -o b -- python $REPO/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-pmc --bases $BASES < /dev/null > $OUT/$N.json 2> $OUT/$N.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mfx_hist" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$TAG", {k: sum(v)/len(v) for k, v in sorted(acc.items())})
PY
