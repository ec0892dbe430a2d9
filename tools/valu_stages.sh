#!/bin/bash
# Where the hist kernel's instructions go: the kernel compiled with stages stubbed out (-DMFX_STUB=1 no K*/histogram,
# =2 no table lookup, =3 neither; the libraries are built into _stub/ by hand, see profiles/r02_valu_stages.txt), each run
# under one rocprofv3 PMC pass on the 1 Gb workload; instruction counts per launch, differences = per-stage cost.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/valu_stages
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cp $REPO/merfin_amd/libmerfin_amd.so /tmp/lib_full.so
for V in full stub1 stub2 stub3 plain; do
  case $V in
    full|plain) cp /tmp/lib_full.so $REPO/merfin_amd/libmerfin_amd.so ;;
    *) cp $REPO/_stub/lib_$V.so $REPO/merfin_amd/libmerfin_amd.so ;;
  esac
  E=""; [ $V = plain ] && E="MFX_HOME_MODE=plain"
  env $E timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex "mfx_hist_kernel" --output-format csv -d $OUT/$V -o b -- python $REPO/bench.py --pmc-child --bases 1000000000 > $OUT/$V.log 2>&1
done
cp /tmp/lib_full.so $REPO/merfin_amd/libmerfin_amd.so
python - <<PY
import csv, glob, collections
for v in ("full", "stub1", "stub2", "stub3", "plain"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/b_counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "mfx_hist_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x)/len(x)) for k, x in sorted(acc.items())}, len(next(iter(acc.values()), [])))
PY
