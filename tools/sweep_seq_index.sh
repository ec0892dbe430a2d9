run() { # label env...
  label=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-streamed > gpurun_out/sw_$label.json 2> gpurun_out/sw_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sw_$label.json"))
print("$label", "%.2f G k-mers/s" % (d["value"]/1e9), "%.2f ms" % d["ms_per_step"], "%.1f GB" % d["config"]["index_gb"], d["config"]["kmissing"], d["config"]["hist_sum_check"])
PY
}
run w4_lf20 MFX_LOAD_FACTOR=0.20
run w4_lf30 MFX_LOAD_FACTOR=0.30
run w5_lf25 MFX_MZ_W=5
run w5_lf20 MFX_MZ_W=5 MFX_LOAD_FACTOR=0.20
run w3_lf25 MFX_MZ_W=3
