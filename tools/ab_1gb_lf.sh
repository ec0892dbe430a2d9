for lf in 0.25 0.06; do
for spec in lane=default coop=tools/_build/ab/lib_coop.so; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$PWD/$lib; fi
  MFX_LOAD_FACTOR=$lf python bench.py --bases 1e9 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-streamed 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$label lf $lf', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], '%.1f GB' % d['config']['index_gb'], d['config']['kmissing'])"
done; done
