#!/bin/bash
# usage: tools/ab_rates.sh <bases> "<k ...>" <label>=<lib path or "default"> ...  -- tools/hist_rates_by_k.py (sequence-only index) per library, same box
BASES=$1; KS=$2; shift 2
export MFX_RATES_KINDS=seq
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$PWD/$lib; fi
  python tools/hist_rates_by_k.py $BASES $KS 2>&1 | grep "^k=" | sed "s/^/$label  /"
done
