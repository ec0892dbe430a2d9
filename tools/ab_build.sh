#!/bin/bash
# Kernel A/B builds: tools/ab_build.sh <name> <extra hipcc flags...>  ->  tools/_build/ab/lib_<name>.so (the library of the
# current sources with the kernel files compiled under the extra flags); select it at run time with MFX_LIB=<path>.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../merfin_amd/csrc"
OUT=../../tools/_build/ab
mkdir -p $OUT/$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
for f in mfx_kernels mfx_wide; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $OUT/$NAME/$f.o; done
/opt/rocm/bin/hipcc $FLAGS -shared -o $OUT/lib_$NAME.so $OUT/$NAME/mfx_kernels.o $OUT/$NAME/mfx_wide.o ../_build/mfx_sort.o ../_build/mfx_api.o ../_build/mfx_db.o \
  ../_build/mfx_variants.o ../_build/mfx_comm.o ../_build/mfx_pack.o -L/opt/rocm/lib -lrccl
echo built $OUT/lib_$NAME.so
