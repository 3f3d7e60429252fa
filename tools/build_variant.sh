#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> gpurun_variants/lib<name>.so (A/B kernel builds)
set -e
NAME=$1; shift
SRC=/root/repo/pytorch-lasso_amd/csrc
OUT=/root/repo/variants; mkdir -p $OUT/obj_$NAME
for f in $(cd $SRC && ls *.hip | sed "s/[.]hip\$//"); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 "$@" -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj_$NAME/*.o -o $OUT/lib$NAME.so
echo built $OUT/lib$NAME.so
