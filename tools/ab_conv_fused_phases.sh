#!/bin/bash
# usage (here, CPU): tools/ab_conv_fused_phases.sh build   -> variants/liblasso_cfabl_<mask>.so for the masks below
#        (GPU box):  tools/ab_conv_fused_phases.sh run     -> per-iteration time of each on the first bench geometry
# Timing ablations of conv_fused.hip (results invalid): LASSO_CF_ABL bits 1 = no synthesis MFMAs, 2 = no overlap-add,
# 4 = no gradient MFMAs, 8 = no epilogue arithmetic, 16 = no z / y traffic in the gradient phase.
MASKS="1 2 3 4 8 16 28 31"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  for m in $MASKS; do bash $ROOT/tools/build_variant.sh cfabl_$m conv_fused.hip -DLASSO_CF_ABL=$m > /dev/null 2>&1 || echo "build $m failed"; done
else
  echo "product"; python $ROOT/tools/bench_conv_fused.py --first
  for m in $MASKS; do echo "LASSO_CF_ABL=$m"; python $ROOT/tools/bench_conv_fused.py --lib $ROOT/variants/liblasso_cfabl_$m.so --first; done
fi
