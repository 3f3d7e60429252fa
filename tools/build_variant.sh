#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]   ->  variants/liblasso_<name>.so
# (an A/B build of ONE translation unit linked against the other objects of the product build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
C=$ROOT/pytorch-lasso_amd/csrc
make -s -C $C
mkdir -p $ROOT/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $C/$SRC -o $ROOT/variants/${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/${SRC%.hip}.o") $ROOT/variants/${NAME}.o -o $ROOT/variants/liblasso_${NAME}.so
echo built $ROOT/variants/liblasso_${NAME}.so
