#!/bin/bash
# usage: tools/ab_sweep.sh <variant> [<variant> ...]   (on the GPU box) -- bench_sweep.py of the product build and of
# variants/liblasso_<variant>.so, interleaved three times (box-to-box differences exceed most A/B differences)
for i in 1 2 3; do
  python tools/bench_sweep.py | cut -c1-60
  for v in "$@"; do python tools/bench_sweep.py --lib variants/liblasso_$v.so | cut -c1-80; done
done
