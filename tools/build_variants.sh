#!/bin/bash
# Build experiment variants of libglomseg.so side by side: tools/build_variants.sh name1 "flags1" name2 "flags2" ...
# -> variants_so/libglomseg_<name>.so (git-ignored; selected with GLOMSEG_LIB=...).
cd "$(dirname "$0")/.."
mkdir -p variants_so
pids=()
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    python -m glomeruli_segmentation_amd.build --out variants_so/libglomseg_$name.so -- $flags > variants_so/build_$name.log 2>&1 &
    pids+=($!)
    if [ ${#pids[@]} -ge 3 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la variants_so/*.so
