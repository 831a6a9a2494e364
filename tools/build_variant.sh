#!/bin/bash
# dev tool: tools/build_variant.sh <tag> <src.hip> [hipcc flags...] -- recompiles ONE translation unit
# with extra flags and links it with the other objects of the normal build into
# tools/variants/libtsf_amd_<tag>.so (kernel experiments: several variants per GPU call, see
# tools/variant_bench.py)
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/time_series_spark_amd/_obj
base=$(basename "$src" .hip)
out=$root/tools/variants
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -pthread "$@" \
    -c "$root/time_series_spark_amd/csrc/$base.hip" -o "$out/${base}_$tag.o"
others=$(ls $obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$out/libtsf_amd_$tag.so" $others "$out/${base}_$tag.o" -lz
rm -f "$out/${base}_$tag.o"
echo "built $out/libtsf_amd_$tag.so"
