#!/bin/bash
# usage: tools/build_full_variant.sh <name> "<extra hipcc flags>"  -> gpurun_variants/lib_<name>.so (EVERY translation unit rebuilt with the flags)
set -e
cd "$(dirname "$0")/../infinitam_amd/csrc"
name=$1; extra=$2
obj=/tmp/itm_fullvariant_$name; mkdir -p $obj ../../gpurun_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra"
objs=""
for f in scene alloc integrate visualise visualise_aux tracker viewbuilder io meshing exchange swapping pending; do
  /opt/rocm/bin/hipcc $FL -c $f.hip -o $obj/$f.o &
  objs="$objs $obj/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/lib_$name.so $objs -ldl
echo built lib_$name.so
