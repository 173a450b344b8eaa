#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [source stem, default visualise]  -> gpurun_variants/lib_<name>.so
# Only the named object is rebuilt with the extra flags; the rest is linked from the in-tree build.
set -e
cd "$(dirname "$0")/../infinitam_amd/csrc"
name=$1; extra=$2; stem=${3:-visualise}
obj=/tmp/itm_variant_$name; mkdir -p $obj ../../gpurun_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra"
/opt/rocm/bin/hipcc $FL -c $stem.hip -o $obj/$stem.o
objs=""
for f in scene alloc integrate visualise visualise_aux tracker viewbuilder io meshing exchange swapping pending; do
  if [ $f = $stem ]; then objs="$objs $obj/$f.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/lib_$name.so $objs -ldl
echo built lib_$name.so
