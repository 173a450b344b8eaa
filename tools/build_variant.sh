#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> gpurun_variants/lib_<name>.so
# Only the visualisation objects are rebuilt with the extra flags; the rest is linked from the in-tree build.
set -e
cd "$(dirname "$0")/../infinitam_amd/csrc"
name=$1; extra=$2
obj=/tmp/itm_variant_$name; mkdir -p $obj ../../gpurun_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra"
/opt/rocm/bin/hipcc $FL -c visualise.hip -o $obj/visualise.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/lib_$name.so scene.o alloc.o integrate.o $obj/visualise.o visualise_aux.o tracker.o viewbuilder.o io.o meshing.o
echo built lib_$name.so
