#!/bin/bash
# Builds an experiment variant of the solver library next to the product one: tools/build_variant.sh <name> <extra hipcc flags...>
# -> chrono-wisc_amd/csrc/variants/libchsph_mi_<name>.so (git-ignored, travels to the GPU box); select it with CHSPH_LIB=<path>.
set -e
cd "$(dirname "$0")/../chrono-wisc_amd/csrc"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-inline-asm "$@" -shared -o variants/libchsph_mi_$name.so sph_solver.hip -ldl
echo variants/libchsph_mi_$name.so
