#!/bin/bash
# developer helper: build a variant of libsmalfit.so with extra -D flags.  usage: tools/build_variant.sh NAME -DX=1 ...
# run a tool against it with SMALFIT_LIB=smalify_amd/_variants/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p smalify_amd/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared "$@" smalify_amd/csrc/smalfit_kernels.hip -o smalify_amd/_variants/$name.so
echo smalify_amd/_variants/$name.so
