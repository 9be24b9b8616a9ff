#!/bin/bash
# A/B builds: tools/build_variant.sh NAME 'sed-expr-for-trace.hip' 'sed-expr-for-internal.h' 'sed-expr-for-denoise.hip' [EXTRA hipcc flags]
#   -> ab_variants/libaiptd_NAME.so (travels to the GPU box; select with AIPT_LIB=ab_variants/libaiptd_NAME.so)
set -e
name=$1; st=${2:-}; si=${3:-}; sd=${4:-}; extra=${5:-}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/ai_path_tracer_denoiser_amd/csrc
w=/tmp/variant_$name; rm -rf $w; mkdir -p $w $root/ab_variants
cp $src/*.cpp $src/*.hip $src/internal.h $w/
sed -i "s#\"../../include/aiptd.h\"#\"$root/include/aiptd.h\"#" $w/internal.h
[ -n "$st" ] && sed -i "$st" $w/trace.hip
[ -n "$si" ] && sed -i "$si" $w/internal.h
[ -n "$sd" ] && sed -i "$sd" $w/denoise.hip
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops $extra"
cd $w
( /opt/rocm/bin/hipcc $F -x hip -c abi.cpp -o abi.o & /opt/rocm/bin/hipcc $F -ffp-contract=off -c trace.hip -o trace.o & /opt/rocm/bin/hipcc $F -c denoise.hip -o denoise.o & 
  /opt/rocm/bin/hipcc $F -ffp-contract=off -x hip -c scene.cpp -o scene.o & /opt/rocm/bin/hipcc $F -ffp-contract=off -x hip -c bvh.cpp -o bvh.o & /opt/rocm/bin/hipcc $F -x hip -c comm.cpp -o comm.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ab_variants/libaiptd_$name.so abi.o trace.o denoise.o scene.o bvh.o comm.o -ldl
ls -la $root/ab_variants/libaiptd_$name.so
