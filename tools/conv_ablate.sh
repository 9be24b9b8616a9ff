#!/bin/bash
# Ablation study of conv3x3_f16x3 (the LDS-tiled kernel; the dominant one until round 3): which part of the kernel does its time
# belong to?  Run with AIPT_F16R_MINPIX=100000000 so that the big levels use it (ablation builds read that variable; the product
# library reads no environment variable for its kernel selection).
#   build (here, no GPU):   tools/conv_ablate.sh build      -> ab_ab_variants/libaiptd_ablate.so (-DAIPT_CONV_ABLATE; ab_variants/ travels to the GPU box)
#   run (GPU box, gpurun):  tools/conv_ablate.sh run        -> gpurun_out/ablate/*.txt (per-layer tables of bench.py --layers)
# AIPT_CONV_ABLATE is a bit mask: 1 no LDS reads + MFMAs, 2 no activation loads, 4 no weight loads, 8 no transform + LDS writes,
# 16 no output stores, 32 no BN sums, 64 LDS reads without MFMAs, 128 no barriers in the chunk loop.  Results are wrong by design.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
C=ai_path_tracer_denoiser_amd/csrc
if [ "${1:-}" = build ]; then
    mkdir -p ab_variants
    make -C $C >/dev/null 2>&1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fhip-fp32-correctly-rounded-divide-sqrt \
        -Xclang -target-feature -Xclang -packed-fp32-ops -DAIPT_CONV_ABLATE -c $C/denoise.hip -o /tmp/denoise_ablate.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_variants/libaiptd_ablate.so $C/abi.o $C/trace.o /tmp/denoise_ablate.o $C/scene.o $C/bvh.o $C/comm.o -ldl
    ls -la ab_variants/libaiptd_ablate.so
    exit 0
fi
O=gpurun_out/ablate; mkdir -p $O
python bench.py --no-cpu-baseline --layers --steps 16 --warmup 4 > /dev/null 2> $O/product.txt
for m in ${ABLATE_MASKS:-0 1 64 2 4 6 8 16 32 128 7 9 15 17 25 31}; do
    AIPT_LIB=$PWD/ab_variants/libaiptd_ablate.so AIPT_CONV_ABLATE=$m AIPT_BENCH_NO_VALIDATE=1 python bench.py --no-cpu-baseline --layers --steps 16 --warmup 4 > /dev/null 2> $O/m$m.txt
    echo "== mask $m"; grep -E "enc1.l2a|enc1.l2b|enc2.l2a|dec2.c1|conv total" $O/m$m.txt | cut -c1-110
done
echo "== product"; grep -E "enc1.l2a|enc1.l2b|enc2.l2a|dec2.c1|conv total" $O/product.txt | cut -c1-110
