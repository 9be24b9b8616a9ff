// ref_thrust_kats.cpp -- known answers for the RNG the reference takes from Thrust (test infrastructure).
//
// The reference draws from thrust::default_random_engine (= minstd_rand) through thrust::uniform_real_distribution<float>
// (Inference/src/pathtrace.cu:52-56,167-169; interactions.h:16,22,25,196).  Thrust is a third-party, un-vendored dependency
// of the reference (CUDA-toolkit Thrust, version not pinned in-tree); the same published source ships in this image as
// rocThrust (/opt/rocm/include/thrust/random/**, ROCm 7.2.0).  This driver compiles THAT library host-only
// (hipcc --offload-host-only; no GPU, no stand-in headers) and evaluates it on seeds read from stdin, so the oracle's
// orc_lcg_next / orc_u01 restatement is pinned to the library's own arithmetic rather than to a reading of it.
// Recipe: oracle/Makefile target `ref` -> oracle/_ref/thrust_kats.  Usage: thrust_kats <n>: n uint32 seeds on stdin;
// per seed 8 words on stdout: the engine's first 3 raw outputs, then 3 draws of U(0,1) and 2 draws of U(-0.5,0.5)
// (float bit patterns) from fresh engines with the same seed.
#include <thrust/random.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const int n = atoi(argv[1]);
    std::vector<uint32_t> seeds(n), out((size_t)n * 8);
    if (fread(seeds.data(), 4, n, stdin) != (size_t)n) return 1;
    for (int k = 0; k < n; k++) {
        uint32_t* o = &out[(size_t)k * 8];
        thrust::default_random_engine e0(seeds[k]);
        for (int j = 0; j < 3; j++) o[j] = (uint32_t)e0();
        thrust::default_random_engine e1(seeds[k]);
        thrust::uniform_real_distribution<float> u01(0, 1);
        for (int j = 0; j < 3; j++) { const float f = u01(e1); memcpy(&o[3 + j], &f, 4); }
        thrust::default_random_engine e2(seeds[k]);
        thrust::uniform_real_distribution<float> uh(-0.5, 0.5);      // pathtrace.cu:169
        for (int j = 0; j < 2; j++) { const float f = uh(e2); memcpy(&o[6 + j], &f, 4); }
    }
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
