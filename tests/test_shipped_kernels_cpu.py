"""The conv kernel instantiations libaiptd.so ships are a pinned, tested set (instantiations reachable only
through environment variables shipped untested).  CPU: the device symbol table equals EXPECTED; -m gpu
(tests/test_gpu_denoise_kernels.py) runs a parity case on every one of them.  No getenv() steers the denoiser's arithmetic."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ai_path_tracer_denoiser_amd", "libaiptd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

EXPECTED = {
    # register-staged persistent kernel (the levels of >= 200 000 pixels): fp32 / fp16 weights, C4 / planar input
    "conv3x3_f16x3r<false,12,3,false,4,false>", "conv3x3_f16x3r<true,12,3,false,4,false>",
    "conv3x3_f16x3r<false,8,3,false,4,true>", "conv3x3_f16x3r<true,8,3,false,4,true>",
    # LDS-tiled split-fp16 kernel: 8-row and 4-row tiles, planar input, fp16 weights
    "conv3x3_f16x3<1,8,false,false,1>", "conv3x3_f16x3<1,8,false,true,1>", "conv3x3_f16x3<1,8,true,false,1>", "conv3x3_f16x3<1,8,true,true,1>",
    "conv3x3_f16x3<1,4,false,false,1>", "conv3x3_f16x3<1,4,false,true,1>",
    # ... 4-row tiles with three waves per row, one per tap row (the default of the small levels)
    "conv3x3_f16x3<1,4,false,false,3>", "conv3x3_f16x3<1,4,false,true,3>",
    # exact f32 MFMA kernel (AIPT_DN_IMPL_MFMA, and the fallback of levels beyond the fp16 operand range)
    "conv3x3_mfma<2,2,1>", "conv3x3_mfma<2,2,2>", "conv3x3_mfma<2,2,3>", "conv3x3_mfma<1,2,1>", "conv3x3_mfma<1,2,2>", "conv3x3_mfma<1,2,3>",
    "conv3x3_mfma<1,1,1>",
    # output layers and the on-GPU cross-check kernel
    "conv3x3_quad<3,3,true>", "conv3x3_quad<3,3,false>", "conv3x3_valu",
}


def shipped_conv_kernels(tmp="/tmp/aiptd_shipped_kernels"):
    os.makedirs(tmp, exist_ok=True)
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", LIB, os.path.join(tmp, "ignored.so")])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    offs = [m.start() for m in re.finditer(re.escape(magic), data)]
    names = set()
    for i, o in enumerate(offs):
        end = offs[i + 1] if i + 1 < len(offs) else len(data)
        b, co = os.path.join(tmp, f"bundle{i}.bin"), os.path.join(tmp, f"dev{i}.co")
        open(b, "wb").write(data[o:end])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        out = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-C", "-W", co], capture_output=True, text=True, check=True).stdout
        for ln in out.splitlines():
            m = re.search(r"\bFUNC\b.*?\b(?:void )?aipt::(conv3x3_\w+(?:<[^>]*>)?)\(", ln)
            if m:
                names.add(m.group(1).replace(" ", ""))
    return names


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_shipped_conv_instantiations_are_the_pinned_set():
    got = shipped_conv_kernels()
    assert got == EXPECTED, f"unexpected: {sorted(got - EXPECTED)}; missing: {sorted(EXPECTED - got)}"


def test_no_environment_variable_steers_the_denoiser():
    """getenv() may appear in csrc/denoise.hip only inside #ifdef AIPT_DEBUG_HOOKS / AIPT_CONV_ABLATE blocks (debug builds)."""
    src = open(os.path.join(ROOT, "ai_path_tracer_denoiser_amd", "csrc", "denoise.hip")).read().splitlines()
    depth_dbg, stack = 0, []
    for n, ln in enumerate(src, 1):
        t = ln.strip()
        if t.startswith("#if"):
            dbg = "AIPT_DEBUG_HOOKS" in t or "AIPT_CONV_ABLATE" in t or "AIPT_CONV_PHASES" in t
            stack.append(dbg)
            depth_dbg += dbg
        elif t.startswith("#endif"):
            depth_dbg -= stack.pop()
        elif t.startswith("#else") and stack and stack[-1]:
            stack[-1] = False
            depth_dbg -= 1
        elif "getenv(" in ln and not t.startswith("//"):
            assert depth_dbg > 0, f"csrc/denoise.hip:{n}: getenv outside a debug-build block: {t}"
