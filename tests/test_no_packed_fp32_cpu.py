"""The shipped library contains no packed-fp32 VALU instruction.

On gfx950 a v_pk_*_f32 instruction can return wrong values in lanes 48..63 of its wave while another wave on the same SIMD issues
fp16 / bf16 MFMAs with gaps between them (tools/coresidency/pk_f32_mfma_erratum.hip; DESIGN.md "Root cause of the co-residency
issue") -- which is what our own split-fp16 conv kernel does beside every other kernel of the library.  csrc/Makefile switches
the `packed-fp32-ops` target feature off; this test disassembles every gfx950 code object inside libaiptd.so and holds it to
that (it runs on the CPU: llvm-objdump reads the fat binary)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ai_path_tracer_denoiser_amd", "libaiptd.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _device_disassembly(tmp_path):
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", LIB, str(tmp_path / "ignored.so")])
    data = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    offs = [m.start() for m in re.finditer(re.escape(magic), data)]
    assert offs, "no offload bundle in libaiptd.so"
    out = []
    for i, o in enumerate(offs):
        end = offs[i + 1] if i + 1 < len(offs) else len(data)
        b = tmp_path / f"bundle{i}.bin"
        b.write_bytes(data[o:end])
        co = tmp_path / f"dev{i}.co"
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", str(co)], capture_output=True, text=True, check=True).stdout)
    return out


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_library_has_no_packed_fp32_instructions(tmp_path):
    dis = _device_disassembly(tmp_path)
    total = sum(d.count("\n") for d in dis)
    mfma = sum(len(re.findall(r"\bv_mfma_", d)) for d in dis)
    assert total > 50000 and mfma > 500, "disassembly looks empty: the check would be vacuous"
    packed = [ln.strip() for d in dis for ln in d.splitlines() if re.search(r"\bv_pk_\w+_f32\b", ln)]
    assert not packed, f"{len(packed)} packed-fp32 instructions in libaiptd.so, e.g. {packed[:3]}"
