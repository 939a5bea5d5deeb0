"""The device code INSIDE the built libairband_hip.so (not a recompilation of the sources): every translation unit's gfx950 code object is taken out of the
library's .hip_fatbin section and disassembled.  Pinned: no packed-f32 vector instruction anywhere (rtlsdr-airband_amd/_build.py DEVICE_FLAGS,
profiles/r05_event_hunt.md section 4: lanes 48 - 63 of such an instruction's result go wrong while another process shares the GPU), and -- so that the check
is not vacuous -- the kernels the product launches are all in there, the matrix-core ones with their MFMA instructions."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _tool(name):
    p = os.path.join(LLVM, name)
    if not os.path.exists(p):
        p = shutil.which(name)
    if not p:
        pytest.skip("no " + name)
    return p


@pytest.fixture(scope="module")
def device_code(built, tmp_path_factory):
    """[(kernel symbols, disassembly lines)] per translation unit of the shipped library"""
    d = tmp_path_factory.mktemp("fatbin")
    fat = str(d / "fat.bin")
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rtlsdr-airband_amd", "libairband_hip.so")  # the product library, whatever AIRBAND_HIP_LIB says
    subprocess.run([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.devnull], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    units = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, co = str(d / ("bundle%d.bin" % n)), str(d / ("unit%d.co" % n))
        open(piece, "wb").write(blob[a:b])
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + piece, "--output=" + co], check=True)
        text = subprocess.run([_tool("llvm-objdump"), "-d", co], check=True, stdout=subprocess.PIPE, text=True).stdout
        kernels = re.findall(r"^[0-9a-f]+ <(\w+)>:", text, re.M)
        units.append((kernels, text.split("\n")))
    return units


def test_the_shipped_kernels_hold_no_packed_f32_instruction(device_code):
    names = [k for ks, _ in device_code for k in ks]
    for want in ("channelizer_dft_kernel", "channelizer_f32_kernel", "channelizer_fft8_kernel", "channelizer_fft_kernel", "demod_kernel", "tone_kernel", "back_kernel",
                 "mix_runs_kernel", "afc_kernel", "emit_iq_kernel"):
        assert any(want in n for n in names), want
    n_instr = 0
    for _, lines in device_code:
        code = [l for l in lines if re.match(r"^\s+[sv]_\w+|^\s+(ds|global|buffer|flat|scratch)_\w+", l)]
        n_instr += len(code)
        bad = [l.strip() for l in code if re.match(r"^\s+v_pk_(mul|fma|add)_f32", l)]
        assert not bad, bad[:3]
    assert n_instr > 100000  # the whole library was looked at (~4e5 instructions)
    assert sum(1 for _, lines in device_code for l in lines if "v_mfma_i32_16x16x64_i8" in l) > 1000
    assert sum(1 for _, lines in device_code for l in lines if "v_mfma_f32_16x16x4_f32" in l) > 100
