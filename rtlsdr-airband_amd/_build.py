"""Builds libairband_hip.so in-tree (csrc/ -> rtlsdr-airband_amd/libairband_hip.so) with hipcc for gfx950.

Host-only parameter derivation (params.cpp) is compiled with g++ so its libm / complex arithmetic is the
platform's; kernels and the C-ABI driver are compiled with hipcc.  demod.hip gets -ffp-contract=off and IEEE
divide/sqrt: its arithmetic has to match the reference's scalar float code bit for bit.
"""
from __future__ import annotations

import os
import subprocess
import sys

import hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARCH = "gfx950"
# Kernel experiments (AIRBAND_EXTRA_DEFINES="-DAB_..." [AIRBAND_BUILD_TAG=name]) never touch the product library: they get their own
# object directory and their own file name (libairband_hip_exp_<tag>.so, loaded with AIRBAND_HIP_LIB=...), and every library reports
# the defines it was compiled with through airband_hip_build_info() -- bench.py prints that string into its JSON line.
EXTRA = os.environ.get("AIRBAND_EXTRA_DEFINES", "").split()
TAG = os.environ.get("AIRBAND_BUILD_TAG") or (hashlib.sha1(" ".join(EXTRA).encode()).hexdigest()[:8] if EXTRA else "")
OBJ = os.path.join(HERE, "build" + ("_exp_" + TAG if EXTRA else ""))
LIB = os.path.join(HERE, "libairband_hip" + ("_exp_" + TAG if EXTRA else "") + ".so")

# No packed-f32 vector instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in any kernel of this library.  Round 5 (profiles/r05_event_hunt.md): while ANOTHER
# PROCESS runs long launches on the same GPU, a packed-f32 instruction now and then leaves lanes 48 - 63 of its result wrong -- the notch filter of the CTCSS
# chain's back kernel (28 % of 1 024-dongle handles with four processes on a GPU), the butterflies of the wavefront FFT's exchange kernel (6 - 11 % of the
# fuzz's runs).  Built without them: 0 of 2 487 handles, 0 of 724 runs on the same loads -- and stage 2 is 0.34 ms FASTER (the compiler's pairing cost more
# moves than it saved).  The host pass of hipcc prints "not a recognized feature" for the flag and ignores it.  An experiment build gets them back with
# AIRBAND_EXTRA_DEFINES="-Xclang -target-feature -Xclang +packed-fp32-ops".
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]

HIP_SOURCES = {
    "channelizer_fft.hip": ["-O3"],
    "channelizer_dft.hip": ["-O3"],
    "channelizer_f32.hip": ["-O3"],
    "misc_kernels.hip": ["-O3", "-ffp-contract=off"],  # mixer sums: the reference's multiply-then-add, no FMA
    "demod.hip": ["-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "airband_hip.cpp": ["-O2", "-x", "hip"],
}
HOST_SOURCES = {"params.cpp": ["-O2", "-ffp-contract=off", "-fno-fast-math"]}


def _newer(src: str, dst: str) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src, os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "airband_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r.stdout


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FORBIDDEN = r"^\s+v_pk_(mul|fma|add)_f32"


def device_disassembly(lib: str):
    """[(kernel symbols, disassembly text)] per translation unit of a BUILT library: every gfx950 code object of its .hip_fatbin section, taken out with
    llvm-objcopy / clang-offload-bundler and disassembled with llvm-objdump -- the linked product, not a recompilation of the sources."""
    import re
    import shutil
    import tempfile

    def tool(name):
        t = os.path.join(LLVM_BIN, name)
        t = t if os.path.exists(t) else shutil.which(name)
        if not t:
            raise RuntimeError("the packed-f32 guard needs " + name + " (ROCm's llvm/bin)")
        return t

    units = []
    with tempfile.TemporaryDirectory(prefix="airband_fatbin_") as d:
        fat = os.path.join(d, "fat.bin")
        _run([tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.devnull])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        if not starts:
            raise RuntimeError("no offload bundle in .hip_fatbin of " + lib)
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            piece, co = os.path.join(d, "bundle%d.bin" % n), os.path.join(d, "unit%d.co" % n)
            open(piece, "wb").write(blob[a:b])
            _run([tool("clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH, "--input=" + piece, "--output=" + co])
            text = _run([tool("llvm-objdump"), "-d", co])
            units.append((re.findall(r"^[0-9a-f]+ <(\w+)>:", text, re.M), text))
    return units


def check_no_packed_f32(lib: str) -> int:
    """THE BUILD'S OWN GATE (round 6; profiles/r05_event_hunt.md): a library whose device code holds a packed-f32 vector instruction is not handed out.  DEVICE_FLAGS
    asks the compiler not to emit them; this looks at what it did -- a different compiler, a flag that stops meaning what it meant, hand-written asm: the build fails
    instead of shipping code whose lanes 48 - 63 can go wrong beside another process.  Returns the number of instructions looked at."""
    import re

    n = 0
    for kernels, text in device_disassembly(lib):
        for line in text.split("\n"):
            if re.match(r"^\s+[sv]_\w+|^\s+(ds|global|buffer|flat|scratch)_\w+", line):
                n += 1
                if re.match(FORBIDDEN, line):
                    raise RuntimeError("packed-f32 instruction in the device code of %s (kernels of the unit: %s ...): %s" % (lib, ", ".join(kernels[:3]), line.strip()))
    if n < 100000 and not EXTRA:  # the product library is ~4e5 instructions: an empty disassembly must not pass as clean
        raise RuntimeError("packed-f32 guard: only %d device instructions found in %s" % (n, lib))
    return n


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    changed = False
    for name, flags in HIP_SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _newer(src, obj):
            info = ["-DAB_BUILD_DEFINES=\"%s\"" % " ".join(EXTRA)] if name == "airband_hip.cpp" else []
            cmd = [hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + flags + DEVICE_FLAGS + EXTRA + info + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)
            changed = True
    for name, flags in HOST_SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _newer(src, obj):
            cmd = ["g++", "-std=c++17", "-fPIC", "-Wall"] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)
            changed = True
    if changed or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,--no-undefined"]
        if verbose:
            print(" ".join(cmd))
        tmp = LIB + ".unchecked"
        cmd[cmd.index("-o") + 1] = tmp
        _run(cmd)
        # experiment builds that ask for the instructions back (AIRBAND_EXTRA_DEFINES="... +packed-fp32-ops", the A/B partner of profiles/r05_event_hunt.md) are the one exception
        if not any("+packed-fp32-ops" in e for e in EXTRA):
            try:
                check_no_packed_f32(tmp)
            except Exception:
                os.unlink(tmp)
                raise
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
