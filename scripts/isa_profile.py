#!/usr/bin/env python3
"""Static instruction tally of a demod kernel per source line (CPU-only tuning aid).

usage: isa_profile.py <kernel-substring> [extra hipcc flags...]
Compiles csrc/demod.hip to gfx950 assembly with line tables and prints, for the first kernel whose mangled name
contains the substring, VALU / SALU / memory instruction counts per source line (top 40) and in total.
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", "demod.hip")

def main():
    key = sys.argv[1]
    extra = sys.argv[2:]
    out = os.path.join(tempfile.gettempdir(), "demod_prof.s")
    import importlib.util  # the product's own flags (rtlsdr-airband_amd/_build.py: optimisation level, contraction, no packed-f32 instructions)
    spec = importlib.util.spec_from_file_location("airband_build_flags", os.path.join(ROOT, "rtlsdr-airband_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17"] + list(b.HIP_SOURCES["demod.hip"]) + list(b.DEVICE_FLAGS) + [
           "-gline-tables-only", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, SRC] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    per = collections.defaultdict(lambda: [0, 0, 0])
    cur = (0, 0)
    tot = [0, 0, 0]
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
    srcs = {}
    def text(key):
        f, ln = key
        name = files.get(f, "")
        if name not in srcs:
            try:
                srcs[name] = open(name).read().split("\n")
            except OSError:
                srcs[name] = []
        t = srcs[name]
        return "%s:%d %s" % (os.path.basename(name), ln, t[ln - 1].strip()[:100] if 0 < ln <= len(t) else "")
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (int(m.group(1)), int(m.group(2))); continue
        t = l.strip()
        k = 0 if t.startswith("v_") else 1 if t.startswith("s_") else 2 if re.match(r"(ds_|global_|buffer_|flat_)", t) else -1
        if k >= 0:
            per[cur][k] += 1; tot[k] += 1
    print("total VALU %d SALU %d MEM %d" % tuple(tot))
    for key, c in sorted(per.items(), key=lambda kv: -kv[1][0])[:60]:
        print("v=%4d s=%4d m=%3d | %s" % (c[0], c[1], c[2], text(key)))

if __name__ == "__main__":
    main()
