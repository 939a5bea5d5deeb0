#!/usr/bin/env python3
"""Static instruction tally per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only): VALU / SALU / LDS / global, registers, occupancy.
usage: isa_count.py file.s [kernel-substring]"""
import re, sys

def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if key not in name or "amdhsa_kernel " + name not in s:
            continue
        cnt = lambda pat: len(re.findall(r"^\s+" + pat, body, re.M))
        meta = s[s.index(".amdhsa_kernel " + name):]
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)
        tail = s[m.end():m.end() + 3000]
        occ = re.search(r"; Occupancy: (\d+)", tail)
        scr = re.search(r"; ScratchSize: (\d+)", tail)
        print("%s\n  VALU %d (pk %d, mfma %d) SALU %d LDS %d (bpermute %d) VMEM %d | vgpr %s occupancy %s scratch %s" % (
            name, cnt(r"v_"), cnt(r"v_pk_"), cnt(r"v_mfma"), cnt(r"s_"), cnt(r"ds_"), cnt(r"ds_bpermute"), cnt(r"(global_|buffer_|flat_|scratch_)"), vg,
            occ.group(1) if occ else "?", scr.group(1) if scr else "?"))

if __name__ == "__main__":
    main()
