#!/usr/bin/env python3
"""Static instruction tally per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only): VALU / SALU / LDS / global, registers, occupancy.
usage: isa_count.py file.s [kernel-substring] [--blocks [MIN]]
--blocks: also the basic blocks of each matching kernel that hold at least MIN (default 16) instructions, with their no-ops, waits and branch targets --
the hot loop of a kernel is a handful of them, and on this part every instruction of every kind is an issue slot."""
import re
import sys


def tally(b):
    cnt = lambda pat: len(re.findall(r"^\s+" + pat, b, re.M))
    return dict(valu=cnt(r"v_"), pk=cnt(r"v_pk_"), mfma=cnt(r"v_mfma"), salu=cnt(r"s_"), lds=cnt(r"ds_"), bperm=cnt(r"ds_bpermute"),
                vmem=cnt(r"(global_|buffer_|flat_|scratch_)"), nop=cnt(r"s_nop"), wait=cnt(r"s_waitcnt"))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    blocks = "--blocks" in sys.argv
    min_block = 16
    if blocks and sys.argv.index("--blocks") + 1 < len(sys.argv) and sys.argv[sys.argv.index("--blocks") + 1].isdigit():
        min_block = int(sys.argv[sys.argv.index("--blocks") + 1])
        args = [a for a in args if a != str(min_block)]
    s = open(args[0]).read()
    key = args[1] if len(args) > 1 else ""
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if key not in name or "amdhsa_kernel " + name not in s:
            continue
        t = tally(body)
        meta = s[s.index(".amdhsa_kernel " + name):]
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)
        tail = s[m.end():m.end() + 3000]
        occ = re.search(r"; Occupancy: (\d+)", tail)
        scr = re.search(r"; ScratchSize: (\d+)", tail)
        print("%s\n  VALU %d (pk %d, mfma %d) SALU %d LDS %d (bpermute %d) VMEM %d | vgpr %s occupancy %s scratch %s" % (
            name, t["valu"], t["pk"], t["mfma"], t["salu"], t["lds"], t["bperm"], t["vmem"], vg, occ.group(1) if occ else "?", scr.group(1) if scr else "?"))
        if blocks:
            parts = re.split(r"^(\.LBB\d+_\d+):(.*)$", body, flags=re.M)
            rows = [("entry", "", parts[0])] + [(parts[i], parts[i + 1], parts[i + 2]) for i in range(1, len(parts) - 2, 3)]
            for label, hdr, b in rows:
                bt = tally(b)
                n = bt["valu"] + bt["salu"] + bt["lds"] + bt["vmem"]
                if n < min_block:
                    continue
                to = re.findall(r"^\s+s_c?branch\S*\s+(\S+)", b, re.M)
                loop = " loop" if "Loop Header" in hdr else (" in-loop" if "in Loop" in hdr or "Parent Loop" in hdr else "")
                print("    %-12s%-8s %4d = V %3d S %3d (nop %2d, wait %2d) LDS %3d VMEM %2d  -> %s" % (label, loop, n, bt["valu"], bt["salu"], bt["nop"], bt["wait"], bt["lds"], bt["vmem"], " ".join(to)))


if __name__ == "__main__":
    main()
