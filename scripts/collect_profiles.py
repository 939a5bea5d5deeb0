#!/usr/bin/env python3
"""Copies the summaries of a scripts/profile_round.sh run (gpurun_out/final) into profiles/ under the round's prefix and writes
profiles/<round>_summary.md: the bench lines, per-kernel times (rocprofv3 --kernel-trace --stats), per-kernel HBM traffic
(--pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) and the SQ instruction counters."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r03"


def short(k):
    return k.replace("void ", "").replace("airband::", "").replace("(anonymous namespace)::", "").split("(")[0]


def stats(dirname):
    rows = []
    for f in glob.glob(os.path.join(SRC, dirname, "*", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "airband" in r["Name"] and "siggen" not in r["Name"]:
                rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        shutil.copy(f, os.path.join(DST, "%s_%s_kernel_stats.csv" % (R, dirname.replace("kt_", ""))))
    return rows


def pmc(dirname, mult=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "airband" in r["Kernel_Name"] and "siggen" not in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        keep = os.path.join(DST, "%s_%s.csv" % (R, dirname))
        with open(f) as fin, open(keep, "w") as fout:  # keep only our kernels' rows: the raw file also lists every memcpy
            for i, line in enumerate(fin):
                if i == 0 or ("airband" in line and "siggen" not in line):
                    fout.write(line)
    out = {}
    for k, v in agg.items():
        out[k] = {c: (sum(x[1:]) / (len(x) - 1) if len(x) > 1 else x[0]) for c, x in v.items()}
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    md = ["# Round %s measurements (1x MI355X; `scripts/profile_round.sh` on one gpurun box, committed build)\n" % R[1:].lstrip("0")]
    host = open(os.path.join(SRC, "host.txt")).read().split("\n") if os.path.exists(os.path.join(SRC, "host.txt")) else ["?", "?"]
    md.append("Host of the GPU box: %s hardware threads, %s.\n" % (host[0], host[1].split(":")[-1].strip() if len(host) > 1 else "?"))
    preface = os.path.join(DST, R + "_summary_preface.md")  # hand-written: which commit / build the run is of
    if os.path.exists(preface):
        md.append(open(preface).read().rstrip("\n") + "\n")
    md.append("## bench.py lines\n")
    md.append("| file | workload | Msamples/s | ms/step | channelizer ms | stage 2 ms | bound | frac | read-only frac (8 TB/s) | end-to-end frac | verified | extra |")
    md.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(os.path.join(SRC, R + "_bench_*.json"))):
        try:
            j = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        shutil.copy(f, DST)
        extra = []
        if j.get("host_path"):
            extra.append("host path %s Msamples/s = %s GB/s (%d feeder threads)" % (j["host_path"]["value"], j["host_path"].get("gbytes_per_s"), j["host_path"].get("feeder_threads", 0)))
        if j.get("cpu_baseline"):
            c = j["cpu_baseline"]
            extra.append("CPU reference %s Msamples/s on %s threads (%s; 1 thread %s; f64-FFT %s)" % (c["value"], c["cores"], c.get("fft"), c.get("value_1_thread"), c.get("value_f64_fft")))
        if j["roofline"].get("traffic"):
            extra.append("PMC traffic %.2f GB / launch" % (j["roofline"]["traffic"] / 1e9))
        if j["roofline"].get("rocprof", {}).get("avg_launch_ms"):
            rp = j["roofline"]["rocprof"]
            extra.append("rocprofv3 clock on the channelizer: %.3f ms avg over %d launches = frac %s (read-only %s)" % (rp["avg_launch_ms"], rp["launches"], rp.get("frac"), rp.get("frac_read_only")))
        cfg = j["config"]
        if j.get("verify_all"):
            v = j["verify_all"]
            extra.append("whole handle: %s" % (("%d dongles, differing %s" % (v["dongles"], v["differing"])) if "dongles" in v else v))
        md.append("| `%s` | %s, %s, fft %s, %s%s%s | %s | %s | %.3f | %.3f | %s | %s | %s | %s | %s | %s |" % (
            os.path.basename(f), cfg["workload"].split(",")[0][:40], cfg.get("sample_format"), cfg.get("fft_size"), cfg["channelizer"], ", afc" if cfg.get("afc") else "",
            ", " + cfg["schedule"][:9] if "pipelined" in cfg["schedule"] else "",
            j["value"], j["ms_per_step"], j["stage_ms"]["channelizer"], j["stage_ms"]["demod"], j["roofline"]["bound"], j["roofline"]["frac"], j["roofline"].get("frac_read_only"),
            j["roofline"].get("end_to_end_frac"), j.get("verified_dongles"), "; ".join(extra)))
    for name, title in (("kt_cfg3", "configs[2], default run (kinds side by side)"), ("kt_cfg3_serial", "configs[2], every stage-2 kernel alone (AIRBAND_BENCH_FLAGS=8)"),
                        ("kt_cfg3_afc", "configs[2] with AFC on one channel of every dongle, stage-2 kernels alone (bench.py --afc 2)"),
                        ("kt_cfg3_force_fft", "configs[2] on the wavefront-FFT channelizer (AIRBAND_BENCH_FLAGS=4)"), ("kt_cfg2", "configs[1] (1 024 AM dongles)"),
                        ("kt_cfg4", "configs[3]'s per-GPU shard (32 768 dongles), kinds side by side"),
                        ("kt_cfg4_serial", "configs[3]'s per-GPU shard (32 768 dongles), every stage-2 kernel alone"),
                        ("kt_am65536", "65 536 AM dongles"), ("kt_cs16", "configs[2] with CS16 dongles"), ("kt_f32", "32 768 CF32 dongles")):
        rows = stats(name)
        if not rows:
            continue
        md.append("\n## Kernel trace: %s\n" % title)
        md.append("| kernel | calls | avg ms | min | max |")
        md.append("|---|---|---|---|---|")
        for k, c, a, lo, hi in sorted(rows, key=lambda r: -r[2]):
            md.append("| `%s` | %d | %.3f | %.3f | %.3f |" % (k, c, a, lo, hi))
    fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
    if fetch or write:
        md.append("\n## HBM-side traffic per launch (PMC, separate passes; FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024; first launch dropped)\n")
        md.append("| kernel | fetched GB | written GB |")
        md.append("|---|---|---|")
        for k in sorted(set(fetch) | set(write)):
            md.append("| `%s` | %.2f | %.2f |" % (k, fetch.get(k, {}).get("FETCH_SIZE", 0) * 2048 / 1e9, write.get(k, {}).get("WRITE_SIZE", 0) * 1024 / 1e9))
    sq = pmc("pmc_sq_serial")
    if sq:
        md.append("\n## SQ counters, every kernel alone (instructions per launch; fractions of wave-cycles)\n")
        md.append("| kernel | VALU | SALU | branch | parked at s_waitcnt | issue stall | VALU active |")
        md.append("|---|---|---|---|---|---|---|")
        for k, d in sq.items():
            wc = d.get("SQ_WAVE_CYCLES", 1)
            md.append("| `%s` | %.3g | %.3g | %.3g | %.2f | %.2f | %.2f |" % (k, d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_BRANCH", 0),
                                                                       d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_VALU", 0) / wc))
    open(os.path.join(DST, R + "_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
