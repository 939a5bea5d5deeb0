#!/usr/bin/env python3
"""Prints DESIGN.md section 5's table from profiles/<round>_bench_*.json (so that the document quotes the tracked files and nothing else).  usage: design_table.py r06"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"


def L(n):
    return json.load(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (R, n))))


def num(v):
    return "{:,}".format(int(round(v))).replace(",", " ")


def row(label, n, roof=None, extra_rt=""):
    j = L(n)
    r, st = j["roofline"], j["stage_ms"]
    fmt = "%.2f (%.2f / %.2f)" if j["ms_per_step"] >= 5 else "%.3f (%.3f / %.3f)"
    ms = fmt % (j["ms_per_step"], st["channelizer"], st["demod"])
    rf = roof if roof else "%.3f (%.3f)" % (r["frac"], r.get("frac_read_only", 0))
    rt = int(round(j["value"] / 2.56 / 1000.0)) * 1000
    return "| %s | %s | %s | %s | %s%s | `%s_bench_%s.json` |" % (label, num(j["value"]), ms, rf, num(rt), extra_rt, R, n)


def main():
    c3 = L("cfg3")
    rp = c3["roofline"].get("rocprof", {})
    of = c3.get("open_fraction", {})
    rows = ["| workload | Msamples/s | ms/step (channelizer / stage 2) | channelizer roofline (HIP events, of 8 TB/s; read-only in brackets) | real-time dongles per GPU | `profiles/` |",
            "|---|---|---|---|---|---|",
            row("**configs[2]** 65 536 × 8 mixed AM/NFM+CTCSS, %s open" % of.get("mean", "0.44"), "cfg3",
                "**%.3f** (%.3f); rocprofv3's clock, %s-launch child run (`roofline.rocprof`): %s ms = %s (%s)" % (
                    c3["roofline"]["frac"], c3["roofline"]["frac_read_only"], rp.get("launches"), rp.get("avg_launch_ms"), rp.get("frac"), rp.get("frac_read_only"))),
            row("configs[1] 1 024 × 8 AM", "cfg2", extra_rt=" (one wavefront's dependent chain, §4.2)"),
            row("configs[1], `--pipelined`", "cfg2_pipelined"),
            row("65 536 × 8 AM (hop 640)", "am65536"),
            row("configs[2], `--pipelined` (the channelizer held to five wavefronts per CU, §4.3)", "cfg3_pipelined", "(kernels share the chip)"),
            row("configs[2], 65 536 DISTINCT channel plans (one coefficient table per dongle, §4.1)", "cfg3_plans65536"),
            row("configs[2] + AFC on one channel per dongle (`--afc 2`)", "cfg3_afc"),
            row("configs[2] with CS16 dongles", "cfg3_cs16"),
            row("configs[2] at 2.4 MS/s (hops of 300 bytes)", "cfg3_2400k"),
            row("configs[2] at 2.0 MS/s (hops of 250 bytes: `AL = 2`)", "cfg3_2000k"),
            row("configs[2] at fft_size 256", "cfg3_fft256")]
    for n, lab in (("cfg3_fft1024", "configs[2] at fft_size 1024 (2 window pieces)"), ("cfg3_fft2048", "fft_size 2048 (4 pieces)"), ("cfg3_fft4096", "fft_size 4096 (8 pieces)"),
                   ("cfg3_fft8192", "fft_size 8192 (2 passes of 8)")):
        r = L(n)["roofline"]
        rows.append(row(lab, n, "nothing full (§4.1): HBM %.2f (read-only), int8 matrix pipe %.2f of peak" % (r.get("frac_read_only", 0), r.get("mfma_frac", 0))))
    rows.append(row("configs[2] forced onto the wavefront FFT (exchange kernel, §4.4; it waits at every exchange since round 5)", "cfg3_force_fft"))
    r = L("f32_32768")["roofline"]
    rows.append(row("32 768 CF32 dongles × 8 mixed, float32 matrix pipe (§4.5)", "f32_32768", "float32 matrix pipe %.2f; HBM %.2f read-only" % (r["frac"], r.get("frac_read_only", 0))))
    rows.append(row("the same forced onto the wavefront FFT", "f32_32768_force_fft"))
    for n, lab in (("f32_32768_fft4096", "the same at fft_size 4096 (two window segments, §4.5)"), ("f32_32768_afc", "the same with AFC on one channel per dongle (float tables re-tuned on the device)"),
                   ("f32_32768_2000k", "the same at 2.0 MS/s (hops of 125 samples: `LAY = 1`)")):
        r = L(n)["roofline"]
        rows.append(row(lab, n, "float32 matrix pipe %.2f; HBM %.2f read-only" % (r["frac"], r.get("frac_read_only", 0))))
    r = L("f32_am16384")["roofline"]
    rows.append(row("16 384 CF32 dongles × 8 AM (hop 320 samples)", "f32_am16384", "float32 matrix pipe %.2f; HBM %.2f read-only" % (r["frac"], r.get("frac_read_only", 0))))
    rows.append(row("configs[3] shard: 32 768 dongles per GPU (stage 2 regrouped by residency, §4.2)", "cfg4_shard"))
    rows.append(row("the same in slot order (`AIRBAND_HIP_FLAG_NO_REGROUP`)", "cfg4_shard_slot_order"))
    rows.append(row("configs[2] + 64 mixers (configs[4] exchange through `airband_hip_allreduce_mixers`, RCCL at world size 1)", "cfg3_mixers64"))
    j = L("cfg3_hostpath")
    rows.append(row("configs[2], the run that also times the host path (%.0f Msamples/s = %s GB/s over PCIe, never `value`)" % (j["host_path"]["value"], j["host_path"].get("gbytes_per_s")), "cfg3_hostpath"))
    cb = c3["cpu_baseline"]
    rows.append("| reference CPU path, %d host threads = the container's CPU quota (%s), float FFT | %s (%s on one thread: %.2f of linear; %s behind the float64 FFT; %s on %d threads, throttled) | — | — | %d | `%s_bench_cfg3.json`, `cpu_baseline` |" % (
        cb["cores"], cb["cpu_model"], num(cb["value"]), num(cb["value_1_thread"]), cb["scaling_vs_linear"], num(cb["value_f64_fft"]), num(cb["value_at_2x_cores_threads"]), 2 * cb["cores"],
        int(cb["value"] / 2.56), R))
    print("\n".join(rows))
    td = c3["roofline"]["traffic_detail"]
    f = sum(v["fetch_size_bytes"] for v in td["other_kernels"].values())
    w = sum(v["write_size_bytes"] for v in td["other_kernels"].values())
    print("\nTRAFFIC channelizer %.2f + %.2f GB, stage 2 %.2f + %.2f GB, total %.1f GB" % (td["fetch_size_bytes"] / 1e9, td["write_size_bytes"] / 1e9, f / 1e9, w / 1e9, (td["fetch_size_bytes"] + td["write_size_bytes"] + f + w) / 1e9))
    print("VERIFY", c3.get("verify"), c3.get("verify_all"), "OPEN", of)


if __name__ == "__main__":
    main()
