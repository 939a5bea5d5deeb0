"""DESIGN.md section 5's table is PRINTED from the tracked bench lines (scripts/design_table.py reads profiles/r06_bench_*.json): a figure quoted there that no longer matches its
file -- a profile run filed without the table being regenerated, a hand-edited number -- fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_section_5_is_the_table_the_tracked_bench_lines_print():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_table.py"), "r06"], check=True, stdout=subprocess.PIPE, text=True, cwd=ROOT).stdout
    rows = [l for l in out.split("\n") if l.startswith("| ")]
    assert len(rows) >= 20
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [r[:120] for r in rows if r not in design]
    assert not missing, missing


def test_at_a_glance_quotes_the_tracked_headline_line():
    import json

    j = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_cfg3.json")))
    head = open(os.path.join(ROOT, "DESIGN.md")).read().split("## 1. The path and its boundary")[0]
    ms, value = j["ms_per_step"], j["value"]
    assert "%.2f ms" % ms in head
    assert "{:,}".format(int(round(value, -2))).replace(",", " ") in head  # Msamples/s, rounded to hundreds, thin-space grouped
    assert "**%.3f**" % j["roofline"]["frac"] in head and "**%.3f**" % j["roofline"]["frac_read_only"] in head
    assert "**%.2f ms**" % j["stage_ms"]["demod"] in head


def test_the_figure_blocks_are_what_the_generator_prints():
    """The at-a-glance table, section 5's table and the clocks paragraph sit between markers and are written by scripts/design_glance.py from the tracked files: hand-edited
    or stale figures fail here."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_glance.py"), "r06", "--check"], stdout=subprocess.PIPE, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout
