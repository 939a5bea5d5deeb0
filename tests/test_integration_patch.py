"""integration/airband_hip.patch is a real patch against the reference tree, and INTEGRATION.md shows the very file it adds."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present on this box")
def test_patch_applies_cleanly_to_the_reference(tmp_path):
    shutil.copytree(os.path.join(REF, "src"), tmp_path / "src")
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(ROOT, "integration", "airband_hip.patch")], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
    patched = (tmp_path / "src" / "rtl_airband.cpp").read_text()
    assert "&demodulate_hip" in patched and "WITH_AIRBAND_HIP" in patched
    # small, and confined to the files INTEGRATION.md lists
    body = open(os.path.join(ROOT, "integration", "airband_hip.patch")).read()
    added = [ln for ln in body.splitlines() if ln.startswith("+") and not ln.startswith("+++")]
    touched = sorted(set(re.findall(r"^\+\+\+ b/src/(\S+)", body, re.M)))
    assert touched == ["CMakeLists.txt", "config.cpp", "config.h.in", "mixer.cpp", "rtl_airband.cpp", "rtl_airband.h", "squelch.cpp", "squelch.h"]
    assert len(added) <= 100, len(added)


def test_integration_md_shows_the_compiled_shim_verbatim():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(ROOT, "integration", "demod_hip.cpp")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", md, re.S)
    assert any(b == shim for b in blocks), "INTEGRATION.md must embed integration/demod_hip.cpp byte for byte (run scripts/sync_integration_md.py)"
