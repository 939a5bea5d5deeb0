#!/usr/bin/env python3
"""Re-embeds integration/demod_hip.cpp into INTEGRATION.md (between the markers), so that the document shows the very file
that oracle/Makefile compiles against the patched reference."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md_path = os.path.join(ROOT, "INTEGRATION.md")
md = open(md_path).read()
shim = open(os.path.join(ROOT, "integration", "demod_hip.cpp")).read()
new = re.sub(r"(<!-- demod_hip.cpp:begin -->\n```cpp\n).*?(```\n<!-- demod_hip.cpp:end -->)", lambda m: m.group(1) + shim + m.group(2), md, flags=re.S)
open(md_path, "w").write(new)
print("INTEGRATION.md synced" if new != md else "INTEGRATION.md already in sync")
