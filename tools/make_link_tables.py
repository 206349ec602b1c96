#!/usr/bin/env python3
"""Write radiosaber_amd/data/link_tables_glibc_2_35.json: the EESM constants E[1..15] / X[1..13] as THIS host's libm evaluates them
(rs_link_tables: exactly the reference's expressions, src/utility/eesm-effective-sinr.h:33-46, AMCModule.cpp:253-261), as hex floats.

Run it only in the build container (g++ 11.4, glibc 2.35, x86-64) -- the libm behind SURVEY.md Appendix A and every fixture under
tests/golden/.  It refuses to write values that differ from tests/golden/appendix_a.json (the values the unmodified reference printed):
the data file is a second copy of those numbers inside the product package, not a new measurement.  radiosaber_amd/build.py turns it
into csrc/rs_link_pinned.inc; rs_config.link_tables = RS_LINK_PINNED_GLIBC_2_35 serves it to the kernels."""
import json
import platform
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import radiosaber_amd as rs  # noqa: E402

t = rs.link_tables()
E = [float.hex(float(x)) for x in t["eesm_e"][1:16]]
X = [float.hex(float(x)) for x in t["eesm_x"][1:14]]
ka = json.loads((ROOT / "tests" / "golden" / "appendix_a.json").read_text())
assert [float.fromhex(h) for h in ka["eesm_E_hex"]] == [float.fromhex(h) for h in E], "this host's libm is not the fixtures' libm"
assert [float.fromhex(h) for h in ka["eesm_X_hex"]] == [float.fromhex(h) for h in X], "this host's libm is not the fixtures' libm"
out = {"_source": f"tools/make_link_tables.py on {platform.libc_ver()[0]} {platform.libc_ver()[1]}, {platform.machine()}; equal to "
                  "tests/golden/appendix_a.json (SURVEY.md Appendix A: printed by the unmodified reference, g++ 11.4 -O0, glibc 2.35)",
       "eesm_E_hex": E, "eesm_X_hex": X}
(ROOT / "radiosaber_amd" / "data" / "link_tables_glibc_2_35.json").write_text(json.dumps(out, indent=1) + "\n")
print("wrote", len(E), "+", len(X), "constants")
