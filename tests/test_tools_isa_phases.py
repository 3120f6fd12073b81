"""tools/isa_phases.py on a hand-written listing: phases follow the `; MARK` comments, assembler conditionals are
evaluated (factor_diag's DPP blocks are written under `.if`), comments, labels and directives do not count."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTING = """\t.text
other_kernel:
\tv_mov_b32_e32 v0, v1
\ts_endpgm
my_kernel:                              ; @my_kernel
\ts_load_dword s0, s[4:5], 0x0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\t; MARK gather
\tv_mul_f32_e32 v1, v2, v3
\tv_mul_f32_e64 v4, v2, v3
\t.if 3 == 0
\tv_fmac_f32_dpp v5, v5, v6 row_newbcast:0 row_mask:0xf bank_mask:0xf
\t.endif
\t.if (3 > 1) && (2 != 0)
\tv_fmac_f32_dpp v7, v7, v6 row_newbcast:0 row_mask:0xf bank_mask:0xf
\t.endif
\t; MARK solve
\tds_bpermute_b32 v8, v9, v10
\ts_waitcnt lgkmcnt(0)
\ts_endpgm
"""


def test_phase_histogram_of_a_listing():
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(LISTING)
        path = f.name
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_phases.py"), path, "my_kernel"],
                             check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(path)
    rows = {ln.split()[0]: ln for ln in out.splitlines() if ln.strip()}
    assert rows["start"].split()[1] == "1"                       # the s_load before the first mark
    assert rows["gather"].split()[1] == "3" and "v_mul_f32 2" in rows["gather"] and "v_fmac_f32_dpp 1" in rows["gather"]
    assert rows["solve"].split()[1] == "2" and "ds_bpermute_b32 1" in rows["solve"]
    assert rows["total"].split()[1] == "6"
