"""The engine's host-side structural analysis (known nodes, aliases, components, classes, gather lists) under
AddressSanitizer / UBSan on random device tables: every accepted circuit must produce consistent tables, every rejected one
an error code — never a crash or an out-of-bounds access."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_structural_analysis_is_sanitizer_clean_on_random_circuits(tmp_path):
    exe = str(tmp_path / "an_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cedarsim.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "host_analysis_fuzz.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "analysed" in r.stdout


def test_sparse_analysis_is_sanitizer_clean_on_random_matrices(tmp_path):
    """KLU-style host analysis of the sparse path (transversal, ordering, symbolic fill, levels, operation lists)."""
    exe = str(tmp_path / "sp_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cedarsim.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "host_sparse_fuzz.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "fail 0" in r.stdout


def test_subtree_form_of_the_sparse_analysis_replays_exactly(tmp_path):
    """SubtreePlan (ch_sparse_host.hpp): arrow matrices — 64 to 100 independent blocks under a border of one to three rows — are split
    into groups + a top block, and a host replay of what sp3_group_kernel / sp3_top_kernel / sp3_back_kernel do with the blobs equals a
    dense solve with partial pivoting; under ASan / UBSan / _GLIBCXX_ASSERTIONS."""
    exe = str(tmp_path / "st_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cedarsim.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "host_subtree_fuzz.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "60 plans replayed" in r.stdout and " 0 failures" in r.stdout


def test_static_pivot_sequence_survives_cancellation_in_mna_matrices(tmp_path):
    """The GPU refactorisation never searches for a pivot, so the sequence the host analysis hands it has to be sound for the values
    it was made from.  Two MNA Jacobians of a random RLC / controlled-source network (the oracle's, written by
    tests/golden/make_mna_jacobian.py: seed 20095 of scripts/extended_fuzz.py at alpha0 = 0 and 1e12): a matching on entries that are
    large in their rows — rounds 1-3 — divides by an exact zero after a few eliminations on both; the sequence taken from an actual
    elimination (ch_sparse_host.hpp numeric_pivot_rows, KLU's rule) solves them to 1e-8 of the right-hand side."""
    import subprocess
    exe = str(tmp_path / "replay")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cedarsim.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "host_matrix_replay.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    g = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([exe, os.path.join(g, "mna_jacobian_seed20095_dc.txt"), os.path.join(g, "mna_jacobian_seed20095_tran.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 bad" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_return_address_scanner_flags_the_pattern(tmp_path):
    """scripts/check_return_address.py on two hand-written device functions: one whose long-branch expansion writes s[30:31]
    without a saved copy (the code-generator defect worked around in csrc/va_rt.hpp), one that saved the pair first."""
    import subprocess
    import sys
    bad = """
_Z3badv: ; @_Z3badv
\ts_waitcnt vmcnt(0)
\ts_getpc_b64 s[30:31]
.Lpost_getpc1:
\ts_add_u32 s30, s30, (.LBB0_2-.Lpost_getpc1)&4294967295
\ts_setpc_b64 s[30:31]
.LBB0_2:
\ts_setpc_b64 s[30:31]
\t.size\t_Z3badv, .Lfunc_end0-_Z3badv
_Z4goodv: ; @_Z4goodv
\tv_writelane_b32 v255, s30, 0
\tv_writelane_b32 v255, s31, 1
\ts_getpc_b64 s[30:31]
\ts_setpc_b64 s[30:31]
\tv_readlane_b32 s30, v255, 0
\ts_setpc_b64 s[30:31]
\t.size\t_Z4goodv, .Lfunc_end1-_Z4goodv
"""
    f = tmp_path / "dev.s"
    f.write_text(bad)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_return_address.py"), str(f)], capture_output=True, text=True)
    assert r.returncode == 1 and "_Z3badv" in r.stdout and "_Z4goodv" not in r.stdout.split("checked")[0].replace("_Z3badv", ""), r.stdout
    f.write_text(bad.split("_Z4goodv: ; @_Z4goodv")[0].replace("s_getpc_b64 s[30:31]", "s_getpc_b64 s[98:99]"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_return_address.py"), str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
