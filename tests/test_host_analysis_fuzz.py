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
