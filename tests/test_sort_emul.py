"""The product's std::sort emulation (rs_sort_emul.h, host+device code) against libstdc++ on the CPU."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_introsort_emulation_matches_libstdcxx(tmp_path):
    exe = tmp_path / "sort_emul_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "csrc" / "sort_emul_check.cpp")],
                   check=True)
    out = subprocess.run([str(exe), "6000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_unordered_map_order_matches_libstdcxx(tmp_path):
    """SubOpt iterates an unordered_map<int,int>; the product computes that order itself (rs_umap_order)."""
    exe = tmp_path / "umap_emul_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "csrc" / "umap_emul_check.cpp")],
                   check=True)
    out = subprocess.run([str(exe), "60000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
