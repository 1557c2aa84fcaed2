"""Build-time safety net for the hand-scheduled GEMM K loop: replays the in-order LDS-return rule over the gfx950
assembly and fails if any MFMA could read a fragment register before its ds_read_b128 has landed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counted_lgkmcnt_waits_cover_every_mfma_operand():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 violations" in r.stdout and int(r.stdout.split()[1]) > 500, r.stdout
