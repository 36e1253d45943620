"""The generated device tables are what their generators produce now (tools/gen_*.py --check), and the erfc table reaches the
accuracy its header states."""
import os
import subprocess
import sys

import helpers


def _run(script):
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", script), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_solid_harmonic_tables_are_current():
    _run("gen_solid_harmonics.py")


def test_erfc_table_is_current_and_accurate():
    out = _run("gen_erfc_table.py")
    err = float(out.strip().splitlines()[-1].split()[-1])  # "erfc via table: max relative error 2.55e-15"
    assert err < 5e-15
