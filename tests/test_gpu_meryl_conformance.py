"""The meryl-directory decoder against a REAL database -- the day one is reachable.

No meryl source, binary or database exists in the build image (the reference's src/meryl and src/utility submodules are
empty), so the directory decoder (merfin_amd/csrc/mfx_db.cpp) has only ever read this repo's own writer
(tests/meryl_layout.py) and stays labelled UNVALIDATED.  This test is the one command that validates it:

    MFX_REAL_MERYL_DB=/path/to/reads.meryl MFX_REAL_MERYL_PRINT=/path/to/meryl-print.txt[.gz] \\
        python -m pytest tests/test_gpu_meryl_conformance.py -m gpu

where the second file is `meryl print reads.meryl` made by upstream meryl.  Both are loaded through the library and the
two (k-mer, count) sets must be identical (tools/meryl_conformance.py).  Without the two variables it skips."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_meryl_directory_decoder_against_upstream_print():
    db, txt = os.environ.get("MFX_REAL_MERYL_DB"), os.environ.get("MFX_REAL_MERYL_PRINT")
    if not (db and txt):
        pytest.skip("set MFX_REAL_MERYL_DB (a database written by upstream meryl) and MFX_REAL_MERYL_PRINT (`meryl print` of it) "
                    "to validate the meryl-directory decoder; none is reachable in this image")
    assert os.path.isdir(db) and os.path.exists(txt)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "meryl_conformance.py"), db, txt], capture_output=True, text=True)
    assert r.returncode == 0 and "CONFORMANT" in r.stdout, r.stdout + r.stderr
