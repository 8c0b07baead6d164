"""Short runs of the randomised campaigns under tools/ (their full-size runs are recorded in the
design notes): each is a script that exits non-zero on the first disagreement."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seeds, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(seeds)], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout


def test_matchers_agree_on_random_batches(device):
    out = _run("fuzz_knn.py", 25)
    assert "0 differences" in out


def test_filtered_scoring_kernels_agree_near_the_threshold(device):
    out = _run("fuzz_score.py", 6)
    assert "0 disagreements" in out


@pytest.mark.parametrize("l2", ["0", "1"])
def test_pipeline_equals_oracle_on_random_sequences(device, l2):
    out = _run("fuzz_pipeline.py", 8, {"FUZZ_L2": l2})
    assert "0 mismatches" in out
