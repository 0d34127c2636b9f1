"""The randomised configuration sweeps of tools/fuzz_*.py, a few seeded cases each (the tools print one line per case and a summary): no mismatch - and no
refusal, the generators only draw configurations the library documents as accepted."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_plan.py", 12, 11), ("fuzz_contexts.py", 6, 12), ("fuzz_guide.py", 12, 13), ("fuzz_train.py", 8, 14), ("fuzz_planner.py", 8, 15)])
def test_fuzz_tool_reports_no_mismatch(tool, cases, seed):
    out = subprocess.run([sys.executable, str(ROOT / "tools" / tool), str(cases), str(seed)], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert out.returncode == 0 and lines, out.stderr[-2000:]
    assert lines[-1].startswith(f"{cases} cases, 0 mismatches"), "\n".join(lines[-cases - 1:])
    assert not [ln for ln in lines if ln.startswith(("MISMATCH", "refused"))], "\n".join(ln for ln in lines if not ln.startswith("ok"))
