"""CPU-only checks of bench.py's record plumbing (no GPU, no kernels): the committed-rocprofv3-summary lookup behind `roofline.duration_source`."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_kernel_stats_sidecar_and_lookup(tmp_path, monkeypatch):
    import bench
    csv = tmp_path / "stats.csv"
    csv.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                   '"void mpdx::fused_program_kernel<A>(mpdx::FusedArgs)",30,1200000,40000.0,50,39000,42000,1\n'
                   '"void mpdx::fused_program_kernel<B>(mpdx::FusedArgs)",30,900000,30000.0,40,29000,31000,1\n'
                   '"void mpdx::conv_block_kernel<0, 5, 1, 32, 32, 1, 8, mpdx::GeoL8<16>, 2>(mpdx::ConvArgs)",120,1200000,10000.0,9,9000,11000,1\n'
                   '"__amd_rocclr_copyBuffer",5,10000,2000.0,1,1000,3000,1\n')
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_stats_json.py"), str(csv), "100", "python bench.py --steps 3"],
                         capture_output=True, text=True, check=True).stdout
    rec = json.loads(out)
    assert rec["batch"] == 100 and rec["csrc_fingerprint"] == bench.csrc_fingerprint()
    assert len(rec["kernels"]) == 3 and all("mpdx::" in k for k in rec["kernels"])   # foreign kernels are dropped
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r07_cfg2_kernel_stats.json").write_text(out)
    stale = dict(rec, csrc_fingerprint="0" * 16)
    stale["kernels"] = {k: dict(v, avg_ns=1.0) for k, v in rec["kernels"].items()}
    (prof / "r08_cfg2_kernel_stats.json").write_text(json.dumps(stale))   # newer round, OTHER sources: must be skipped
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "csrc_fingerprint", lambda: rec["csrc_fingerprint"])
    us, name = bench._rocprof_class_us(100, "fused_program_kernel", 2)   # a class of two launches per pass, two kernel instances
    assert name == "r07_cfg2_kernel_stats.json" and abs(us - 70.0) < 1e-9   # 2 x the call-weighted average of 40 and 30 us
    us, _ = bench._rocprof_class_us(100, "conv_block_kernel<0, 5, 1, 32, 32, ", 7)
    assert abs(us - 70.0) < 1e-9
    assert bench._rocprof_class_us(6400, "fused_program_kernel", 2) == (None, None)          # other batch
    assert bench._rocprof_class_us(100, "conv_ws_kernel<16, 32, false", 7) == (None, None)    # kernel not in the profile
