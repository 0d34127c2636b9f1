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


def test_compact_line_carries_the_quoted_numbers_in_under_6_kb(capsys, tmp_path, monkeypatch):
    """the LAST stdout line is a compact record (< 6 KB) that still carries `roofline`, `cpu_baseline` and - flat - every figure DESIGN.md
    section 3 quotes; the full record goes to bench_full.json + stderr.  Fed with a full record a real run produced (profiles/r05_bench_cfg2.json)."""
    import bench
    full = json.loads((ROOT / "profiles" / "r05_bench_cfg2.json").read_text().strip().splitlines()[-1])
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    bench.emit(full)
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6144, len(lines[0])
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in c, k
    assert c["value"] == full["value"] and c["config"]["workload"].startswith("cfg2")
    assert c["roofline"]["bound"] == "mfma" and c["roofline"]["frac"] == full["roofline"]["frac"] and c["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert c["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == 16
    assert c["cfg3_plan_ms"] == full["guided"]["cfg3"]["plan_ms"] and c["cfg4_plan_ms"] == full["guided"]["cfg4"]["plan_ms"]
    assert c["cfg5_shard_plan_ms"] == full["sharded"]["plan_ms_per_rank_max"]
    assert c["cfg3_equal_to_3sf"] == [True] * 4 and c["cfg4_within_fp32_class"] is True
    assert c["guide_us_per_launch"]["panda"] == full["guided"]["cfg4"]["guide_kernel"]["us_per_launch"]
    assert c["serving_ms_per_context"] == full["serving"]["ms_per_context"]
    assert c["train_ms"] == {"b32_D4": full["training"]["ms_per_train_step"], "b128_D14": full["training"]["batch128_D14"]["ms_per_train_step"],
                             "b512_D14": full["training"]["batch512_D14"]["ms_per_train_step"],
                             "three_level_b128_D14": full["training"].get("three_level_batch128_D14", {}).get("ms_per_train_step")}
    assert len(c["trained"]["free_rate_guided"]) == 3
    assert json.loads((tmp_path / "bench_full.json").read_text()) == full
    err = [ln for ln in cap.err.splitlines() if ln.startswith("# bench_full ")]
    assert len(err) == 1 and json.loads(err[0][len("# bench_full "):]) == full
    # N > 1: the fabric's first-contact fields ride in the compact line
    multi = dict(full, n_gpus=8)
    multi["sharded"] = dict(full["sharded"], process_group_world_size=8, backend="nccl", rccl_version="2.22.3")
    c8 = bench.compact_line(multi)
    assert c8["rccl_world_size"] == 8 and c8["checksums_ok"] is True and "collective" in c8["gather_ms"] and len(json.dumps(c8)) < 6144


def test_profile_tools_share_the_bench_fingerprint_function():
    """The tools that stamp a profile with the kernel-source fingerprint must use bench.csrc_fingerprint itself: a private copy in tools/pmc_traffic.py once
    missed a file family bench.py had started to leave out, and every PMC traffic figure of round 6 was labelled `measured_on_these_sources: False`."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for tool in ("pmc_traffic.py", "kernel_stats_json.py"):
        src = (root / "tools" / tool).read_text()
        assert "bench.csrc_fingerprint()" in src and "hashlib" not in src, tool
