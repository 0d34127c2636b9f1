"""N>1 path on CPU: context sharding + the final gather on the gloo backend, world_size 2 (and 3, uneven blocks)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from mpd_public_amd.parallel import shard_range, expand_contexts, plan_contexts, gather_trajectories, verify_gather, shard_checksum


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 128, 1024, 1025):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_expand_contexts_layout():
    s = torch.arange(6.0).reshape(3, 2)
    hs, hg = expand_contexts(s, s + 10, 4)
    assert hs.shape == (12, 2)
    assert torch.equal(hs[4:8], s[1].expand(4, 2)) and torch.equal(hg[8:], (s[2] + 10).expand(4, 2))


def _stub_planner(hc, B, npc, **kw):
    """Deterministic stand-in for model.plan: trajectory b = start*(1-s) + goal*s, tagged with its in-context index."""
    hs, hg = hc[0], hc[63]
    s = torch.linspace(0, 1, 64).reshape(1, 64, 1)
    x = hs[:, None, :] * (1 - s) + hg[:, None, :] * s
    return x + (torch.arange(B) % npc).reshape(B, 1, 1) * 1e-3


def _reference(start, goal, n):
    x, _ = plan_contexts(None, start, goal, n, planner=_stub_planner, horizon=64)
    return x


def _worker(rank, world, port, C, n, max_batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    start, goal = torch.rand((C, 4), generator=g), torch.rand((C, 4), generator=g)
    local, (lo, hi) = plan_contexts(None, start, goal, n, rank=rank, world_size=world, max_batch=max_batch, planner=_stub_planner, horizon=64)
    assert local.shape[0] == (hi - lo) * n
    full = gather_trajectories(local, C, n)
    ok = torch.equal(full, _reference(start, goal, n))
    # the one-hop variant (world-1 direct sends + receives per rank, one batch) must deliver the same bits, and the checksum
    # protocol must accept the gathered tensor and reject a corrupted one
    hop = gather_trajectories(local, C, n, mode="one_hop")
    ok &= torch.equal(hop, full)
    ok &= verify_gather(full, local, C, n) and verify_gather(hop, local, C, n)
    bad = full.clone()
    bad[-1, 3, 1] += 1e-6
    ok &= not verify_gather(bad, local, C, n)
    q.put((rank, bool(ok), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,C,n,max_batch", [(2, 8, 5, 8192), (2, 7, 3, 6), (3, 10, 4, 8)])
def test_sharded_plan_and_gather_gloo(world, C, n, max_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, n, max_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] == (C * n, 64, 4) for r in res)


def test_checksum_is_position_weighted_over_bit_patterns():
    from mpd_public_amd.parallel import block_checksums, _CK_MULT
    x = torch.tensor([1.0, -2.5, 0.0, 3.25])
    bits = x.view(torch.int32).to(torch.int64)
    want = sum(int(b) * (i * _CK_MULT + 1) for i, b in enumerate(bits)) + 4
    want = (want + 2 ** 63) % 2 ** 64 - 2 ** 63   # wrapping int64 arithmetic
    assert int(shard_checksum(x)) == want
    assert int(shard_checksum(x[:0])) == 0
    assert verify_gather(x, x, 1, 4)   # no process group: the tensor checks itself
    # what a plain sum of bit patterns cannot see: a permutation inside the block, two compensating bit errors
    y = torch.randn(6, 8, 4)
    perm = y[[1, 0, 2, 3, 4, 5]]
    assert int(shard_checksum(perm)) != int(shard_checksum(y))
    z = y.clone().view(torch.int32)
    z.view(-1)[3] += 1
    z.view(-1)[77] -= 1
    assert int(z.to(torch.int64).sum()) == int(y.view(torch.int32).to(torch.int64).sum())   # the plain sum is blind to it
    assert int(shard_checksum(z.view(torch.float32))) != int(shard_checksum(y))
    # block-local positions: a block's checksum does not depend on where it sits in the gathered tensor
    cs = block_checksums(y, [2, 1, 3])
    assert [int(v) for v in cs] == [int(shard_checksum(y[0:2])), int(shard_checksum(y[2:3])), int(shard_checksum(y[3:6]))]


# ---------------------------------------------------------------------------------------------- bench.py's sharded leg on the CPU rig
def _rig_worker(rank, world, port, fail_rank, q):
    """bench.sharded_leg on gloo with a stand-in planner: ONE context per rank (the 8-GPU run's shard arithmetic at its smallest),
    both gathers, the checksum protocol, the pre-attempt agreement and the record's shape - no GPU, no kernels."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def fake_plan(hs, hg):
        s = torch.linspace(0, 1, 64).reshape(1, 64, 1)
        return (hs[:, None, :] * (1 - s) + hg[:, None, :] * s + rank * 1e-3).contiguous()
    # an asymmetric failure BEFORE the one-hop attempt (one rank's pre-flight check fails): the ranks must skip the attempt TOGETHER
    pre = (lambda r: "injected: this rank cannot post its sends" if r == fail_rank else None) if fail_rank >= 0 else None
    rec = bench.sharded_leg(rank, world, dist, "cpu", plans=1, n_ctx=1, n_samples=5, plan_fn=fake_plan, hop_timeout_s=20.0, hop_precheck=pre)
    q.put((rank, rec))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fail_rank", [(8, -1), (3, 1)])
def test_bench_sharded_leg_on_the_gloo_rig(world, fail_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rig_worker, args=(r, world, port, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rec = res[0]
    assert rec["ranks"] == world == rec["process_group_world_size"] and rec["backend"] == "gloo"
    assert rec["workload"].startswith("cfg5 shard per rank: 1 contexts x 5 = 5") and f"{world} contexts in total" in rec["workload"]
    assert len(rec["plan_ms_per_rank"]["all"]) == world and rec["all_gather_bytes_per_rank"] == 5 * 64 * 14 * 4
    assert rec["scaling"] == "weak" and rec["denoising_steps_per_s"] > 0
    hop = rec["gather_ms"]["one_hop_send_recv"]
    if fail_rank < 0:
        assert "error" not in hop and "both variants bit-identical" in rec["gather_verified"]
    else:   # nobody entered the grouped send / receive (the workers returned at once) and the headline record is intact
        assert "error" in hop and rec["one_hop_denoising_steps_per_s"] is None
        assert all("error" in res[r]["gather_ms"]["one_hop_send_recv"] for r in range(world))
        assert res[0]["gather_ms"]["one_hop_send_recv"]["skipped_by_agreement"] and "injected" in res[fail_rank]["gather_ms"]["one_hop_send_recv"]["error"]


def test_bench_watchdog_prints_the_headline_and_exits():
    """a hung collective in the sharded leg must not cost the bench line: the watchdog prints the record held so far and exits 0"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench._Watchdog(0.5, 0, {'metric': 'denoising-steps/s', 'value': 1.0}); time.sleep(30)" % str(root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    import json
    assert len(r.stdout.strip().splitlines()) == 1   # stdout carries exactly ONE line: the compact record (the full one goes to stderr + bench_full.json)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 1.0 and "watchdog" in rec["sharded_error"] and rec["cpu_baseline"] is None
    full = [ln for ln in r.stderr.splitlines() if ln.startswith("# bench_full ")]
    assert len(full) == 1 and "watchdog" in json.loads(full[0][len("# bench_full "):])["sharded"]["error"]
