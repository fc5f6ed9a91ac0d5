"""world_size-2 gloo test of the multi-rank path bench.py uses for N > 1: frame
pairs are sharded across ranks with no data-path collective; the only exchanges
are the barrier, the MAX of the elapsed time and a small gather of summaries.
The per-pair compute is stubbed with the CPU oracle (the checker), so the test
proves that sharding + gathering reproduces the single-process result."""
import hashlib
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_ranges_partition_everything():
    from oflk_dist import shard_range

    for total in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


WORKER = textwrap.dedent("""
    import hashlib, json, sys, time
    sys.path.insert(0, {product!r}); sys.path.insert(0, {oracle!r})
    import numpy as np
    from oflk_dist import Group, shard_range
    from oflk_synth import synth_pair
    import oflk_oracle as O
    g = Group("gloo")
    assert g.world == 2
    B = 5
    b0, b1 = shard_range(B, g.rank, g.world)
    g.barrier()
    t0 = time.perf_counter()
    mine = {{}}
    for i in range(b0, b1):
        p, c = synth_pair(48, 64, pair_index=i)
        u, v = O.lucas_kanade_pyramidal(p, c, 2, 5, 2)
        mine[i] = hashlib.sha256((u + np.float32(0)).tobytes() + (v + np.float32(0)).tobytes()).hexdigest()
    el = time.perf_counter() - t0 + 0.25 * g.rank       # rank 1 is "slower"
    mx = g.max_over_ranks(el)
    tot = g.sum_over_ranks(b1 - b0)
    allr = g.gather_objects(mine)
    if g.rank == 0:
        merged = {{}}
        for d in allr: merged.update(d)
        print("RESULT " + json.dumps({{"max": mx, "own": el, "total": tot, "hashes": merged}}))
    g.close()
""")


def test_two_rank_gloo_sharding_reproduces_single_process(tmp_path, oracle):
    from oflk_synth import synth_pair

    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(product=str(ROOT / "optical-flow-fpga_amd" / "python"), oracle=str(ROOT / "oracle")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    import json

    res = json.loads(line[len("RESULT "):])
    assert res["total"] == 5
    assert res["max"] >= res["own"] + 0.2  # MAX over ranks picked the slower rank
    for i in range(5):
        p, c = synth_pair(48, 64, pair_index=i)
        u, v = oracle.lucas_kanade_pyramidal(p, c, 2, 5, 2)
        h = hashlib.sha256((u + np.float32(0)).tobytes() + (v + np.float32(0)).tobytes()).hexdigest()
        assert res["hashes"][str(i)] == h


# ---------------------------------------------------------------------------
# bench.py's own layout / timed region / aggregation code (oflk_dist.job_layout, run_timed,
# job_throughput) under a 2-rank gloo group: what `bench.py --gpus N` executes around the plan calls
# ---------------------------------------------------------------------------
BENCH_WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, {product!r}); sys.path.insert(0, {oracle!r})
    import numpy as np
    from oflk_dist import Group, env_rank, job_layout, job_throughput, run_timed
    from oflk_synth import synth_pair
    import oflk_oracle as O
    rank, local_rank, world = env_rank()
    g = Group("gloo")
    out = {{}}
    for cfg, pairs in (("1080p", 2), ("4k64", None)):
        lay = job_layout(cfg, rank, world, pairs, 48, 64)     # small frames: the oracle stands in for the plan
        if cfg == "4k64":
            lay = job_layout(cfg, rank, world, None, 24, 32)
        calls = []
        acc = {{"abs_u": 0.0}}
        def step():
            calls.append(1)
            if rank == 1: time.sleep(0.02)                    # rank 1 is the slower one
        def sync(): pass
        marks = []
        el = run_timed(g, step, sync, steps=3, warmup=2, before_timed=lambda: marks.append(len(calls)))
        for b in range(lay.pairs_local):
            p, c = synth_pair(lay.height, lay.width, pair_index=lay.pair_begin + b)
            u, v = O.lucas_kanade_pyramidal(p, c, 2, 5, 1)
            acc["abs_u"] += float(np.abs(u).sum(dtype=np.float64))
        job = job_throughput(g, lay, 3, el, acc)
        out[cfg] = {{"layout": [lay.pairs_local, lay.pair_begin, lay.pairs_total, lay.scaling], "calls": len(calls),
                    "warm": marks[0], "elapsed_max": el, "job": job}}
    if rank == 0:
        print("RESULT " + json.dumps(out))
    g.close()
""")


def test_bench_layout_timing_and_aggregation_two_ranks(tmp_path, oracle):
    import json

    from oflk_dist import job_layout
    from oflk_synth import synth_pair

    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER.format(product=str(ROOT / "optical-flow-fpga_amd" / "python"), oracle=str(ROOT / "oracle")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29573", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    # weak: every rank its own 2 pairs; strong: the 64-pair job cut in two
    assert res["1080p"]["layout"] == [2, 0, 4, "weak"] and res["4k64"]["layout"] == [32, 0, 64, "strong"]
    for cfg, (h, w) in (("1080p", (48, 64)), ("4k64", (24, 32))):
        r = res[cfg]
        assert r["calls"] == 5 and r["warm"] == 2                      # W untimed, then exactly K timed steps
        assert r["elapsed_max"] >= 3 * 0.02                            # MAX over ranks: the slower rank's time
        total = r["job"]["pairs_per_step"]
        assert total == (4 if cfg == "1080p" else 64)
        assert abs(r["job"]["Mpix_per_s"] - total * h * w * 3 / r["elapsed_max"] / 1e6) < 1e-6 * r["job"]["Mpix_per_s"]
        # the SUM over ranks of per-rank totals equals one process doing the whole job
        want = 0.0
        for i in range(total):
            p, c = synth_pair(h, w, pair_index=i)
            u, _ = oracle.lucas_kanade_pyramidal(p, c, 2, 5, 1)
            want += float(np.abs(u).sum(dtype=np.float64))
        assert abs(r["job"]["sums"]["abs_u"] - want) <= 1e-9 * max(want, 1.0)
    # the layouts of all ranks tile the job
    for world in (1, 2, 4, 8):
        spans = [job_layout("4k64", r, world) for r in range(world)]
        assert [s.pair_begin for s in spans] == [sum(x.pairs_local for x in spans[:i]) for i in range(world)]
        assert sum(s.pairs_local for s in spans) == 64


def test_bench_refuses_a_world_size_mismatch():
    """`--gpus N` under a launcher that started another number of ranks used to warn and report a different
    job.  (With NO launcher in the environment bench.py starts its own N ranks: tests/test_gpu_round2.py.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()


# ---------------------------------------------------------------------------
# optional gather of the flow shards (bench.py --gather): one collective, after the timed region
# ---------------------------------------------------------------------------
GATHER_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, {product!r})
    import torch
    from oflk_dist import Group, env_rank, gather_flows, job_layout
    rank, local_rank, world = env_rank()
    g = Group("gloo")
    out = {{}}
    for cfg, pairs in (("1080p", 3), ("4k64", None)):
        lays = [job_layout(cfg, r, world, pairs, 6, 8) for r in range(world)]
        if cfg == "4k64":
            lays = [job_layout(cfg, r, world, None, 6, 8) for r in range(world)]
            lays[1].pairs_local -= 1          # a ragged job: the second shard one pair shorter (padding path)
        me = lays[rank]
        idx = torch.arange(me.pair_begin, me.pair_begin + me.pairs_local, dtype=torch.float32)
        u = idx[:, None, None].expand(me.pairs_local, 6, 8).contiguous()          # pair k's u is k everywhere
        v = -u
        r = gather_flows(g, u, v, lays, lambda: None)
        if rank == 0:
            out[cfg] = {{"pairs": r["pairs"], "ms": r["gather_ms"], "bytes": r["bytes_received"],
                        "u_first": [float(x) for x in r["u"][:, 0, 0]], "v_ok": bool(torch.equal(r["v"], -r["u"])),
                        "shape": list(r["u"].shape)}}
    if rank == 0:
        print("RESULT " + json.dumps(out))
    g.close()
""")


def test_gather_of_flow_shards_two_ranks(tmp_path):
    import json

    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER.format(product=str(ROOT / "optical-flow-fpga_amd" / "python")))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29575", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    # weak config: 3 pairs per rank, job order 0..5; bytes that crossed ranks = the second rank's shard
    assert res["1080p"]["pairs"] == 6 and res["1080p"]["u_first"] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0] and res["1080p"]["v_ok"]
    assert res["1080p"]["bytes"] == 2 * 3 * 6 * 8 * 4 and res["1080p"]["shape"] == [6, 6, 8]
    # ragged: 32 + 31 pairs, the padding of the shorter shard never shows up
    assert res["4k64"]["pairs"] == 63 and res["4k64"]["u_first"] == [float(i) for i in range(63)] and res["4k64"]["v_ok"]
