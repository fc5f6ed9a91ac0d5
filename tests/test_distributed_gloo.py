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
