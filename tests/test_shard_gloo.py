"""N>1 host plumbing on CPU: world-size-2 gloo run of the sharding / gathering code that
bench.py --gpus N uses around the (GPU-only) hot path.  The per-rank compute here is the CPU
oracle -- this test checks the plumbing, not the kernels."""
import os
import subprocess
import sys
import textwrap

import helpers
from mtsv_tools_amd.shard import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_exactly():
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo_sharded_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch, torch.distributed as dist
        import helpers
        from oracle import oracle as O
        from mtsv_tools_amd.shard import shard_bounds, gather_lines
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        entries, gene, unit = helpers.tricky_db(seed=7)
        reads = helpers.tricky_reads(entries, gene, unit, seed=4, n_each=12)
        ix = O.Index.build(entries)
        lo, hi = shard_bounds(len(reads), rank, world)
        bases, off = helpers.reads_to_batch(reads[lo:hi])
        dist.barrier()
        hits, _ = ix.bin_batch(bases, off, threads=2)
        text = "".join(O.format_line(f"r{{lo + r}}", hits[hits["read"] == r]) for r in range(hi - lo))
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the max-over-ranks timing reduction
        assert t.item() == world
        joined = gather_lines(text, dist)
        if rank == 0:
            b2, o2 = helpers.reads_to_batch(reads)
            h2, _ = ix.bin_batch(b2, o2, threads=2)
            whole = "".join(O.format_line(f"r{{r}}", h2[h2["read"] == r]) for r in range(len(reads)))
            assert sorted(joined.splitlines()) == sorted(whole.splitlines()) and len(whole) > 100
            print("OK", len(whole.splitlines()))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OK" in out.stdout
