"""World-size-2 gloo test of the N>1 path: shard-by-index partition, barrier + MAX-over-ranks timing, digest gather.
The per-rank prover here is the CPU oracle on a tiny circuit (tests may use it); on the GPU box bench.py runs the same
plumbing with the HIP prover and backend nccl (= RCCL)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import __graft_entry__ as g
pkg = g.load_package()
import oracle_lib, circuits
from plonky2_aes_amd import dist as pd
d = pd.init("gloo")
rank, world = d.get_rank(), d.get_world_size()
total = 5
lo, hi = pd.shard_range(total, rank, world)
data, pws = circuits.assert_byte(pkg, [10 * i + 1 for i in range(total)])
oc = oracle_lib.OracleCircuit(data.blob)
d.barrier(); t0 = time.perf_counter()
proofs = []
for i in range(lo, hi):
    st, p = oc.prove(pws[i].map)
    assert st == 0
    proofs.append((i, p))
d.barrier(); dt = time.perf_counter() - t0
tmax = pd.max_over_ranks(d, dt)
assert tmax >= dt
dig = pd.gather_digests(d, proofs)
if rank == 0:
    print("RESULT " + json.dumps({"digests": dig, "tmax": tmax, "ranges": [pd.shard_range(total, r, world) for r in range(world)]}))
d.destroy_process_group()
'''


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from plonky2_aes_amd.dist import shard_range
    for total in (0, 1, 7, 32, 256, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharded_proving(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0]
    import json
    res = json.loads(line[7:])
    assert [i for i, _ in res["digests"]] == [0, 1, 2, 3, 4]          # every proof index exactly once
    assert res["ranges"] == [[0, 3], [3, 5]]
    # single-process reference: same digests whatever the sharding
    import __graft_entry__ as g
    pkg = g.load_package()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import circuits
    import oracle_lib
    data, pws = circuits.assert_byte(pkg, [10 * i + 1 for i in range(5)])
    oc = oracle_lib.OracleCircuit(data.blob)
    ref = [hashlib.sha256(oc.prove(pw.map)[1]).hexdigest() for pw in pws]
    assert [d for _, d in res["digests"]] == ref
