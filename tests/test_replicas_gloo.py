"""N > 1 path on CPU: two ranks over gloo run the replica logic bench.py uses on RCCL (cdv_slam_amd/replicas.py):
rank r <- sequence r, barrier + max-over-ranks timing, ONE all_gather of the per-rank metrics, no data-path collective.
Each rank advances its own synthetic sequence with the CPU oracle (test infrastructure) so that the gathered
trajectory checksums are real, rank-dependent numbers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sequence_metric(seed):
    """one BA update of the `tiny` synthetic sequence with this seed, on the CPU oracle -> pose checksum"""
    from cdv_slam_amd import synth
    from oracle import oracle as O
    st = synth.make_state("tiny", features=False, seed=seed)
    poses, _, info = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj,
                              st.kk, st.t0, st.n, 2, np.float64)
    assert info == 0
    return float(np.abs(poses).sum())


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from cdv_slam_amd.replicas import ReplicaGroup, aggregate_rate, summarise
    grp = ReplicaGroup(backend="gloo", device=torch.device("cpu"))
    assert (grp.rank, grp.world) == (rank, world)
    seed = grp.sequence_seed()
    metric = _sequence_metric(seed)
    grp.barrier()
    elapsed = 1.0 + 0.5 * rank               # pretend rank 1 was slower: the job takes as long as its slowest rank
    emax = grp.max_over_ranks(elapsed)
    rows = grp.gather_metrics([metric, 10.0 / elapsed])
    shard = grp.shard(range(5))
    grp.close()
    q.put((rank, seed, metric, emax, rows, shard, aggregate_rate(10, emax, world), summarise(rows)))


def test_two_replicas_gather_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, m0, e0, rows0, sh0, agg0, sum0), (r1, s1, m1, e1, rows1, sh1, agg1, sum1) = res
    assert (r0, r1) == (0, 1) and (s0, s1) == (1234, 1235)
    assert m0 != m1                                  # independent sequences
    assert e0 == e1 == 1.5                           # max over ranks
    assert rows0 == rows1                            # every rank holds the full gather
    assert rows0[0][0] == m0 and rows0[1][0] == m1   # row r is rank r's metric, bit-exact through the collective
    assert rows0[0][1] == 10.0 and rows0[1][1] == 10.0 / 1.5
    assert sh0 == [0, 2, 4] and sh1 == [1, 3]        # disjoint cover of the sequence list
    assert agg0 == agg1 == pytest.approx(2 * 10 / 1.5)
    assert sum0["ranks"] == 2 and sum0["max"][0] == max(m0, m1)


def test_single_replica_needs_no_process_group(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    from cdv_slam_amd.replicas import ReplicaGroup
    grp = ReplicaGroup(device=torch.device("cpu"))
    assert grp.world == 1 and grp.gather_metrics([1.0, 2.0]) == [[1.0, 2.0]]
    assert grp.max_over_ranks(3.0) == 3.0 and grp.shard([7, 8]) == [7, 8]
    grp.barrier()
    grp.close()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` run PLAINLY (no torch.distributed.run, no WORLD_SIZE): the parent starts two rank
    processes before touching any GPU and the gathered line says n_gpus 2 (dry run: gloo on CPU, trivial step -- the
    launch / rendezvous / gather plumbing of the N-GPU bench, not a measurement)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--dry-run"], env=env, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert len(line["per_rank"]) == 2 and line["per_rank"][0][0] != line["per_rank"][1][0]   # two different sequences
    # a world that does not match --gpus is an error, not a silent 1-GPU run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, timeout=120)
    assert bad.returncode != 0


def test_bench_parent_stops_the_group_when_a_rank_dies():
    """a rank that exits before the rendezvous: the parent notices, terminates the other rank (which would otherwise
    sit in init_process_group until torch's own timeout), exits non-zero and says where the per-rank logs are"""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--dry-run"], env=dict(env, CDV_BENCH_TEST_FAIL_RANK="1"), capture_output=True, timeout=200)
    assert out.returncode != 0
    assert time.perf_counter() - t0 < 120
    assert b"rank 1 exited with 3" in out.stderr and b"bench_rank" in out.stderr
