"""GPU: BASELINE config 5's control flow on the one-GPU box — two data-parallel ranks of the training loop sharing
cuda:0, gloo standing in for RCCL (which refuses two ranks on one device).  What is checked is what SURVEY.md §8(e)
asks of the multi-GPU path: per-rank env / replay shards, replicas identical after every exchange, the all-reduce placed
between the two HIP graphs of a sub-update (graph A: losses + backward into the flat bucket; graph B: 1/world, clip,
RMSprop), and that graphed and eager sub-updates take the same steps.  The ranks are child processes
(tools/dist_rehearsal.py under torch.distributed.run)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, graph_updates, nproc=2, extra=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.makedirs(tmp, exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_rehearsal.py"), "--out", tmp,
           "--graph-updates", str(int(graph_updates))] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return [dict(np.load(os.path.join(tmp, f"rank{k}.npz"))) for k in range(nproc)]


def test_two_ranks_train_with_split_update_graphs(tmp_path):
    g = _run(str(tmp_path / "graph"), True)
    e = _run(str(tmp_path / "eager"), False)
    for runs, graphed in ((g, True), (e, False)):
        a, b = runs
        assert int(a["steps"]) == int(b["steps"]) == 95
        assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["sq"], b["sq"])      # replicas bit-identical
        assert float(a["vloss"]) != float(b["vloss"]) and float(a["reward"]) != float(b["reward"])   # on different data
        if graphed:
            assert list(a["graphs"]) == ["policy", "value"] and a["split"].all()         # graph A | all-reduce | graph B ran
        else:
            assert len(a["graphs"]) == 0
    # graphed and eager sub-updates take the same steps (same kernels, same fixed-order reductions, x/2 == x*0.5)
    assert np.abs(g[0]["w"] - e[0]["w"]).max() <= 1e-6
    # and the exchange mattered: a lone rank from rank 0's start ends elsewhere
    solo = _run(str(tmp_path / "solo"), True, nproc=1)
    assert not np.array_equal(solo[0]["w"], g[0]["w"])


def test_ranks_that_choose_different_event_forms_recapture_stale_graphs_together(tmp_path):
    """Whether an update event reads filed bootstrap values is each rank's own choice (how much ITS windows overlap); recapturing
    a stale graph is not — its warm-up steps all-reduce.  Second episode after a re-allocation of the stacked ring, rank 0 made to
    choose the filed form and rank 1 the plain one: both finish, replicas bit-identical.  (Round 5's two-rank bench rehearsal died
    here once: rank 1 recaptured alone and its agreement met rank 0's gradient bucket.)"""
    a, b = _run(str(tmp_path / "forced"), True, extra=["--force-cached-rank", "0", "--envs", "2048"])   # (a batch the critic reads in place)
    assert int(a["steps"]) == int(b["steps"]) == 190
    assert int(a["ring_gen"]) == int(b["ring_gen"]) >= 2
    assert int(a["cached_events"]) > int(b["cached_events"])            # they did choose differently
    assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["sq"], b["sq"])


def test_safemaddpg_ranks_that_differ_in_their_first_cached_event(tmp_path):
    """The sequence the two-rank bench rehearsal died in (round 5): SAFEMADDPG's plain value graph is captured first and reads a
    gathered batch; the cached form's capture then creates the stacked ring, which makes the plain graph stale inside the same
    event.  Rank 0 goes on in the cached form, rank 1 in the plain one — which used to recapture alone."""
    a, b = _run(str(tmp_path / "safe"), True, extra=["--alg", "safemaddpg", "--envs", "2048", "--force-cached-rank", "0", "--force-from-start", "1"])
    assert int(a["steps"]) == int(b["steps"]) == 95
    assert int(a["cached_events"]) > 0 and int(b["cached_events"]) == 0
    assert list(a["graphs"]) == ["policy", "value"] and np.array_equal(a["w"], b["w"]) and np.array_equal(a["sq"], b["sq"])


def test_a_capture_that_fails_on_one_rank_moves_every_rank_to_eager_sub_updates(tmp_path):
    """Rank 1's first capture raises before its warm-up steps; rank 0's goes through (two warm-up all-reduces, then the
    agreement).  Rank 1 catches up with the all-reduces it missed (trainer._realign_after_failed_capture), the agreement says no,
    BOTH ranks run eager sub-updates from then on — and end bit-identical, on the weights two eager ranks reach anyway."""
    a, b = _run(str(tmp_path / "fail"), True, extra=["--fail-capture-rank", "1"])
    assert int(a["steps"]) == int(b["steps"]) == 95
    assert len(a["graphs"]) == 0 and len(b["graphs"]) == 0
    assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["sq"], b["sq"])
    e = _run(str(tmp_path / "eager"), False)
    assert np.array_equal(a["w"], e[0]["w"])            # the failed capture left no step of its own behind


@pytest.mark.parametrize("in_graph", ["1", "0"])
def test_update_graphs_next_to_a_live_rccl_communicator(in_graph):
    """RCCL itself (world size 1 — all this box allows): HIP-graph capture beside a live NCCL communicator and its watchdog
    thread for three training episodes.  in_graph = 1 (opt-in, FLEX_ALLREDUCE_IN_GRAPH=1): the trainer first tries ONE graph per
    sub-update with ncclAllReduce of the flat bucket captured inside it, and falls back to graph A | all-reduce | graph B
    if RCCL refuses the capture; in_graph = 0 is the default split form.  Either way the probe must finish and say
    which of the two ran."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLEX_ALLREDUCE_IN_GRAPH=in_graph)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1_probe.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "nccl world-1 probe ok" in r.stdout
    assert ("SPLIT_GRAPHS" in r.stdout) if in_graph == "0" else ("ALLREDUCE_IN_GRAPH" in r.stdout or "SPLIT_GRAPHS" in r.stdout)
    print(r.stdout.strip().splitlines()[-1])
