"""bench.py at N > 1 rehearsed on ONE card: two and four ranks pinned to cuda:0 with the gloo backend (DYD_BENCH_DEVICE /
DYD_BENCH_BACKEND; the driver's real runs use RCCL, one rank per GPU).  Checks the line's contract and the sharded dedup /
reference filter of configs[3] (local pre-dedup, one all-gather of the locally unique keys, every rank settles its hash slice,
one all-reduce of the verdicts) against torch.unique — weak (rows per rank fixed) and strong (a fixed table cut into shards)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_card(world):
    import torch

    rows = 300_000 if world == 2 else 150_000
    env = dict(os.environ, DYD_BENCH_DEVICE="0", DYD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--rows", str(rows), "--ramp-ms", "10", "--strong-rows", "400000"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["steps"] == 3 and line["scaling"] == "weak" and line["unit"] == "rows/s"
    assert line["config"]["rows_per_gpu"] == rows and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["cpu_baseline"] is None and line["host_inclusive"] is None      # rank 0 at N = 1 only
    ex = line["sharded_exchange"]
    assert ex["world"] == world and ex["rows_total"] == world * rows and ex["backend"] == "gloo"
    # what the ranks drew (bench.py: generator seeds 900 + rank on the device)
    dev = torch.device("cuda:0")
    N = world * rows
    ids = torch.cat([torch.randint(0, int(0.9 * N) + 1, (rows,), generator=torch.Generator(device=dev).manual_seed(900 + r),
                                   device=dev, dtype=torch.int64) for r in range(world)])
    assert ex["kept_rows_total"] == int(torch.unique(ids).numel())              # drop_duplicates(keep="first") over all shards
    assert ex["ref_hits_total"] == int((ids % 10 == 0).sum().item())           # the reference set: every id divisible by 10
    assert ex["allgather_bytes_dedup"] == 16 * sum(int(torch.unique(ids[r * rows:(r + 1) * rows]).numel()) for r in range(world))
    assert ex["stages_ms_rank0"]["dedup_collectives_ms"] > 0 and ex["allreduce_bytes_dedup"] * 16 == ex["allgather_bytes_dedup"]
    # every rank settles about 1/world of the gathered keys, whatever its place in the row order
    per = ex["per_rank"]
    assert [p["rank"] for p in per] == list(range(world)) and sum(p["dedup_slice_keys"] for p in per) == ex["allreduce_bytes_dedup"]
    assert max(p["dedup_slice_keys"] for p in per) < 1.2 * ex["allreduce_bytes_dedup"] / world
    st = line["sharded_exchange_strong"]
    assert st["scaling"] == "strong" and st["rows_total"] == 400000 and st["rows_per_gpu"] == 400000 // world and st["world"] == world
    fs = line["fused_strong"]
    assert fs["scaling"] == "strong" and fs["rows_total"] == 400000 and fs["rows_per_gpu"] == 400000 // world and fs["rows_per_s"] > 0
    srows = 400000 // world
    ids = torch.cat([torch.randint(0, int(0.9 * 400000) + 1, (srows,), generator=torch.Generator(device=dev).manual_seed(900 + r),
                                   device=dev, dtype=torch.int64) for r in range(world)])
    assert st["kept_rows_total"] == int(torch.unique(ids).numel()) and st["ref_hits_total"] == int((ids % 10 == 0).sum().item())


def test_one_rank_runs_the_exchange_over_rccl():
    """the same code path at N = 1, but over RCCL itself: a group of one, device tensors in the collectives
    (all_gather_into_tensor of 16-byte keys, all_reduce of bytes) at configs[3]'s per-GPU size"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", "c4", "--steps", "2", "--warmup", "1", "--ramp-ms", "10",
           "--cpu-sample", "0", "--host-rows", "0", "--pipeline", "0", "--dense", "0", "--strong-rows", "20000000"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    ex, st = line["sharded_exchange"], line["sharded_exchange_strong"]
    assert ex["backend"] == "nccl" and ex["world"] == 1 and ex["rows_per_gpu"] == 12_500_000 and ex["scaling"] == "weak"
    assert st["backend"] == "nccl" and st["rows_total"] == 20_000_000 and st["scaling"] == "strong"
    for e in (ex, st):
        assert 0.55 * e["rows_total"] < e["kept_rows_total"] < 0.7 * e["rows_total"]          # ~40 % duplicate URLs
        assert 0.08 * e["rows_total"] < e["ref_hits_total"] < 0.12 * e["rows_total"]
