"""bench.py at N > 1 rehearsed on ONE card: two and four ranks pinned to cuda:0 with the gloo backend (DYD_BENCH_DEVICE /
DYD_BENCH_BACKEND; the driver's real runs use RCCL, one rank per GPU).  Checks the line's contract and the sharded dedup /
reference filter of configs[3] (local pre-dedup, one all-gather of the locally unique keys, probe) against torch.unique."""
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
           "--rows", str(rows), "--ramp-ms", "10"]
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
    assert ex["stages_ms_rank0"]["dedup_allgather_ms"] > 0
