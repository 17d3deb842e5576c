"""bench.py's N > 1 control flow on a 1-GPU box: two ranks launched the way the driver launches them (torch.distributed.run,
--nproc-per-node 2, rendezvous on 127.0.0.1) share the one device and talk over gloo (NVBIO_BENCH_SHARE_GPU=1: RCCL refuses two ranks
on one device).  Covers what a multi-GPU node would otherwise see first: rank / device set-up, the gather pre-flight and the ranks'
agreement on its outcome (the C++ / RCCL route declines here -- two ranks on one device -- and every rank must fall back to
torch.distributed.gather together), the double-buffered gather inside the timed loop, the max-over-ranks timing, and the sharded
end-to-end leg (replicated index, block-sharded reads, one gather of 16-byte alignment records to rank 0, record order verified) at a toy
genome.  A hang or a mis-ordered collective fails (or times out) here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("route", ["torch_fallback", "cxx_host_transport"])
def test_two_ranks_share_one_gpu_through_the_whole_bench(route):
    """route = cxx_host_transport: the C++ gather (nvbio_hip_gather_records, CxxRecordGather's tables, the e2e leg's record gather through
    DeviceGroup's entry point) runs between the two processes with only ncclSend / ncclRecv swapped for gloo -- what is left unexercised on a
    multi-GPU node is RCCL itself; route = torch_fallback: the C++ route declines and every rank falls back together."""
    env = dict(os.environ, NVBIO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    if route == "cxx_host_transport":
        env["NVBIO_BENCH_HOST_TRANSPORT"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--reads", "300000",
           "--genome", "4e6", "--e2e-reads", "40000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["gather"] is True
    assert out["config"]["gather_path"] == "torch.distributed.gather" if route == "torch_fallback" else out["config"]["gather_path"].startswith("cxx_host_transport")
    assert out["parity"]["bit_exact"] is True
    assert out["value"] > 0 and abs(out["value"] - 2 * 300000 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-6
    leg = out["e2e_sharded_leg"]
    assert "error" not in leg, leg
    assert leg["n_gpus"] == 2 and leg["gather"] is True and leg["gathered_records_verified"] is True
    assert leg["aligned"] > 0.9 and leg["best_at_true_position"] > 0.8
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_two_ranks_one_gpu_%s.json" % route), "w") as f:
        f.write(lines[0] + "\n")
