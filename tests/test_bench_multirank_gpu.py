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
    # rank 0 prints the legs, one line each, then the headline as the LAST line (the one the driver parses: < 4 KB, benchlib/headline.py)
    out = json.loads(lines[-1])
    assert len(lines[-1].encode()) < 4096 and "leg" not in out
    legs = {d["leg"]: d["data"] for d in map(json.loads, lines[:-1])}
    assert all(k in out for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "parity"))
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert len(out["config"]["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in out["config"]["per_rank_ms_per_step"])
    assert abs(max(out["config"]["per_rank_ms_per_step"]) - out["ms_per_step"]) / out["ms_per_step"] < 1e-6        # the maximum over ranks
    assert out["config"]["rccl_ranks_seen"] == (0 if route == "torch_fallback" else 2)                              # ranks whose records came through the C++ gather
    assert out["config"]["gather"] is True
    assert out["config"]["gather_path"] == "torch.distributed.gather" if route == "torch_fallback" else out["config"]["gather_path"].startswith("cxx_host_transport")
    assert out["parity"]["bit_exact"] is True
    assert out["value"] > 0 and abs(out["value"] - 2 * 300000 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-6
    leg = legs["e2e_sharded_leg"]
    assert "error" not in leg, leg
    assert out["e2e_sharded"]["Mreads_per_s"] > 0
    assert leg["n_gpus"] == 2 and leg["gather"] is True and leg["gathered_records_verified"] is True
    assert leg["aligned"] > 0.9 and leg["best_at_true_position"] > 0.8
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_two_ranks_one_gpu_%s.json" % route), "w") as f:
        f.write(lines[-1] + "\n")
