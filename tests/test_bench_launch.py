"""`python bench.py --gpus N` as typed: the script starts its own ranks as child
processes under torch.distributed.run and relays rank 0's JSON line.  Runs on the CPU: the ranks are real
processes in a real gloo group, only the GPU work is left out (FQD_BENCH_SELFTEST=1)."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_command_is_the_drivers_own():
    cmd = _bench().launch_command(4, ["--gpus", "4", "--steps", "3"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_rank0_line_is_picked_out_of_the_noise():
    b = _bench()
    good = json.dumps({"metric": "m", "value": 1.5, "n_gpus": 2})
    text = "NCCL INFO banner\n{not json}\n" + good + "\n{\"other\": 1}\ntrailing\n"
    assert b.rank0_line(text) == good
    assert b.rank0_line("nothing here\n") is None


def test_gpus_2_as_typed_starts_its_own_ranks():
    env = dict(os.environ, FQD_BENCH_SELFTEST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                        # ONE line, the ranks' chatter is not relayed
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
