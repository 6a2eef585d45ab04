"""bench.py --gpus N started without a launcher re-runs itself under torch.distributed.run (CPU check of the command;
the GPU suite runs it for real: tests/test_gpu_bench_launch.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_command_is_the_drivers_launch_line(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "7", "--warmup", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_above_one_without_world_size_takes_the_spawn_path(monkeypatch):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "5"])
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv=None: 17 + n)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 19
    else:
        raise AssertionError("main() should have exited with the launcher's code")
