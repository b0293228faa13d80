"""`python bench.py --gpus N` must drive all N GPUs by itself (one command, like the reference's base/base_trainer.py:33-38):
without WORLD_SIZE it re-executes under torch.distributed.run, one rank per GPU.  No GPU needed: the spawn is intercepted."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    return importlib.import_module("bench")


def test_launch_command_is_the_drivers_form(bench):
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29611"]
    assert cmd[10] == os.path.join(ROOT, "bench.py")
    assert cmd[11:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_main_reexecs_itself_when_not_under_a_launcher(bench, monkeypatch):
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 0

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cpu"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cpu"]


def test_rank_process_does_not_respawn(bench, monkeypatch):
    """Under the launcher (WORLD_SIZE set) main() must go on to the device checks, not spawn again."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd: pytest.fail("re-spawned inside a rank"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "MI355X" in str(e.value.code)
