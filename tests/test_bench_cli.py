"""bench.py's command line, without a GPU: `python bench.py --gpus N` with no RANK in the environment starts its N ranks itself
under torch.distributed.run (the driver's contract wraps the script for N > 1; the plain form used to die on an assertion), and
the summary of the proposal stage's counts the line carries."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("gpn_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_multi_gpu_command_relaunches_itself(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("RANK", raising=False)
    args = bench.parse()
    assert bench.relaunch_under_torchrun(args) == 7, "the ranks' exit code is passed through"
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "5", "--warmup", "2"], "the script's own arguments are handed on unchanged"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_main_relaunches_only_without_a_rank(monkeypatch):
    bench = _bench()
    called = []
    monkeypatch.setattr(bench, "relaunch_under_torchrun", lambda args: called.append(args.gpus) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == [2]


def test_proposal_summary():
    bench = _bench()
    assert bench.proposal_summary([None, None]) is None
    s = bench.proposal_summary([None, [100, 18002, 487, 11211, 0, 0, 300], [90, 17000, 480, 11000, 0, 0, 290]])
    assert s["points_per_step"] == [18002, 17000] and s["proposals_min_max"] == [480, 487]
