"""bench.py --gpus N without torchrun: the launcher starts N fresh rank processes with the torch.distributed.run
environment and relays rank 0's line.  A stub worker stands in for the GPU program (no GPU here)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / "stub_worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_bench_module_import_is_gpu_free():
    """importing bench (the launcher's process) must not import torch: the parent never touches HIP"""
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; print('ok')" % ROOT
    assert subprocess.check_output([sys.executable, "-c", code]).decode().strip() == "ok"


def test_launcher_rank_wiring(tmp_path, capsys):
    import bench
    worker = _stub(tmp_path, """
        import json, os, sys
        import torch.distributed as dist
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert int(os.environ["LOCAL_RANK"]) == r and os.environ["MASTER_ADDR"] == "127.0.0.1"
        dist.init_process_group("gloo")           # the rendezvous the ranks would use (MASTER_PORT picked by the launcher)
        import torch
        t = torch.ones(1); dist.all_reduce(t)
        dist.barrier(); dist.destroy_process_group()
        if r == 0:
            print("noise before the result")
            print(json.dumps({"n_gpus": w, "ranks_seen": int(t.item()), "argv": sys.argv[1:]}))
    """)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"], worker=worker, timeout=120)
    assert rc == 0
    line = capsys.readouterr().out.strip().splitlines()[-1]
    out = json.loads(line)
    assert out == {"n_gpus": 2, "ranks_seen": 2, "argv": ["--gpus", "2", "--steps", "3"]}


def test_launcher_propagates_failure(tmp_path, capsys):
    import bench
    worker = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(60)   # rank 0 would hang in a collective: the launcher must end it
    """)
    rc = bench.launch_ranks(2, [], worker=worker, timeout=30)
    assert rc == 3
    assert capsys.readouterr().out.strip() == ""


def test_launcher_keeps_and_echoes_every_ranks_stderr(tmp_path, capsys):
    """a rank that dies says why: its stderr is in a file of its own and is echoed with the failure"""
    import bench
    worker = _stub(tmp_path, """
        import os, sys, time
        r = os.environ["RANK"]
        sys.stderr.write("rank %s: starting\\n" % r)
        if r == "1":
            sys.stderr.write("rank 1: hipErrorNoDevice (pretend)\\n")
            sys.exit(7)
        time.sleep(60)
    """)
    logs = tmp_path / "ranks"
    rc = bench.launch_ranks(2, [], worker=worker, timeout=30, log_dir=str(logs))
    assert rc == 7
    cap = capsys.readouterr()
    assert cap.out.strip() == ""
    assert "rank 1 exited with code 7" in cap.err and "hipErrorNoDevice (pretend)" in cap.err and "rank 0: starting" in cap.err
    assert (logs / "rank0.err").read_text().startswith("rank 0: starting")
    assert "pretend" in (logs / "rank1.err").read_text()


def test_launcher_times_out_with_the_logs(tmp_path, capsys):
    import bench
    worker = _stub(tmp_path, """
        import os, sys, time
        sys.stderr.write("rank %s waits for a peer that never comes\\n" % os.environ["RANK"])
        sys.stderr.flush()
        time.sleep(120)
    """)
    rc = bench.launch_ranks(2, [], worker=worker, timeout=3)
    assert rc == 124
    err = capsys.readouterr().err
    assert "timed out" in err and err.count("waits for a peer") == 2


def test_bench_fails_cleanly_without_gpu():
    """`python bench.py --gpus 2 --backend gloo` on a box without a GPU: non-zero exit, no result line, no hang"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, timeout=300)
    assert p.returncode != 0
    assert b'"metric"' not in p.stdout
