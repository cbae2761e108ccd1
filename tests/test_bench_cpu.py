"""bench.py without a GPU: it must refuse loudly (no CPU fallback may ever produce a line), for one rank and for the
self-launched multi-rank form alike."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=300)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the refusal cannot be shown")
    for args in (("--steps", "1", "--warmup", "0"), ("--gpus", "2", "--steps", "1", "--warmup", "0")):
        res = _run(*args)
        assert res.returncode != 0, args
        assert "MI355X" in res.stderr + res.stdout, (args, res.stderr[-400:])
        for line in res.stdout.splitlines():            # and no JSON line of results
            try:
                assert "value" not in json.loads(line)
            except ValueError:
                pass


def test_bench_flag_table_matches_the_contract():
    res = _run("--help")
    assert res.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--extras-timeout", "--no-workflow"):
        assert flag in res.stdout


def test_design_table_is_in_sync_with_the_bench_line_on_file():
    """DESIGN.md section 8's measured column is generated (scripts/design_table.py) from profiles/rNN_bench_line.json:
    a new bench line without a regenerated table fails here, so the document cannot go stale behind the numbers."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, os.path.join(root, "scripts", "design_table.py"), "--check"], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    assert os.path.getsize(os.path.join(root, "DESIGN.md")) <= 64 * 1024      # a current-state document, not a notebook
