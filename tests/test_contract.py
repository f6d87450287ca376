"""The driver-facing files: bench.py's JSON line and __graft_entry__."""

import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
           "--seconds-per-shot", "0.01", *extra]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=120,
                          env={**os.environ, **(env or {})})


def test_bench_prints_one_json_line_and_claims_no_gpu_number():
    proc = _bench()
    assert proc.returncode == 0, proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in out
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        assert out["metric"] == json.load(fh)["metric"]
    assert out["value"] is None and out["roofline"] is None and out["vs_baseline"] is None
    assert out["gpus_used"] == 0
    assert (out["steps"], out["warmup"], out["n_gpus"]) == (3, 1, 1)
    assert out["ms_per_step"] >= 10.0  # one 10 ms actuation per step at least
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1


def test_bench_is_silent_on_ranks_other_than_zero():
    proc = _bench("--gpus", "2", env={"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert proc.returncode == 0 and proc.stdout.strip() == ""
    proc = _bench("--gpus", "2", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert json.loads(proc.stdout)["n_gpus"] == 2


def test_bench_does_not_import_torch():
    code = ("import sys; sys.argv=['bench.py','--steps','1','--warmup','0','--seconds-per-shot','0'];"
            f"sys.path.insert(0, {ROOT!r}); import bench; bench.main(); "
            "assert 'torch' not in sys.modules")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr


def test_graft_entry_build_and_smoke(capsys):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry

    assert entry.device_artifacts() == []
    entry.build()
    entry.smoke()
    text = capsys.readouterr().out
    assert "build:" in text and "smoke:" in text


def _pytest(args, cwd):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTEST_")}
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", *args],
                          cwd=cwd, env=env, capture_output=True, text=True, timeout=120)


def _tree_with_conftest(tmp_path):
    """``<tmp>/repo/tests/`` holding a copy of this suite's conftest."""
    import shutil

    tests = tmp_path / "repo" / "tests"
    tests.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "conftest.py"), tests / "conftest.py")
    return tests


def test_gpu_selection_is_empty_and_reports_success(tmp_path):
    # no test carries the gpu marker; an empty selection under exactly `-m gpu`
    # exits 0, anything else that selects nothing keeps pytest's status 5
    tests = _tree_with_conftest(tmp_path)
    (tests / "test_cpu.py").write_text("def test_ok():\n    assert True\n")
    proc = _pytest(["-m", "gpu", str(tests)], tests)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "1 deselected" in proc.stdout and "no GPU tests exist by design" in proc.stdout
    assert _pytest(["-k", "nothing_has_this_name", str(tests)], tests).returncode == 5


def test_gpu_selection_still_fails_on_a_collection_error(tmp_path):
    tests = _tree_with_conftest(tmp_path)
    (tests / "test_broken.py").write_text("def test_(:\n")
    proc = _pytest(["-m", "gpu", str(tests)], tests)
    assert proc.returncode not in (0, 5), proc.stdout + proc.stderr


def test_gpu_selection_of_nothing_fails_when_nothing_was_collected(tmp_path):
    # an empty directory or a wrong path is not "no GPU tests by design"
    tests = _tree_with_conftest(tmp_path)
    assert _pytest(["-m", "gpu", str(tests)], tests).returncode == 5


def test_gpu_selection_of_nothing_fails_once_the_tree_holds_device_code(tmp_path):
    # with a kernel in the tree, a GPU test that lost its marker must not pass
    # as "nothing to run"
    tests = _tree_with_conftest(tmp_path)
    (tests / "test_cpu.py").write_text("def test_ok():\n    assert True\n")
    csrc = tmp_path / "repo" / "pkg" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "kernel.hip").write_text("// device code\n")
    proc = _pytest(["-m", "gpu", str(tests)], tests)
    assert proc.returncode == 5, proc.stdout + proc.stderr
    assert "kernel.hip" in proc.stderr


def test_no_test_in_the_suite_is_marked_gpu(request):
    marked = [item.nodeid for item in request.session.items if item.get_closest_marker("gpu")]
    assert marked == []
