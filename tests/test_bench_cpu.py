"""bench.py host-side contract that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_tuning_variables_unless_asked():
    """Process-global tuning hooks of the library (TV_* environment variables, transvae/hip/_lib.py) change the timed path:
    bench.py exits before touching the GPU when one is set and --allow-tuning-env is not given (VERDICT r03, weak point 13)."""
    env = dict(os.environ, TV_WGRAD_KX3="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "tuning variables are set" in (r.stderr + r.stdout) and "TV_WGRAD_KX3" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]       # no metric line


def test_bench_without_a_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device bench.py exits with a message and prints no metric line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if not k.startswith("TV_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
