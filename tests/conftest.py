import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparse-document-attention-rag-_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# A helper process that starts other programs on request.  It is created here, before anything in this process has
# touched the GPU: a process that has initialised the GPU must not exec another program (and fork + exec from the test
# process would be exactly that), while this helper never touches the GPU, so what it starts begins from a clean state.
_HELPER_SRC = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    try:
        r = subprocess.run(req["cmd"], env=req["env"], capture_output=True, text=True, timeout=req["timeout"])
        out = {"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-4000:]}
    except subprocess.TimeoutExpired as e:
        out = {"rc": -999, "stdout": "", "stderr": "timeout: " + str(e)}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
"""
_helper = None


def run_in_fresh_process(cmd, env, timeout):
    """Run ``cmd`` through the helper; returns {"rc", "stdout", "stderr"} or None when there is no helper."""
    import json

    if _helper is None or _helper.poll() is not None:
        return None
    _helper.stdin.write(json.dumps({"cmd": cmd, "env": env, "timeout": timeout}) + "\n")
    _helper.stdin.flush()
    return json.loads(_helper.stdout.readline())


def pytest_configure(config):
    global _helper
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    import subprocess

    try:
        _helper = subprocess.Popen([sys.executable, "-c", _HELPER_SRC], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    except OSError:
        _helper = None


def pytest_unconfigure(config):
    global _helper
    if _helper is not None:
        try:
            _helper.stdin.close()
            _helper.wait(timeout=10)
        except Exception:
            _helper.kill()
        _helper = None


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
