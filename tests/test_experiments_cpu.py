"""Experiments kept out of the product build stay reproducible: the recorded patch of the sixteen-wave latency form
(tools/ubench/lat16, profiles/r06_latency_form.md) still applies to the tree, and nothing of it is in the library's source list."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sixteen_wave_experiment_is_not_in_the_product_build():
    from surfd_amd import build as B
    assert "conv_lat16.hip" not in B.SOURCES
    assert not os.path.exists(os.path.join(B.CSRC, "conv_lat16.hip")) and not os.path.exists(os.path.join(B.CSRC, "conv2_dev.h"))
    for f in ("conv_lat16.hip", "conv2_dev.h", "hook.patch", "README.md"):
        assert os.path.exists(os.path.join(ROOT, "tools", "ubench", "lat16", f)), f


def test_sixteen_wave_experiment_patch_still_applies():
    if not os.path.isdir(os.path.join(ROOT, ".git")) or not shutil.which("git"):
        pytest.skip("not a git checkout (the GPU box's snapshot)")
    r = subprocess.run(["git", "apply", "--check", "tools/ubench/lat16/hook.patch"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
