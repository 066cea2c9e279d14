"""RCCL touches hardware inside the 1-GPU lease: tools/rccl_world1_check.py in its own process (a process group is
process-global state), world_size 1, backend "nccl"."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_broadcast_arena_and_reductions_leave_generation_unchanged():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_check.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    print(out[-2000:])
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in out, out[-4000:]
