"""Starting one rank per GPU on this node (train.sh:1 uses the same launcher module).  Deliberately torch-free: bench.py loads
this file by path from a parent process that must not touch the GPU before its child ranks exist."""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, script, argv, port=None, python=None):
    """The command line that starts `script argv...` as n ranks on this node, one per GPU (train.sh:1 uses the same module):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P script argv..."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def spawn_ranks(n, script, argv, env=None):
    """Run `script` as n fresh child ranks and return the launcher's exit code.  Must be called from a process that has NOT
    initialised the GPU (the children are new processes, never an exec of this one)."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(launcher_command(n, script, argv), env=e).returncode
