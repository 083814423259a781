"""Bounded multi-process launcher for the world_size>1 tests.

Workers are FRESH python processes (never forks of a process that has touched the GPU), each gets
`(rank, world, port, outdir)`, the rendezvous port is a free one handed out by the kernel, and the
whole group is killed -- and the test fails with the workers' output -- if it is not done within
`timeout` seconds.  The driver's GPU run of round 1 died on an unbounded `mp.spawn(join=True)`.
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(module: str, fn: str, world: int, outdir: str, timeout: float = 120.0, env=None):
    """Run `module.fn(rank, world, port, outdir)` in `world` fresh processes; raises on failure."""
    port = free_port()
    e = dict(os.environ)
    e.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GLOO_SOCKET_IFNAME": "lo",
              "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONUNBUFFERED": "1",
              "F2G_DIST_TIMEOUT_S": "60"})
    e.update(env or {})
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import %s as m; "
            "m.%s(int(sys.argv[1]), %d, %d, %r)"
            % (ROOT, os.path.join(ROOT, "oracle"), HERE, module, fn, world, port, outdir))
    procs, logs = [], []
    for r in range(world):
        log = open(os.path.join(outdir, f"worker{r}.log"), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, "-c", code, str(r)], env=e, stdout=log,
                                      stderr=subprocess.STDOUT, start_new_session=True))
    deadline = time.time() + timeout
    failed = None
    try:
        while any(p.poll() is None for p in procs):
            bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:
                failed = f"worker {bad[0]} exited with {procs[bad[0]].returncode}"
                break
            if time.time() > deadline:
                failed = f"workers not done after {timeout:.0f}s"
                break
            time.sleep(0.05)
        if failed is None and any(p.returncode != 0 for p in procs):
            failed = "exit codes %s" % [p.returncode for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()            # exactly the PIDs started here
                p.wait()
    if failed:
        tails = []
        for r, log in enumerate(logs):
            log.seek(0)
            tails.append(f"--- worker {r} ---\n" + log.read()[-3000:])
        raise AssertionError(failed + "\n" + "\n".join(tails))
    for log in logs:
        log.close()


def _preflight(rank, world, port, outdir):
    import datetime

    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=30))
    t = torch.full((4,), float(rank + 1))
    dist.all_reduce(t)
    assert float(t[0]) == 3.0
    dist.destroy_process_group()


def gloo_rendezvous_works(outdir: str) -> bool:
    """Can two fresh CPU-only processes on this box meet over gloo on 127.0.0.1 at all?  Purely
    environmental (no package code, no GPU): the GPU two-rank test skips -- with this reason --
    when the box cannot, instead of reporting an environment problem as a parity failure."""
    try:
        run_workers("_mp", "_preflight", 2, outdir, timeout=60.0)
        return True
    except AssertionError as e:
        sys.stderr.write("[f2g] gloo preflight failed: %s\n" % str(e)[:2000])
        return False
