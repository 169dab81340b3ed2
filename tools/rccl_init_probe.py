#!/usr/bin/env python3
"""GPU box: does a one-rank ncclCommInitRank return?  Runs tools/micro/rccl_init_repro (built here with gcc) `--trials` times per
RCCL library under NCCL_DEBUG=INFO, each under a timeout; when a trial hangs, every thread's kernel-side state (comm, wchan, syscall,
kernel stack) is dumped from /proc before the process group is killed.  Libraries tried: the system RCCL (what a Node rank loads:
librccl.so.1 from /opt/rocm/lib) and, for comparison, the one bundled with torch (what bench.py's ranks share with torch).
Usage: python tools/rccl_init_probe.py [--trials 3] [--timeout 40] [--env K=V ...] > profiles/r03/rccl_init_probe.txt"""
import argparse
import glob
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def thread_dump(pid):
    out = []
    for t in sorted(glob.glob("/proc/%d/task/*" % pid)):
        def rd(name):
            try:
                return open(os.path.join(t, name)).read().strip()
            except OSError as ex:
                return "<%s>" % ex.strerror
        out.append("  tid %s comm=%s wchan=%s syscall=%s" % (os.path.basename(t), rd("comm"), rd("wchan"), rd("syscall")[:60]))
        st = rd("stack")
        if st and not st.startswith("<"):
            out += ["      " + l for l in st.splitlines()[:8]]
    try:
        fds = []
        for fd in sorted(os.listdir("/proc/%d/fd" % pid), key=int):
            try:
                fds.append("%s->%s" % (fd, os.readlink("/proc/%d/fd/%s" % (pid, fd))))
            except OSError:
                pass
        out.append("  fds: " + " ".join(fds)[:1500])
        out.append("  tcp listen/conn (/proc/net/tcp, state 0A = LISTEN):")
        for l in open("/proc/%d/net/tcp" % pid).read().splitlines()[1:12]:
            out.append("      " + " ".join(l.split()[:4]))
    except OSError:
        pass
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--timeout", type=float, default=40.0)
    ap.add_argument("--env", action="append", default=[], help="extra K=V for the child (e.g. NCCL_SOCKET_IFNAME=lo)")
    ap.add_argument("--libs", default="system,torch")
    a = ap.parse_args()
    exe = "/tmp/rccl_init_repro"
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tools", "micro", "rccl_init_repro.c"), "-ldl"])
    libs = {}
    if "system" in a.libs:
        libs["system"] = ("librccl.so.1", "libamdhip64.so")
    if "torch" in a.libs:
        try:
            import torch
            tl = os.path.join(os.path.dirname(torch.__file__), "lib")
            hip = sorted(glob.glob(os.path.join(tl, "libamdhip64.so*")))
            rc = sorted(glob.glob(os.path.join(tl, "librccl.so*")))
            if hip and rc:
                libs["torch"] = (rc[0], hip[0])
        except Exception as ex:
            print("# torch not importable: %s" % ex)
    print("# hostname %s; interfaces: %s" % (os.uname().nodename, " ".join(sorted(os.listdir("/sys/class/net")))))
    print("# /etc/hosts: " + " | ".join(l.strip() for l in open("/etc/hosts") if l.strip() and not l.startswith("#")))
    sys.stdout.flush()
    for name, (rccl, hip) in libs.items():
        for k in range(a.trials):
            env = dict(os.environ, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,ENV,NET,BOOTSTRAP", HSA_ENABLE_IPC_MODE_LEGACY="0")
            for kv in a.env:
                key, _, val = kv.partition("=")
                env[key] = val
            if name == "torch":
                env["LD_LIBRARY_PATH"] = os.path.dirname(rccl) + ":" + env.get("LD_LIBRARY_PATH", "")
            t0 = time.time()
            p = subprocess.Popen([exe, rccl, hip], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, start_new_session=True)
            try:
                out, _ = p.communicate(timeout=a.timeout)
                verdict = "returned rc %d in %.1f s" % (p.returncode, time.time() - t0)
                dump = ""
            except subprocess.TimeoutExpired:
                dump = thread_dump(p.pid)
                os.killpg(p.pid, signal.SIGKILL)
                out, _ = p.communicate()
                verdict = "HUNG: no return within %.0f s" % a.timeout
            text = out.decode(errors="replace")
            print("== %s RCCL (%s), trial %d %s: %s" % (name, rccl, k + 1, " ".join(a.env), verdict))
            lines = text.splitlines()
            keep = lines if (dump or k == 0) else [l for l in lines if l.startswith("[repro")]
            print("\n".join("   " + l for l in keep[-120:]))
            if dump:
                print("  -- threads at the time of the kill --")
                print(dump)
            sys.stdout.flush()


if __name__ == "__main__":
    main()
