"""Builds libpmx_hip.so (the C ABI of include/pmx.h + the gfx950 kernels) in-tree with hipcc.

Every .hip source is compiled to its own object (in parallel, rebuilt only when it or a header is newer) and the objects
are linked into the shared library; a hash of the sources is linked in as `pmx_source_hash()` so that a loader can
tell a stale binary from the tree it sits in."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpmx_hip.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["pmx_step.hip", "pmx_api.hip", "pmx_train.hip", "pmx_actor.hip", "pmx_critic.hip", "pmx_heads.hip"]
HEADERS = ["pmx_device.h", os.path.join("..", "..", "include", "pmx.h")]
# -ffp-contract=off: the float64 reward sums and the float32 GAE scan must round like the reference's Python/torch ops
# (the network kernels opt back in per file with `#pragma clang fp contract(fast)`)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# per-file additions.  pmx_train.hip: MFMA results in ordinary VGPRs -- with accumulators placed in the AGPR half the one-pass
# attention backward moved every dQ tile to a VGPR and back around each matrix instruction (24 v_accvgpr_* of ~70 vector
# instructions per inner iteration on a kernel that is bound by its vector ALU)
FILE_FLAGS = {"pmx_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the pmx HIP library cannot be built")


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def source_hash():
    """sha256 over the kernel sources and headers (what pmx_source_hash() of a matching library returns, 16 hex digits)."""
    h = hashlib.sha256()
    for d in _deps()[:-1]:
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(repr((CFLAGS, sorted(FILE_FLAGS.items()))).encode())       # a changed compiler flag is a changed library too
    return h.hexdigest()[:16]


def stale():
    """True when the library is missing or was built from other sources than the ones in the tree (the hash of the
    sources it was built from sits beside it; file times do not survive a copy to another machine)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".hash"):
        return True
    with open(LIB + ".hash") as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile in-tree.  Serialised by a lock file and written through a temporary name, so that several ranks of one
    job importing the package at once (torch.distributed.run) cannot corrupt the library."""
    import fcntl
    if not force and not stale():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():          # another process built it while we waited
                return LIB
            cc = hipcc()
            os.makedirs(OBJDIR, exist_ok=True)
            hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
            hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
            jobs, objs = [], []
            for s in SOURCES:
                src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
                objs.append(obj)
                if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
                    cmd = [cc] + CFLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd))
                    jobs.append((s, subprocess.Popen(cmd, cwd=CSRC)))
            failed = [s for s, p in jobs if p.wait() != 0]
            if failed:
                raise RuntimeError("hipcc failed on " + ", ".join(failed))
            stamp = os.path.join(OBJDIR, "pmx_stamp.cpp")
            with open(stamp, "w") as f:
                f.write('extern "C" const char *pmx_source_hash(void) { return "%s"; }\n' % source_hash())
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [cc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + [stamp, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp, LIB)
            with open(LIB + ".hash", "w") as f:
                f.write(source_hash())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
