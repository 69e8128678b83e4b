"""Build libvivit_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvivit_hip.so")
SOURCES = ["gemm_f32.hip", "symeig_small.hip", "sytrd.hip", "sytrd_persist.hip", "sy2sb.hip", "sb2st.hip", "q2apply.hip", "q2slide.hip", "stedc.hip", "stein.hip", "symeig_large.hip", "elementwise.hip", "factors.hip", "jacobians.hip", "skinny.hip", "profile.hip", "api.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source to an object, link the shared library. Returns the library path."""
    headers = [os.path.join(CSRC, h) for h in ("common.h", "device_utils.h", "eig_internal.h")] + [
        os.path.join(HERE, "..", "include", "vivit_hip.h")]
    objdir = os.path.join(CSRC, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


ASAN_LIB = os.path.join(HERE, "libvivit_hip_hostasan.so")


def sanitizer_runtime():
    """Path of clang's shared AddressSanitizer runtime (to LD_PRELOAD into the python that loads ASAN_LIB)."""
    out = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_host_sanitized(verbose=False):
    """HOST side only (`--offload-host-only`: launch planning, argument checks, workspace carving; no device code) of every
    source with AddressSanitizer + UndefinedBehaviorSanitizer, for the CPU box: GPU sanitizers are not available on the
    pool.  The result can refuse calls and plan launches; any kernel launch through it fails.  Returns the library path."""
    objdir = os.path.join(CSRC, "obj_asan")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O1", "-g", "--offload-host-only", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
             "-fPIC", "-std=c++17", "-Wno-unused-function"]
    headers = [os.path.join(CSRC, h) for h in ("common.h", "device_utils.h", "eig_internal.h")] + [
        os.path.join(HERE, "..", "include", "vivit_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if _stale(o, [s] + headers):
            procs.append((src, subprocess.Popen([_hipcc()] + flags + ["-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc (host sanitizer build) failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if _stale(ASAN_LIB, objs):
        # a host-only object still refers to its (absent) device fat binary: define each of those symbols as an EMPTY
        # clang offload bundle (magic + zero entries), which the HIP runtime registers and never finds a kernel in
        undef = subprocess.run(["nm", "-u"] + objs, stdout=subprocess.PIPE, text=True).stdout.split()
        names = sorted({w for w in undef if w.startswith("__hip_fatbin_")})
        stub_c = os.path.join(objdir, "fatbin_stub.c")
        with open(stub_c, "w") as f:
            for nme in names:
                f.write('__attribute__((aligned(4096))) const char %s[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n' % nme)
        stub_o = os.path.join(objdir, "fatbin_stub.o")
        subprocess.check_call(["gcc", "-c", "-fPIC", stub_c, "-o", stub_o])
        subprocess.check_call([_hipcc(), "-shared", "-fPIC", "--offload-host-only", "-fsanitize=address,undefined", "-o", ASAN_LIB] + objs + [stub_o])
    return ASAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
