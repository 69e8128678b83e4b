"""Build libvivit_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

Provenance: the library exports ``vivit_hip_source_hash()`` = ``"<source_hash()>-<flags_digest(flags)>"``: the content
hash of the tree it was compiled from (every file of csrc/ + include/vivit_hip.h) and the digest of the compile flags of
ALL its objects.  The function lives in an object generated at link time (csrc/obj/link/buildinfo.c), outside csrc/obj/,
so a library relinked by hand from product objects plus one compiled with other defines (the timing-only variants of
scripts/probe/) has no such symbol unless its script supplies one -- it cannot report the product's hash.
``_lib.load()`` refuses a library whose hash differs from the sources beside it or whose flags are not a known build's,
so neither a stale binary nor a variant can be tested by accident.  Objects are rebuilt when the content hash of
(source, headers, flags) recorded beside them differs -- not by modification time.
"""
import glob
import hashlib
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


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _headers():
    # (csrc/bx_kloop_asm.inc: the generated asm K loop; the experiment blocks of bx_kloop_asm_variants.inc are not part of the product)
    return (sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(CSRC, "bx_kloop_asm.inc")]
            + [os.path.join(HERE, "..", "include", "vivit_hip.h")])


def source_hash():
    """Content hash (32 hex digits) of everything the library is compiled from: csrc/*.hip, csrc/*.h, csrc/bx_kloop_asm.inc and
    include/vivit_hip.h.  None when the sources are not beside the package (a binary-only install)."""
    hips = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not hips or not os.path.exists(os.path.join(HERE, "..", "include", "vivit_hip.h")):
        return None
    return _digest(hips + _headers())[:32]


def flags_digest(flags):
    """8 hex digits naming a set of compile flags (second half of ``vivit_hip_source_hash()``)."""
    return hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8]


def _buildinfo_object(objdir, flags, extra_cflags=()):
    """Compile the link-time object that defines ``vivit_hip_source_hash()`` for a library whose objects were ALL built
    by this module with ``flags`` (every stale object has just been recompiled, the stamps say so)."""
    linkdir = os.path.join(objdir, "link")
    os.makedirs(linkdir, exist_ok=True)
    text = ('const char *vivit_hip_source_hash(void) { return "%s-%s"; }\n' % (source_hash(), flags_digest(flags)))
    c, o = os.path.join(linkdir, "buildinfo.c"), os.path.join(linkdir, "buildinfo.o")
    try:
        with open(c) as f:
            same = f.read() == text
    except OSError:
        same = False
    if not same or not os.path.exists(o):
        with open(c, "w") as f:
            f.write(text)
        subprocess.check_call(["gcc", "-c", "-fPIC"] + list(extra_cflags) + [c, "-o", o])
        return o, True
    return o, False


def _stamp_stale(obj, stamp):
    """True when ``obj`` is missing or was not built from the inputs whose digest is ``stamp``."""
    try:
        with open(obj + ".stamp") as f:
            return not os.path.exists(obj) or f.read().strip() != stamp
    except OSError:
        return True


def build(force=False, verbose=True):
    """Compile every HIP source to an object, link the shared library. Returns the library path."""
    headers = _headers()
    objdir = os.path.join(CSRC, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        flags = FLAGS
        stamp = _digest([s] + headers, flags)
        if force or _stamp_stale(o, stamp):
            cmd = [_hipcc()] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, o, stamp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, o, stamp, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}", file=sys.stderr)
        else:
            with open(o + ".stamp", "w") as f:
                f.write(stamp + "\n")
    if failed:
        raise RuntimeError("hipcc compilation failed")
    info, fresh = _buildinfo_object(objdir, FLAGS)
    if force or procs or fresh or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + [info]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


ASAN_LIB = os.path.join(HERE, "libvivit_hip_hostasan.so")
ASAN_FLAGS = ["-O1", "-g", "--offload-host-only", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
              "-fPIC", "-std=c++17", "-Wno-unused-function"]


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
    flags = ASAN_FLAGS
    headers = _headers()
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        fl = flags
        stamp = _digest([s] + headers, fl)
        if _stamp_stale(o, stamp):
            procs.append((src, o, stamp, subprocess.Popen([_hipcc()] + fl + ["-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, o, stamp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc (host sanitizer build) failed on {src}:\n{out}")
        with open(o + ".stamp", "w") as f:
            f.write(stamp + "\n")
        if verbose and out.strip():
            print(out)
    info, fresh = _buildinfo_object(objdir, flags)
    if procs or fresh or _stale(ASAN_LIB, objs):
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
        subprocess.check_call([_hipcc(), "-shared", "-fPIC", "--offload-host-only", "-fsanitize=address,undefined", "-o", ASAN_LIB] + objs + [stub_o, info])
    return ASAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
