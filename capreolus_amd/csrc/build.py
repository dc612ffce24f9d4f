"""Builds the C-ABI libraries for gfx950 with hipcc (cross-compiles without a GPU):

    libcapreolus_amd.so        the product library (include/capreolus_amd.h; no profiling hooks, no mutable global state)
    libcapreolus_amd_prof.so   the same objects, with the sources that carry profiling hooks (csrc/capamd_profiling.h) compiled
                               once more with -DCAPAMD_PROFILING - bound by bench.py's per-pass timing legs and scripts/ only
                               (capreolus_amd._lib.profiling_build())
    libcapamd_pyhost.so        pyhost.c: the CPython helper that turns a run's fp16 score vector into the {qid: {docid: score}}
                               dictionaries `predict` returns (gcc against the interpreter's headers; no device code, not part of the
                               C-ABI; capreolus_amd/pyhost.py)

Incremental: an object is rebuilt when its source or anything in its depfile (-MMD) is newer than it.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)          # hazard_lint, when build.py is imported as capreolus_amd.csrc.build
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcapreolus_amd.so")
OUT_PROF = os.path.join(HERE, "libcapreolus_amd_prof.so")
OUT_PYHOST = os.path.join(HERE, "libcapamd_pyhost.so")
PYHOST_SRC = os.path.join(HERE, "pyhost.c")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + HERE]
PROF_SOURCES = ("bert.hip", "lists.hip", "pacrr.hip")     # sources whose code differs under -DCAPAMD_PROFILING


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _deps(obj, src):
    d = obj + ".d"
    if not os.path.exists(d):
        return None
    words = open(d).read().replace("\\\n", " ").split()
    return [w for w in words[1:] if w != src] + [src]


def _obj_stale(obj, src):
    if not os.path.exists(obj):
        return True
    deps = _deps(obj, src)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _jobs():
    """[(source, object, extra flags)]: every source for the product library, PROF_SOURCES once more for the profiling one"""
    jobs = []
    for src in sources():
        jobs.append((src, src[:-4] + ".o", []))
        if os.path.basename(src) in PROF_SOURCES:
            jobs.append((src, src[:-4] + ".prof.o", ["-DCAPAMD_PROFILING"]))
    return jobs


def stale():
    return (not os.path.exists(OUT)) or (not os.path.exists(OUT_PROF)) or any(_obj_stale(o, s) for s, o, _ in _jobs()) or any(
        os.path.getmtime(o) > min(os.path.getmtime(OUT), os.path.getmtime(OUT_PROF)) for _, o, _ in _jobs() if os.path.exists(o))


def build_pyhost(force=False, verbose=False):
    import sysconfig

    if not force and os.path.exists(OUT_PYHOST) and os.path.getmtime(OUT_PYHOST) >= os.path.getmtime(PYHOST_SRC):
        return OUT_PYHOST
    cmd = ["gcc", "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], PYHOST_SRC, "-o", OUT_PYHOST]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT_PYHOST


def build(force=False, verbose=False):
    build_pyhost(force, verbose)
    if not force and not stale():
        return OUT
    procs, rebuilt = [], []
    for src, obj, extra in _jobs():
        if not force and not _obj_stale(obj, src):
            continue
        cmd = ["hipcc"] + CFLAGS + extra + ["-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        rebuilt.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    # the inline-assembly MFMAs' results must not be touched too early by code the compiler put behind them (hazard_lint.py): checked
    # on every object (re)compiled in this run - a finding is a wrong-answer bug in that build, so the library is not linked
    import hazard_lint

    if not os.path.exists(hazard_lint.OBJDUMP):      # (a toolchain without llvm-objdump: the check cannot run - say so, do not fail the build)
        print(f"build.py: {hazard_lint.OBJDUMP} not found - the MFMA hazard lint is skipped", file=sys.stderr)
        rebuilt = []
    for obj in rebuilt:
        found = hazard_lint.lint_object(obj)
        if found:
            func, mfma, ins, seen, need = found[0]
            os.remove(obj)
            raise RuntimeError(f"{os.path.basename(obj)}: {len(found)} MFMA result(s) read too early, e.g. in {func}: '{ins}' "
                               f"{seen} wait states behind '{mfma}' ({need} needed) - see csrc/hazard_lint.py")
    objs = [o for _, o, extra in _jobs() if not extra]
    prof = {s: o for s, o, extra in _jobs() if extra}
    prof_objs = [prof.get(s, o) for s, o, extra in _jobs() if not extra]
    for out, ol in ((OUT, objs), (OUT_PROF, prof_objs)):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + ol
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
