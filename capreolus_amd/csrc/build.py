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
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcapreolus_amd.so")
OUT_PROF = os.path.join(HERE, "libcapreolus_amd_prof.so")
OUT_PYHOST = os.path.join(HERE, "libcapamd_pyhost" + (__import__("sysconfig").get_config_var("EXT_SUFFIX") or ".so"))   # (the interpreter's ABI tag: a helper built for another Python is never loaded)
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


def _hazard_lint():
    """csrc/hazard_lint.py as a module, whichever way this file was loaded (package import or `python build.py`) - without putting csrc/ on
    sys.path (`import build` then resolved to this file for the whole process)"""
    import importlib.util

    name = "capreolus_amd_csrc_hazard_lint"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "hazard_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _lint_stamp(obj):
    return obj + ".lint"


def build(force=False, verbose=False):
    try:
        build_pyhost(force, verbose)      # optional: capreolus_amd/pyhost.py falls back to the Python expression without it
    except (OSError, subprocess.CalledProcessError) as e:
        print(f"build.py: the CPython helper was not built ({type(e).__name__}: {e}); predict() uses its Python fallback", file=sys.stderr)
    lint = _hazard_lint()
    objs_all = [o for _, o, _ in _jobs()]
    unlinted = [o for o in objs_all if os.path.exists(o) and not (os.path.exists(_lint_stamp(o)) and os.path.getmtime(_lint_stamp(o)) >= os.path.getmtime(o))]
    if not force and not stale() and not unlinted:
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
    # on EVERY object that goes into the libraries and has not passed since it was last compiled (a stamp file next to it) - a finding is
    # a wrong-answer bug in that build, so the library is not linked; a toolchain without llvm-objdump cannot be checked and does not
    # link either (CAPAMD_SKIP_HAZARD_LINT=1 says "I know")
    if os.environ.get("CAPAMD_SKIP_HAZARD_LINT") == "1":
        print("build.py: CAPAMD_SKIP_HAZARD_LINT=1 - the MFMA hazard lint is skipped", file=sys.stderr)
    else:
        if lint.objdump() is None:
            raise RuntimeError("llvm-objdump not found: the inline-assembly MFMA kernels cannot be checked (csrc/hazard_lint.py); "
                               "set ROCM_PATH, or CAPAMD_SKIP_HAZARD_LINT=1 to link unchecked objects")
        for obj in objs_all:
            if os.path.exists(_lint_stamp(obj)) and os.path.getmtime(_lint_stamp(obj)) >= os.path.getmtime(obj):
                continue
            found = lint.lint_object(obj)
            if found:
                func, mfma, ins, seen, need = found[0]
                os.remove(obj)
                raise RuntimeError(f"{os.path.basename(obj)}: {len(found)} MFMA result(s) touched too early, e.g. in {func}: '{ins}' "
                                   f"{seen} wait states behind '{mfma}' ({need} needed) - see csrc/hazard_lint.py")
            open(_lint_stamp(obj), "w").close()
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
