"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the host-side native code (GPU ASan is
not available on the pool, so: CPU builds only).  Each test compiles a small driver with
-fsanitize=address,undefined into tests/support/_san/ (cached by source mtime) and runs it; a sanitizer report
makes the driver exit non-zero (-fno-sanitize-recover)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUP = os.path.join(ROOT, "tests", "support")
OUT = os.path.join(SUP, "_san")
CSRC = os.path.join(ROOT, "meshlessmultigridpoisson_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17", "-pthread"]


def _build(exe, srcs, extra=()):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, exe)
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.run(["g++"] + SAN + ["-o", path] + list(srcs) + list(extra), check=True)
    return path


def _run(path, env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    e.update(env or {})
    r = subprocess.run([path], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_plan_builder_and_interpreter_under_asan_ubsan():
    srcs = [os.path.join(SUP, "sanitize_plan_main.cpp"), os.path.join(SUP, "plan_emulate.cpp"),
            os.path.join(CSRC, "device", "plan.cpp"), os.path.join(CSRC, "device", "level_plan.cpp")]
    out = _run(_build("sanitize_plan", srcs))
    assert "0 failure(s)" in out


def test_host_classes_setup_under_asan_ubsan():
    """The host mirror of Grid / Multigrid / FractionalStepGrid (csrc/host: kNN, ordering, RBF-FD weights on the
    host path, CSR assembly, transfers, .msh round trip) built with the sanitizers and linked against the real
    libmmgp.so (which is only asked for its device count here: no GPU in this container)."""
    import glob
    host_srcs = sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))
    srcs = [os.path.join(SUP, "sanitize_host_main.cpp")] + host_srcs
    pkg = os.path.join(ROOT, "meshlessmultigridpoisson_amd")
    exe = _build("sanitize_host", srcs, ["-L" + pkg, "-lmmgp", "-Wl,-rpath," + pkg])
    out = _run(exe, {"MMG_NUM_THREADS": "4", "LD_LIBRARY_PATH": pkg + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")})
    assert "sanitize_host_main: ok" in out
