"""The call sequence of the JNI shim (jni/myrrix_als_jni.c, committed unbuilt: no JDK in this image), replayed
in C++ against the library on the reference's known-answer cases: tests/cpp/test_jni_call_sequence.cpp."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "test_jni_call_sequence")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))


def build():
    subprocess.check_call(["make", "-C", CPP, "test_jni_call_sequence"], stdout=subprocess.DEVNULL)


def write_case(path, case):
    R, Y0, E = case["R"], case["Y0"], case["expected_XYT"]
    with open(path, "w") as f:
        f.write("%d %r %d %d %d %d %r\n" % (case["features"], case["threshold"], case["max_iterations"], case["flags"], len(R), len(R[0]),
                                           case["tol"]))
        for M in (R, Y0, E):
            for row in M:
                f.write(" ".join(repr(float(v)) for v in row) + "\n")


def test_shim_and_adapter_sources_are_complete():
    """Every native method the Java adapter declares has its JNI function, and the shim only calls entry
    points the header declares."""
    import re
    java = open(os.path.join(ROOT, "java/net/myrrix/online/factorizer/als/HipAlternatingLeastSquares.java")).read()
    shim = open(os.path.join(ROOT, "jni/myrrix_als_jni.c")).read()
    header = open(os.path.join(ROOT, "include/myrrix_als.h")).read()
    natives = set(re.findall(r"private static native \w+(?:\[\])? (native\w+)\(", java))
    assert len(natives) >= 12
    assert natives == set(re.findall(r"JNI_FN\((native\w+)\)", shim))
    for fn in set(re.findall(r"\b(mals_\w+)\(", shim)):
        assert re.search(r"\b%s\(" % fn, header), fn
    strip = lambda t: re.sub(r"/\*.*?\*/|//[^\n]*", "", t, flags=re.S)      # noqa: E731 -- comments may say "..."
    assert "..." not in strip(shim) and "..." not in strip(java)            # no elided bodies


def test_sequence_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = tmp_path / "case.txt"
    write_case(str(p), GOLDEN["als_default"])
    r = subprocess.run([BIN, str(p), "1", "0", "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["als_default", "als_reconstruct_r", "als_negative_input"])
@pytest.mark.parametrize("members,backend,piece", [(1, 0, 2), (1, 0, 1 << 20), (2, 1, 2), (3, 1, 1)])
def test_jni_call_sequence_reproduces_the_known_answers(tmp_path, name, members, backend, piece):
    build()
    p = tmp_path / "case.txt"
    write_case(str(p), GOLDEN[name])
    r = subprocess.run([BIN, str(p), str(members), str(backend), str(piece)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


# ---- the Solver SPI behind the reference's own hook (MatrixUtils.java:44-49 -> net.myrrix.common.math.JBlasLinearSystemSolver)
SPI_BIN = os.path.join(CPP, "test_solver_spi_sequence")


def test_solver_spi_sources_are_complete():
    """JBlasLinearSystemSolver (same FQCN the reference loads reflectively) + NativeSolver + their JNI functions:
    every native method has its function, the shim only calls declared entry points, and the class implements
    exactly the reference's package-private interface."""
    import re
    lss = open(os.path.join(ROOT, "java/net/myrrix/common/math/JBlasLinearSystemSolver.java")).read()
    ns = open(os.path.join(ROOT, "java/net/myrrix/common/math/NativeSolver.java")).read()
    shim = open(os.path.join(ROOT, "jni/myrrix_solver_jni.c")).read()
    header = open(os.path.join(ROOT, "include/myrrix_als.h")).read()
    assert "package net.myrrix.common.math;" in lss and "public final class JBlasLinearSystemSolver implements LinearSystemSolver" in lss
    assert "public Solver getSolver(RealMatrix M)" in lss and "public boolean isNonSingular(RealMatrix M)" in lss   # LinearSystemSolver.java:39,45
    assert "public float[] solveDToF(double[] b)" in ns and "public double[] solveFToD(float[] b)" in ns            # Solver.java:35,41
    natives = set(re.findall(r"static native \w+(?:\[\])? (native\w+)\(", ns))
    assert natives == {"nativeCreate", "nativeRecompute", "nativeSolveDToF", "nativeSolveFToD", "nativeDestroy"}
    assert natives == set(re.findall(r"JNI_FN\((native\w+)\)", shim))
    assert "Java_net_myrrix_common_math_NativeSolver_" in shim
    for fn in set(re.findall(r"\b(mals_\w+)\(", shim)):
        assert re.search(r"\b%s\(" % fn, header), fn
    strip = lambda t: re.sub(r"/\*.*?\*/|//[^\n]*", "", t, flags=re.S)      # noqa: E731
    assert all("..." not in strip(t) for t in (lss, ns, shim))


def test_solver_spi_call_sequence_on_the_host():
    """Generation.recomputeSolver's inputs (Generation.java:142-158) through mals_solver_*: healthy side, inf-norm < 1
    (IllConditionedSolverException before any solver is built), fewer rows than features (SingularMatrixSolverException
    with the apparent rank), isNonSingular.  No GPU needed."""
    subprocess.check_call(["make", "-C", CPP, "test_solver_spi_sequence"], stdout=subprocess.DEVNULL)
    r = subprocess.run([SPI_BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALL PASSED (host)" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_solver_spi_call_sequence_with_the_device_gramian():
    """... plus NativeSolver.recompute: M^T M of the factors resident in the factorizer's group on the device."""
    subprocess.check_call(["make", "-C", CPP, "test_solver_spi_sequence"], stdout=subprocess.DEVNULL)
    r = subprocess.run([SPI_BIN, "device"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL PASSED (host + device)" in r.stdout, r.stdout + r.stderr
