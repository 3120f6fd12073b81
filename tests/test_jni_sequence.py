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
