"""The C++ host mirror of the reference interface (include/myrrix/factorizer.hpp) drives the C-ABI
directly: tests/cpp/test_als_known_answers.cpp re-states AlternatingLeastSquaresTest,
NegativeInputTest and MatrixUtilsTest on it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "test_als_known_answers")


def build():
    subprocess.check_call(["make", "-C", CPP, "test_als_known_answers"], stdout=subprocess.DEVNULL)


@pytest.mark.gpu
def test_reference_unit_tests_through_the_cpp_mirror():
    build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL PASSED" in r.stdout


def test_cpp_mirror_builds_and_has_no_cpu_fallback():
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stdout


def test_model_file_through_the_cpp_mirror(tmp_path):
    """include/myrrix/serializer.hpp (GenerationSerializer) on the C-ABI: host code only."""
    from tests.test_model_oracle import TINY_STREAM
    subprocess.check_call(["make", "-C", CPP, "test_model_file"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(CPP, "test_model_file"), str(tmp_path), TINY_STREAM.hex()], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_read_input_files_through_the_cpp_mirror(tmp_path):
    """myrrix::readInputFiles (include/myrrix/generation.hpp) = InputFilesReader.readInputFiles on the device: two files
    in last-modified order, a header, a removal, a pruned entry, both kinds of tag with the reference's own hash vectors,
    knownItemIDs, and the bad-line abort as std::ios_base::failure."""
    subprocess.check_call(["make", "-C", CPP, "test_read_input_files"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(CPP, "test_read_input_files"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
