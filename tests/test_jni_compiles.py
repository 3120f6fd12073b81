"""The JNI shims (jni/*.c) go through a C compiler: `gcc -fsyntax-only -Wall -Wextra -Werror` against
tools/jni_stub/jni.h, a compile-check stub of the ~20 JNI entry points they use (the image has no JDK, so until
round 5 these files had only ever been checked with regular expressions).  Catches: wrong argument counts and types
against include/myrrix_als.h, misspelt JNI calls, int / long mix-ups in the array copies."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jni_shims_compile_cleanly():
    files = sorted(glob.glob(os.path.join(ROOT, "jni", "*.c")))
    assert len(files) == 3   # factorizer, Solver SPI, input files + top-N
    for f in files:
        r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tools", "jni_stub"),
                            "-I" + os.path.join(ROOT, "include"), f], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_the_stub_declares_only_what_the_shims_use():
    """Keeps the stub honest: every table entry is called somewhere in jni/*.c (SetFloatArrayRegion aside, kept for
    symmetry), so a shim that starts using a JNI function the stub lacks fails the compile test instead of silently
    growing the stub's surface."""
    import re
    stub = open(os.path.join(ROOT, "tools", "jni_stub", "jni.h")).read()
    declared = set(re.findall(r"\(JNICALL\* (\w+)\)", stub))
    used = set()
    for f in glob.glob(os.path.join(ROOT, "jni", "*.c")):
        used |= set(re.findall(r"\(\*(?:c->)?env\)->(\w+)", open(f).read()))
    assert used <= declared, used - declared
    assert declared - used <= {"SetFloatArrayRegion"}, declared - used


def test_serving_shim_and_its_java_class_declare_the_same_natives():
    """jni/myrrix_serving_jni.c <-> java/.../generation/NativeGeneration.java: the same native methods, the same number of
    arguments and the matching JNI types (no JDK here to tell us at link time)."""
    import re
    c = open(os.path.join(ROOT, "jni", "myrrix_serving_jni.c")).read()
    j = open(os.path.join(ROOT, "java", "net", "myrrix", "online", "generation", "NativeGeneration.java")).read()
    ctype = {"int": "jint", "long": "jlong", "float": "jfloat", "boolean": "jboolean", "long[]": "jlongArray", "int[]": "jintArray",
             "float[]": "jfloatArray", "String": "jstring", "void": "void"}
    natives = {}
    for m in re.finditer(r"private static native (\S+) (\w+)\(([^)]*)\);", j, re.S):
        args = [a.strip().rsplit(" ", 1)[0] for a in m.group(3).split(",") if a.strip()]
        natives[m.group(2)] = (ctype[m.group(1)], [ctype[a] for a in args])
    exported = {}
    for m in re.finditer(r"JNIEXPORT (\w+) JNICALL JNI_FN\((\w+)\)\(JNIEnv\* env, jclass cls([^)]*)\)", c, re.S):
        args = [a.strip().rsplit(" ", 1)[0] for a in m.group(3).split(",") if a.strip()]
        exported[m.group(2)] = (m.group(1), args)
    assert natives and natives == exported, (sorted(set(natives) ^ set(exported)), [k for k in natives if k in exported and natives[k] != exported[k]])
