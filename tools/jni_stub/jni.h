/* tools/jni_stub/jni.h -- COMPILE-CHECK STUB, not a JNI implementation.
 *
 * The build image has no JDK, so jni/myrrix_als_jni.c and jni/myrrix_solver_jni.c could never go through a compiler
 * before they reach a maintainer's machine.  This header declares exactly the part of the Java Native Interface those
 * two files use -- the primitive types, the array / string / method handles as opaque pointers, and a JNINativeInterface_
 * table holding only the entries they call, with the signatures of the JNI specification (Java SE "JNI Functions",
 * chapter 4) -- so that tests/test_jni_compiles.py can run them through `gcc -fsyntax-only -Wall -Werror`.  The table
 * is NOT laid out like the real one (a real jni.h has ~230 slots in a fixed order): nothing compiled against this file
 * may ever be linked or loaded into a JVM.  It is test infrastructure and ships to nobody. */
#ifndef MALS_JNI_STUB_H
#define MALS_JNI_STUB_H
#include <stdarg.h>
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_COMMIT 1
#define JNI_ABORT 2

typedef unsigned char jboolean;
typedef signed char jbyte;
typedef unsigned short jchar;
typedef short jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jfloatArray;
typedef jarray jdoubleArray;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (JNICALL* GetObjectClass)(JNIEnv* env, jobject obj);
  jmethodID (JNICALL* GetMethodID)(JNIEnv* env, jclass clazz, const char* name, const char* sig);
  void (JNICALL* CallVoidMethod)(JNIEnv* env, jobject obj, jmethodID methodID, ...);
  jboolean (JNICALL* ExceptionCheck)(JNIEnv* env);
  void (JNICALL* ExceptionClear)(JNIEnv* env);
  jstring (JNICALL* NewStringUTF)(JNIEnv* env, const char* utf);
  jsize (JNICALL* GetArrayLength)(JNIEnv* env, jarray array);
  jint* (JNICALL* GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  jlong* (JNICALL* GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
  jfloat* (JNICALL* GetFloatArrayElements)(JNIEnv* env, jfloatArray array, jboolean* isCopy);
  jdouble* (JNICALL* GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* isCopy);
  void (JNICALL* ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (JNICALL* ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
  void (JNICALL* ReleaseFloatArrayElements)(JNIEnv* env, jfloatArray array, jfloat* elems, jint mode);
  void (JNICALL* ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
  void (JNICALL* SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  void (JNICALL* SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void (JNICALL* SetFloatArrayRegion)(JNIEnv* env, jfloatArray array, jsize start, jsize len, const jfloat* buf);
  void (JNICALL* SetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, const jdouble* buf);
};

#endif
