/*
 * myrrix_serving_jni.c -- JNI shim for the rows of SURVEY.md section 8(f) either side of the factorization:
 *   * input files -> matrices (InputFilesReader.readInputFiles, online-local/src/net/myrrix/online/generation/
 *     InputFilesReader.java:64-211)          -> mals_ingest_* (read_dir / finish / counts / ids / CSR / tags / known items)
 *   * top-N scoring (ServerRecommender.recommend / recommendToMany / recommendToAnonymous,
 *     online/src/net/myrrix/online/ServerRecommender.java:366-508,561-606) -> mals_recommend, _to_many, _set_known_items
 * bound by net.myrrix.online.generation.NativeGeneration (java/net/myrrix/online/generation/NativeGeneration.java).  The
 * reference has no FFI on these paths either; this is the binding a maintainer adds.
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK); compiled with -fsyntax-only -Wall -Wextra -Werror against
 * tools/jni_stub/jni.h by tests/test_jni_compiles.py.  Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       -o libmyrrix_serving_jni.so jni/myrrix_serving_jni.c -Lmyrrix-recommender_amd/csrc -lmyrrix_als
 *
 * Arrays are pinned or copied for the duration of one call only (no GetPrimitiveArrayCritical: serving threads must not
 * stall one another); results go back with Set<Type>ArrayRegion.  Status codes only, no exception crosses.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "myrrix_als.h"

#define JNI_FN(name) Java_net_myrrix_online_generation_NativeGeneration_##name

static mals_handle as_handle(jlong h) { return (mals_handle)(intptr_t)h; }
static mals_ingest as_ingest(jlong g) { return (mals_ingest)(intptr_t)g; }

/* ---- input files (InputFilesReader.java:64-211) ------------------------------------------------------------------ */

/* new ingest object on `device`; zeroThreshold = model.decay.zeroThreshold (IFR:194-211); wantKnownItems = !model.noKnownItems.
 * 0 on failure. */
JNIEXPORT jlong JNICALL JNI_FN(nativeIngestCreate)(JNIEnv* env, jclass cls, jint device, jfloat zero_threshold, jboolean want_known_items) {
  (void)env; (void)cls;
  mals_ingest g = NULL;
  if (mals_ingest_create((int32_t)device, (float)zero_threshold, &g) != MALS_OK) return 0;
  if (mals_ingest_set_option(g, MALS_INGEST_OPT_KNOWN_ITEMS, want_known_items ? 1 : 0) != MALS_OK) {
    (void)mals_ingest_destroy(g);
    return 0;
  }
  return (jlong)(intptr_t)g;
}

JNIEXPORT void JNICALL JNI_FN(nativeIngestDestroy)(JNIEnv* env, jclass cls, jlong g) {
  (void)env; (void)cls;
  if (g) (void)mals_ingest_destroy(as_ingest(g));
}

JNIEXPORT jstring JNICALL JNI_FN(nativeIngestLastError)(JNIEnv* env, jclass cls, jlong g) {
  (void)cls;
  const char* e = g ? mals_ingest_last_error(as_ingest(g)) : "no ingest object";
  return (*env)->NewStringUTF(env, e ? e : "");
}

/* readInputFiles(inputDir): every *.csv / *.csv.gz / *.csv.zip of the directory in last-modified order, then the finish.
 * `dir` arrives as modified-UTF-8 bytes + terminating 0 in a byte-sized int array-free form: the Java side passes
 * File.getAbsolutePath().getBytes(UTF_8) widened to longs (paths are short; this keeps the stub's surface at the array
 * accessors the other shims already use).  info (long[8]): lines, badLines, headerLines, skippedLines, records, users, items, nnz. */
JNIEXPORT jint JNICALL JNI_FN(nativeReadInputDir)(JNIEnv* env, jclass cls, jlong g, jlongArray dir_bytes, jlongArray info) {
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, dir_bytes);
  if (n <= 0 || n > 4096 || (*env)->GetArrayLength(env, info) < 8) return MALS_INVALID_ARG;
  char* path = (char*)malloc((size_t)n + 1);
  if (!path) return MALS_OOM;
  jlong* b = (*env)->GetLongArrayElements(env, dir_bytes, NULL);
  if (!b) {
    free(path);
    return MALS_OOM;
  }
  for (jsize i = 0; i < n; ++i) path[i] = (char)b[i];
  path[n] = 0;
  (*env)->ReleaseLongArrayElements(env, dir_bytes, b, JNI_ABORT);
  int32_t n_files = 0;
  int rc = mals_ingest_read_dir(as_ingest(g), path, &n_files);   /* "Too many bad lines" (IFR:96-98) comes back as a status + text */
  free(path);
  if (rc == MALS_OK) rc = mals_ingest_finish(as_ingest(g));
  if (rc != MALS_OK) return rc;
  mals_ingest_text_info_t ti;
  ti.struct_size = (int32_t)sizeof ti;
  rc = mals_ingest_text_info(as_ingest(g), &ti);
  if (rc != MALS_OK) return rc;
  int64_t n_records = 0, n_users = 0, n_items = 0, nnz = 0;
  rc = mals_ingest_counts(as_ingest(g), &n_records, &n_users, &n_items, &nnz);
  if (rc != MALS_OK) return rc;
  const jlong out[8] = {(jlong)ti.lines, (jlong)ti.bad_lines, (jlong)ti.header_lines, (jlong)ti.skipped_lines,
                        (jlong)n_records, (jlong)n_users, (jlong)n_items, (jlong)nnz};
  (*env)->SetLongArrayRegion(env, info, 0, 8, out);
  return MALS_OK;
}

/* dense index -> id of one side (side 0 = users / X, 1 = items / Y): ids must hold the side's count */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestIds)(JNIEnv* env, jclass cls, jlong g, jint side, jlongArray ids) {
  (void)cls;
  /* the native side writes the side's count of ids: the array must hold them (a short one would corrupt the Java heap) */
  int64_t n_users = 0, n_items = 0;
  const int rc0 = mals_ingest_counts(as_ingest(g), NULL, &n_users, &n_items, NULL);
  if (rc0 != MALS_OK) return rc0;
  if (!ids || (side != 0 && side != 1) || (int64_t)(*env)->GetArrayLength(env, ids) < (side == 0 ? n_users : n_items)) return MALS_INVALID_ARG;
  jlong* p = (*env)->GetLongArrayElements(env, ids, NULL);
  if (!p) return MALS_OOM;
  const int rc = mals_ingest_get_ids(as_ingest(g), (int)side, (int64_t*)p);
  (*env)->ReleaseLongArrayElements(env, ids, p, rc == MALS_OK ? 0 : JNI_ABORT);
  return rc;
}

/* itemTagIDs (which = 0) / userTagIDs (1), IFR:152-165: pass an array of n_item_tag_ids / n_user_tag_ids longs */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestTagIds)(JNIEnv* env, jclass cls, jlong g, jint which, jlongArray ids) {
  (void)cls;
  mals_ingest_text_info_t ti;
  ti.struct_size = (int32_t)sizeof ti;
  const int rc0 = mals_ingest_text_info(as_ingest(g), &ti);
  if (rc0 != MALS_OK) return rc0;
  const int64_t need = which == 0 ? ti.n_item_tag_ids : ti.n_user_tag_ids;
  if (!ids || (which != 0 && which != 1) || need < 0 || (int64_t)(*env)->GetArrayLength(env, ids) < need) return MALS_INVALID_ARG;
  jlong* p = (*env)->GetLongArrayElements(env, ids, NULL);
  if (!p) return MALS_OOM;
  const int rc = mals_ingest_get_tag_ids(as_ingest(g), (int32_t)which, (int64_t*)p);
  (*env)->ReleaseLongArrayElements(env, ids, p, rc == MALS_OK ? 0 : JNI_ABORT);
  return rc;
}

/* tag id counts and the size of knownItemIDs after the finish: out (long[3]) = itemTagIDs, userTagIDs, known entries (-1: none) */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestSetSizes)(JNIEnv* env, jclass cls, jlong g, jlongArray out) {
  (void)cls;
  if ((*env)->GetArrayLength(env, out) < 3) return MALS_INVALID_ARG;
  mals_ingest_text_info_t ti;
  ti.struct_size = (int32_t)sizeof ti;
  const int rc = mals_ingest_text_info(as_ingest(g), &ti);
  if (rc != MALS_OK) return rc;
  const jlong v[3] = {(jlong)ti.n_item_tag_ids, (jlong)ti.n_user_tag_ids, (jlong)ti.n_known_items};
  (*env)->SetLongArrayRegion(env, out, 0, 3, v);
  return MALS_OK;
}

/* R by row (side 0) or by column (side 1) as CSR over the dense indices, for the maps the Java side still wants
 * (RbyRow / RbyColumn of the Generation): rowPtr has count + 1 longs, colIdx / val have nnz entries */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestCsr)(JNIEnv* env, jclass cls, jlong g, jint side, jlongArray row_ptr, jintArray col_idx,
                                              jfloatArray val) {
  (void)cls;
  int64_t n_users = 0, n_items = 0, nnz = 0;
  const int rc0 = mals_ingest_counts(as_ingest(g), NULL, &n_users, &n_items, &nnz);
  if (rc0 != MALS_OK) return rc0;
  if (!row_ptr || !col_idx || !val || (side != 0 && side != 1) ||
      (int64_t)(*env)->GetArrayLength(env, row_ptr) < (side == 0 ? n_users : n_items) + 1 ||
      (int64_t)(*env)->GetArrayLength(env, col_idx) < nnz || (int64_t)(*env)->GetArrayLength(env, val) < nnz)
    return MALS_INVALID_ARG;
  jlong* rp = (*env)->GetLongArrayElements(env, row_ptr, NULL);
  jint* ci = (*env)->GetIntArrayElements(env, col_idx, NULL);
  jfloat* v = (*env)->GetFloatArrayElements(env, val, NULL);
  int rc = MALS_OOM;
  if (rp && ci && v) rc = mals_ingest_get_csr(as_ingest(g), (int)side, (int64_t*)rp, (int32_t*)ci, (float*)v);
  const jint mode = rc == MALS_OK ? 0 : JNI_ABORT;
  if (v) (*env)->ReleaseFloatArrayElements(env, val, v, mode);
  if (ci) (*env)->ReleaseIntArrayElements(env, col_idx, ci, mode);
  if (rp) (*env)->ReleaseLongArrayElements(env, row_ptr, rp, mode);
  return rc;
}

/* knownItemIDs as CSR over the dense user indices (IFR:172-191): ptr has users + 1 longs, itemIdx the known entries */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestKnownItems)(JNIEnv* env, jclass cls, jlong g, jlongArray ptr, jintArray item_idx) {
  (void)cls;
  int64_t n_users = 0;
  mals_ingest_text_info_t ti;
  ti.struct_size = (int32_t)sizeof ti;
  int rc0 = mals_ingest_counts(as_ingest(g), NULL, &n_users, NULL, NULL);
  if (rc0 == MALS_OK) rc0 = mals_ingest_text_info(as_ingest(g), &ti);
  if (rc0 != MALS_OK) return rc0;
  if (!ptr || !item_idx || ti.n_known_items < 0 || (int64_t)(*env)->GetArrayLength(env, ptr) < n_users + 1 ||
      (int64_t)(*env)->GetArrayLength(env, item_idx) < ti.n_known_items)
    return MALS_INVALID_ARG;
  jlong* p = (*env)->GetLongArrayElements(env, ptr, NULL);
  jint* ii = (*env)->GetIntArrayElements(env, item_idx, NULL);
  int rc = MALS_OOM;
  if (p && ii) rc = mals_ingest_get_known_items(as_ingest(g), (int64_t*)p, (int32_t*)ii);
  const jint mode = rc == MALS_OK ? 0 : JNI_ABORT;
  if (ii) (*env)->ReleaseIntArrayElements(env, item_idx, ii, mode);
  if (p) (*env)->ReleaseLongArrayElements(env, ptr, p, mode);
  return rc;
}

/* both matrices (and knownItemIDs, if built) handed to a factorizer handle on the same device without leaving HBM:
 * the handle borrows the arrays, keep the ingest object alive while it uses them */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestInstall)(JNIEnv* env, jclass cls, jlong g, jlong handle) {
  (void)env; (void)cls;
  return mals_ingest_install(as_ingest(g), as_handle(handle));
}

/* ... or to every member of a factorizer group, cut at its bounds, device to device (mals_ingest_install_group; flags:
 * MALS_INSTALL_COPY = the members own copies and the ingest may be closed) */
JNIEXPORT jint JNICALL JNI_FN(nativeIngestInstallGroup)(JNIEnv* env, jclass cls, jlong g, jlong group, jint flags) {
  (void)env; (void)cls;
  return mals_ingest_install_group(as_ingest(g), (mals_group)(intptr_t)group, (int32_t)flags);
}

/* ---- top-N (ServerRecommender.java:366-508, RecommendIterator.java:62-109, TopN.java:49-128) ------------------------- */

/* the single-GPU handle of member `member` of a factorizer group (HipAlternatingLeastSquares keeps the group): its factors and
 * R stay resident after call(), so the generation that was just built can be served from the same memory.  0 on failure. */
JNIEXPORT jlong JNICALL JNI_FN(nativeGroupHandle)(JNIEnv* env, jclass cls, jlong group, jint member) {
  (void)env; (void)cls;
  mals_handle h = NULL;
  int32_t rank = 0;
  if (!group || mals_group_local((mals_group)(intptr_t)group, (int32_t)member, &h, &rank) != MALS_OK) return 0;
  return (jlong)(intptr_t)h;
}

/* generation.getKnownItemIDs() for the handle's users (SR:394-425): CSR over its local user rows of dense item indices;
 * rowPtr null = back to the rows of R */
JNIEXPORT jint JNICALL JNI_FN(nativeSetKnownItems)(JNIEnv* env, jclass cls, jlong handle, jlongArray row_ptr, jintArray item_idx) {
  (void)cls;
  if (!row_ptr) return mals_set_known_items(as_handle(handle), 0, NULL, NULL, MALS_MEM_HOST);
  const jsize n = (*env)->GetArrayLength(env, row_ptr);
  if (n < 1 || !item_idx) return MALS_INVALID_ARG;
  jlong* rp = (*env)->GetLongArrayElements(env, row_ptr, NULL);
  jint* ii = (*env)->GetIntArrayElements(env, item_idx, NULL);
  int rc = MALS_OOM;
  /* the offsets index item_idx: they must stay inside it */
  if (rp && ii && (rp[0] != 0 || rp[n - 1] < 0 || rp[n - 1] > (jlong)(*env)->GetArrayLength(env, item_idx))) rc = MALS_INVALID_ARG;
  else if (rp && ii) rc = mals_set_known_items(as_handle(handle), (int64_t)n - 1, (const int64_t*)rp, (const int32_t*)ii, MALS_MEM_HOST);
  if (ii) (*env)->ReleaseIntArrayElements(env, item_idx, ii, JNI_ABORT);
  if (rp) (*env)->ReleaseLongArrayElements(env, row_ptr, rp, JNI_ABORT);
  return rc;
}

/* recommend(userID, howMany, considerKnownItems, null) for a batch of model users by dense index (SR:382-441): items /
 * scores hold users x howMany results, best first, -1 / -inf padded; counts (may be null) the results per user */
JNIEXPORT jint JNICALL JNI_FN(nativeRecommend)(JNIEnv* env, jclass cls, jlong handle, jlongArray user_idx, jint how_many,
                                              jboolean consider_known_items, jlongArray items, jfloatArray scores, jintArray counts) {
  (void)cls;
  const jsize nq = (*env)->GetArrayLength(env, user_idx);
  if (how_many < 1 || (int64_t)(*env)->GetArrayLength(env, items) < (int64_t)nq * how_many ||
      (int64_t)(*env)->GetArrayLength(env, scores) < (int64_t)nq * how_many || (counts && (*env)->GetArrayLength(env, counts) < nq))
    return MALS_INVALID_ARG;
  jlong* u = (*env)->GetLongArrayElements(env, user_idx, NULL);
  jlong* it = (*env)->GetLongArrayElements(env, items, NULL);
  jfloat* sc = (*env)->GetFloatArrayElements(env, scores, NULL);
  jint* cn = counts ? (*env)->GetIntArrayElements(env, counts, NULL) : NULL;
  int rc = MALS_OOM;
  if (u && it && sc && (cn || !counts))
    rc = mals_recommend(as_handle(handle), (const int64_t*)u, (int32_t)nq, (int32_t)how_many, consider_known_items ? 1 : 0, (int64_t*)it, (float*)sc,
                        (int32_t*)cn);
  const jint mode = rc == MALS_OK ? 0 : JNI_ABORT;
  if (cn) (*env)->ReleaseIntArrayElements(env, counts, cn, mode);
  if (sc) (*env)->ReleaseFloatArrayElements(env, scores, sc, mode);
  if (it) (*env)->ReleaseLongArrayElements(env, items, it, mode);
  if (u) (*env)->ReleaseLongArrayElements(env, user_idx, u, JNI_ABORT);
  return rc;
}

/* recommendToMany (SR:366-441) and recommendToAnonymous (SR:561-606): query q owns the rows vectorPtr[q] .. vectorPtr[q+1] of
 * `vectors` (features floats each; vectorPtr null = one each), its exclusion list is excludeIdx[excludePtr[q] ..
 * excludePtr[q+1]) (dense item indices; both null = none).  Score = the reference's mean of dots (RecommendIterator.java:93-104). */
JNIEXPORT jint JNICALL JNI_FN(nativeRecommendToMany)(JNIEnv* env, jclass cls, jlong handle, jfloatArray vectors, jlongArray vector_ptr, jint n_queries,
                                                    jint how_many, jlongArray exclude_ptr, jlongArray exclude_idx, jlongArray items,
                                                    jfloatArray scores, jintArray counts) {
  (void)cls;
  if (n_queries < 0 || how_many < 1 || (vector_ptr && (*env)->GetArrayLength(env, vector_ptr) < n_queries + 1) ||
      (!exclude_ptr) != (!exclude_idx) || (exclude_ptr && (*env)->GetArrayLength(env, exclude_ptr) < n_queries + 1) ||
      (int64_t)(*env)->GetArrayLength(env, items) < (int64_t)n_queries * how_many ||
      (int64_t)(*env)->GetArrayLength(env, scores) < (int64_t)n_queries * how_many || (counts && (*env)->GetArrayLength(env, counts) < n_queries))
    return MALS_INVALID_ARG;
  jfloat* v = (*env)->GetFloatArrayElements(env, vectors, NULL);
  jlong* vp = vector_ptr ? (*env)->GetLongArrayElements(env, vector_ptr, NULL) : NULL;
  jlong* ep = exclude_ptr ? (*env)->GetLongArrayElements(env, exclude_ptr, NULL) : NULL;
  jlong* ei = exclude_idx ? (*env)->GetLongArrayElements(env, exclude_idx, NULL) : NULL;
  jlong* it = (*env)->GetLongArrayElements(env, items, NULL);
  jfloat* sc = (*env)->GetFloatArrayElements(env, scores, NULL);
  jint* cn = counts ? (*env)->GetIntArrayElements(env, counts, NULL) : NULL;
  int rc = MALS_OOM;
  if (v && it && sc && (vp || !vector_ptr) && (ep || !exclude_ptr) && (ei || !exclude_idx) && (cn || !counts)) {
    /* what the offsets promise must be inside the arrays they index */
    mals_handle hh = as_handle(handle);
    const int64_t n_vec = vp ? (int64_t)vp[n_queries] : (int64_t)n_queries;
    const int64_t features = (int64_t)mals_features(hh);
    if (n_vec < 0 || features <= 0 || (int64_t)(*env)->GetArrayLength(env, vectors) < n_vec * features ||
        (ep && (ep[n_queries] < 0 || (int64_t)(*env)->GetArrayLength(env, exclude_idx) < (int64_t)ep[n_queries])))
      rc = MALS_INVALID_ARG;
    else
    rc = mals_recommend_to_many(as_handle(handle), (const float*)v, (const int64_t*)vp, (int32_t)n_queries, (int32_t)how_many, (const int64_t*)ep,
                                (const int64_t*)ei, (int64_t*)it, (float*)sc, (int32_t*)cn);
  }
  const jint mode = rc == MALS_OK ? 0 : JNI_ABORT;
  if (cn) (*env)->ReleaseIntArrayElements(env, counts, cn, mode);
  if (sc) (*env)->ReleaseFloatArrayElements(env, scores, sc, mode);
  if (it) (*env)->ReleaseLongArrayElements(env, items, it, mode);
  if (ei) (*env)->ReleaseLongArrayElements(env, exclude_idx, ei, JNI_ABORT);
  if (ep) (*env)->ReleaseLongArrayElements(env, exclude_ptr, ep, JNI_ABORT);
  if (vp) (*env)->ReleaseLongArrayElements(env, vector_ptr, vp, JNI_ABORT);
  if (v) (*env)->ReleaseFloatArrayElements(env, vectors, v, JNI_ABORT);
  return rc;
}

JNIEXPORT jstring JNICALL JNI_FN(nativeLastError)(JNIEnv* env, jclass cls, jlong handle) {
  (void)cls;
  const char* e = handle ? mals_last_error(as_handle(handle)) : "no handle";
  return (*env)->NewStringUTF(env, e ? e : "");
}
