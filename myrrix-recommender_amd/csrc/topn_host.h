// topn_host.h -- host side of top-N scoring (mals_recommend*, include/myrrix_als.h; kernels and the argument for
// exactness in topn_kernels.h).  Included by mals_api.hip inside its anonymous namespace, after mals_handle_s.
#pragma once

constexpr int TOPN_SLOTS = 6;  // passes in flight, each on its own stream (3 -> 6: 3-4 % at 64-128 queries per pass, nothing at 240)

// Everything one pass of the filter path owns.  Passes are independent (Y, X and the known items are only read), so
// pass p runs on stream p % TOPN_SLOTS: while the streaming filter kernel of one pass has the chip, the small kernels
// either side of it (prepare / sample / threshold of the next pass, scatter / rescore / final of the previous one) run
// beside it instead of in its shadow -- serialised on one stream they and the launch gaps between them were two thirds
// of a pass.
struct TopnSlot {
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  // inputs of the pass: ONE device block, the image of the slot's pinned block (one copy); the pointers below are views
  // into it
  uint8_t* d_in = nullptr;
  size_t din_cap = 0;
  uint8_t* h_in = nullptr;            // pinned input block (offsets, rows, vectors, exclusion lists)
  size_t in_cap = 0;
  const float* d_vecs = nullptr;      // [n_vecs][k] (caller's vectors) or X itself (model users: rows d_vrow)
  const int64_t* d_vrow = nullptr;    // vector v = row d_vrow[v] of d_vecs; NULL: row v
  const int32_t* d_vptr = nullptr;    // [nq + 1]
  const int64_t* d_rows = nullptr;    // [nq]: local row of the query's user (known items), -1 = none
  const int64_t* d_excl_ptr = nullptr;
  const int64_t* d_excl_idx = nullptr;
  float *d_tau = nullptr, *d_bmax = nullptr;  // thresholds; the sample's bucket maxima [queries][16 x TOPN_SAMPLE_GROUPS] ...
  uint32_t* d_bidx = nullptr;                 // ... and the items that attain them
  unsigned* d_count = nullptr;
  uint32_t* d_cand = nullptr;
  uint64_t* d_pairs = nullptr;
  void* d_img = nullptr;        // the pass's queries as split bf16 MFMA operands (topn_prepare_kernel)
  unsigned* d_wcount = nullptr; // hits per wave of the filter kernel, [n_waves] + one overflow word
  uint2* d_whits = nullptr;     // [n_waves][TOPN_WAVE_CAP] (item, query)
  size_t wh_cap = 0;
  size_t cand_cap = 0, pairs_cap = 0;
  uint8_t* h_stage = nullptr;   // pinned: the pass's results, [nq][how_many] pairs | counts | taus | overflow word (topn_final_kernel
                                // writes them there), decoded while later passes run
  size_t stage_cap = 0;
};

struct TopnWorkspace {
  TopnSlot slot[TOPN_SLOTS];
  hipEvent_t ev_begin = nullptr;  // the caller's stream at the start of the call: every slot stream waits for it
  // dense path
  float* d_scores = nullptr;
  uint32_t* d_sel = nullptr;
  TopnState* d_state = nullptr;
  unsigned* d_hist = nullptr;
  size_t scores_cap = 0, sel_cap = 0;
};

void topn_free(mals_handle h) {
  TopnWorkspace* w = static_cast<TopnWorkspace*>(h->tn_ws);
  if (!w) return;
  for (TopnSlot& s : w->slot) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    free_dev(s.d_in);
    free_dev(s.d_tau); free_dev(s.d_bmax); free_dev(s.d_bidx); free_dev(s.d_count); free_dev(s.d_cand);
    free_dev(s.d_pairs); free_dev(s.d_img); free_dev(s.d_wcount); free_dev(s.d_whits);
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (w->ev_begin) (void)hipEventDestroy(w->ev_begin);
  free_dev(w->d_scores); free_dev(w->d_sel); free_dev(w->d_state); free_dev(w->d_hist);
  delete w;
  h->tn_ws = nullptr;
}

// grow a buffer only `stream` uses
template <typename P>
int topn_grow(mals_handle h, hipStream_t stream, P*& p, size_t& cap, size_t want) {
  if (want <= cap) return MALS_OK;
  HIPCHK(h, hipStreamSynchronize(stream));  // a pass still in flight may use the old buffer
  free_dev(p);
  cap = 0;
  HIPCHK(h, hipMalloc(&p, sizeof(P) * want));
  cap = want;
  return MALS_OK;
}

struct TopnOut {
  int64_t* items;
  float* scores;
  int32_t* n;
};
// what one call asks for (host pointers)
struct TopnRequest {
  int n_queries = 0, how_many = 0;
  const int64_t* user_idx = nullptr;     // model users: query q's single vector is X[user_idx[q]] ...
  bool skip_known = false;               // ... and its user's known items are skipped
  const float* vectors = nullptr;        // or caller's vectors: [n_vectors][k],
  const int64_t* vec_ptr = nullptr;      //   query q owns vectors [vec_ptr[q], vec_ptr[q+1]) (NULL: one each)
  const int64_t* excl_ptr = nullptr;     // optional per-query exclusion lists (item indices)
  const int64_t* excl_idx = nullptr;
  int64_t* item_idx_out = nullptr;
  float* score_out = nullptr;
  int32_t* n_out = nullptr;
  // a coalesced pass of the serving front (below): queries of several callers -- every query has its own output block and
  // its own "skip the known items" flag
  const TopnOut* out_q = nullptr;
  const uint8_t* skip_known_q = nullptr;
};

inline TopnOut topn_out(const TopnRequest& rq, size_t qq) {
  if (rq.out_q) return rq.out_q[qq];
  return {rq.item_idx_out + qq * (size_t)rq.how_many, rq.score_out + qq * (size_t)rq.how_many, rq.n_out ? rq.n_out + qq : nullptr};
}

struct TopnPass {
  int q0 = 0, nq = 0;
  int64_t v0 = 0, n_vecs = 0;
  bool have_rows = false, have_excl = false;
};

struct TopnCand {
  uint32_t key;
  int64_t idx;
};
// false: a non-finite score among the results (the reference: Preconditions.checkState(isFinite(result)), RecommendIterator.java:105)
bool topn_emit(std::vector<TopnCand>& cand, int how_many, int64_t* item_idx_out, float* score_out, int32_t* n_out) {
  // best score first; equal scores in ascending item index (the reference's order among ties is its hash order)
  std::sort(cand.begin(), cand.end(), [](const TopnCand& a, const TopnCand& b) { return a.key != b.key ? a.key > b.key : a.idx < b.idx; });
  const int n = (int)std::min<size_t>(cand.size(), (size_t)how_many);
  if (n_out) *n_out = n;
  for (int j = 0; j < how_many; ++j) {
    if (j < n) {
      item_idx_out[j] = cand[(size_t)j].idx;
      score_out[j] = key_score(cand[(size_t)j].key);
    } else {
      item_idx_out[j] = -1;
      score_out[j] = -std::numeric_limits<float>::infinity();
    }
  }
  // (NaN sorts above +inf in score_key: a non-finite score, if there is one, is the first result)
  return n == 0 || std::isfinite(score_out[0]);
}

// The pass's vectors, offsets, known-item rows and exclusion lists on the device: assembled in the slot's pinned input
// block and sent with ONE asynchronous copy on `stream` (the host never waits for a pass that is still running; the slot's
// previous pass has been decoded before the slot is used again).
int topn_upload_pass(mals_handle h, TopnSlot& sl, hipStream_t stream, const TopnRequest& rq, TopnPass& ps) {
  const int k = h->cfg.features;
  SideState& x = h->side[MALS_SIDE_X];
  const bool own_vectors = !rq.user_idx;
  if (rq.user_idx || !rq.vec_ptr) {
    ps.v0 = ps.q0;
    ps.n_vecs = ps.nq;
  } else {
    ps.v0 = rq.vec_ptr[ps.q0];
    ps.n_vecs = rq.vec_ptr[ps.q0 + ps.nq] - ps.v0;
  }
  const int64_t n_ex = (rq.excl_ptr && rq.excl_idx) ? rq.excl_ptr[ps.q0 + ps.nq] - rq.excl_ptr[ps.q0] : 0;
  // layout of the block (8-byte aligned pieces)
  const size_t o_vptr = 0, o_rows = o_vptr + 8 * ((TOPN_FILTER_QUERIES + 2) / 2), o_uidx = o_rows + 8 * TOPN_FILTER_QUERIES,
               o_eptr = o_uidx + 8 * TOPN_FILTER_QUERIES, o_vecs = o_eptr + 8 * (TOPN_FILTER_QUERIES + 1),
               o_eidx = o_vecs + ((own_vectors ? sizeof(float) * (size_t)ps.n_vecs * (size_t)k : 0) + 15) / 16 * 16,
               total = o_eidx + 8 * (size_t)n_ex;
  if (total > sl.in_cap) {
    HIPCHK(h, hipStreamSynchronize(stream));  // an earlier copy may still be reading the old block
    if (sl.h_in) (void)hipHostFree(sl.h_in);
    sl.h_in = nullptr;
    sl.in_cap = 0;
    HIPCHK(h, hipHostMalloc(&sl.h_in, total + total / 2, hipHostMallocDefault));
    sl.in_cap = total + total / 2;
  }
  if (int rc = topn_grow(h, stream, sl.d_in, sl.din_cap, total + total / 2)) return rc;
  uint8_t* in = sl.h_in;
  int32_t* vptr = reinterpret_cast<int32_t*>(in + o_vptr);
  for (int q = 0; q <= ps.nq; ++q) vptr[q] = (rq.user_idx || !rq.vec_ptr) ? q : (int32_t)(rq.vec_ptr[ps.q0 + q] - ps.v0);
  ps.have_rows = ps.have_excl = false;
  sl.d_vptr = reinterpret_cast<const int32_t*>(sl.d_in + o_vptr);
  sl.d_rows = reinterpret_cast<const int64_t*>(sl.d_in + o_rows);
  sl.d_excl_ptr = reinterpret_cast<const int64_t*>(sl.d_in + o_eptr);
  sl.d_excl_idx = reinterpret_cast<const int64_t*>(sl.d_in + o_eidx);
  if (rq.user_idx) {
    std::memcpy(in + o_uidx, rq.user_idx + ps.q0, sizeof(int64_t) * (size_t)ps.nq);
    sl.d_vecs = x.F;
    sl.d_vrow = reinterpret_cast<const int64_t*>(sl.d_in + o_uidx);
    if (rq.skip_known || rq.skip_known_q) {
      int64_t* rows = reinterpret_cast<int64_t*>(in + o_rows);
      for (int q = 0; q < ps.nq; ++q)
        rows[q] = (!rq.skip_known_q || rq.skip_known_q[ps.q0 + q]) ? rq.user_idx[ps.q0 + q] - x.row_offset : -1;
      ps.have_rows = true;
    }
  } else {
    std::memcpy(in + o_vecs, rq.vectors + ps.v0 * k, sizeof(float) * (size_t)ps.n_vecs * (size_t)k);
    sl.d_vecs = reinterpret_cast<const float*>(sl.d_in + o_vecs);
    sl.d_vrow = nullptr;
  }
  if (n_ex > 0) {
    int64_t* eptr = reinterpret_cast<int64_t*>(in + o_eptr);
    for (int q = 0; q <= ps.nq; ++q) eptr[q] = rq.excl_ptr[ps.q0 + q] - rq.excl_ptr[ps.q0];
    std::memcpy(in + o_eidx, rq.excl_idx + rq.excl_ptr[ps.q0], sizeof(int64_t) * (size_t)n_ex);
    ps.have_excl = true;
  }
  HIPCHK(h, hipMemcpyAsync(sl.d_in, in, total, hipMemcpyHostToDevice, stream));
  return MALS_OK;
}

// ---- dense path: exact scores of every item (on the caller's stream, with slot 0's input block; nothing else of the
// workspace is in flight when it runs) -------------------------------------------------------------------------------
int topn_select_threshold(mals_handle h, const float* d_scores, int64_t n_row, int nq, int how_many, TopnState* d_st, unsigned* d_hist,
                          unsigned* slabs_out) {
  hipLaunchKernelGGL(topn_init_kernel, dim3((unsigned)((nq * 256 + 255) / 256)), dim3(256), 0, h->stream, d_st, d_hist, nq, how_many);
  const unsigned slabs = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_row + 4095) / 4096, (int64_t)(h->n_cu * 8 + nq - 1) / nq));
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(topn_hist_kernel, dim3(slabs, (unsigned)nq), dim3(256), 0, h->stream, d_scores, n_row, pass, d_st, d_hist);
    hipLaunchKernelGGL(topn_pick_kernel, dim3((unsigned)nq), dim3(256), 0, h->stream, d_st, d_hist, pass);
  }
  HIPCHK(h, hipGetLastError());
  *slabs_out = slabs;
  return MALS_OK;
}

int topn_pass_dense(mals_handle h, TopnWorkspace* w, TopnSlot& sl, const TopnRequest& rq, const TopnPass& ps) {
  SideState& y = h->side[MALS_SIDE_Y];
  SideState& x = h->side[MALS_SIDE_X];
  const int k = h->cfg.features, nq = ps.nq, how_many = rq.how_many;
  const int64_t n_items = y.n_total;
  const int cap_ties = 1024;
  const size_t per_q = 2 * ((size_t)how_many + cap_ties);
  if (int rc = topn_grow(h, h->stream, w->d_scores, w->scores_cap, (size_t)TOPN_MAX_QUERIES * (size_t)n_items)) return rc;
  if (int rc = topn_grow(h, h->stream, w->d_sel, w->sel_cap, (size_t)TOPN_MAX_QUERIES * per_q)) return rc;
  if (!w->d_state) HIPCHK(h, hipMalloc(&w->d_state, sizeof(TopnState) * TOPN_MAX_QUERIES));
  if (!w->d_hist) HIPCHK(h, hipMalloc(&w->d_hist, sizeof(unsigned) * 256 * TOPN_MAX_QUERIES));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_items + 63) / 64, (int64_t)h->n_cu * 8));
  hipLaunchKernelGGL(topn_exact_dense_kernel, dim3(grid), dim3(256), sizeof(float) * 64 * (size_t)(k + 1), h->stream, y.F, n_items, k, sl.d_vecs,
                     sl.d_vrow, sl.d_vptr, nq, w->d_scores);
  if (ps.have_rows)
    hipLaunchKernelGGL(topn_mask_kernel, dim3(64, (unsigned)nq), dim3(256), 0, h->stream, h->known_ptr ? h->known_ptr : x.row_ptr,
                       h->known_ptr ? h->known_idx : x.col, sl.d_rows, nq, 1, n_items, w->d_scores);
  if (ps.have_excl)
    hipLaunchKernelGGL(topn_exclude_kernel, dim3(64, (unsigned)nq), dim3(256), 0, h->stream, sl.d_excl_ptr, sl.d_excl_idx, nq, n_items, 1, n_items,
                       w->d_scores);
  if (h->tag_bits)
    hipLaunchKernelGGL(topn_mask_tags_kernel, dim3((unsigned)(((n_items + 31) / 32 + 255) / 256)), dim3(256), 0, h->stream, h->tag_bits, nq, n_items,
                       w->d_scores);
  unsigned slabs = 1;
  if (int rc = topn_select_threshold(h, w->d_scores, n_items, nq, how_many, w->d_state, w->d_hist, &slabs)) return rc;
  hipLaunchKernelGGL(topn_collect_kernel, dim3(slabs, (unsigned)nq), dim3(256), 0, h->stream, w->d_scores, n_items, w->d_state, how_many, cap_ties,
                     w->d_sel);
  HIPCHK(h, hipGetLastError());
  std::vector<uint32_t> out((size_t)nq * per_q);
  std::vector<TopnState> st((size_t)nq);
  HIPCHK(h, hipMemcpyAsync(out.data(), w->d_sel, sizeof(uint32_t) * out.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(st.data(), w->d_state, sizeof(TopnState) * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<float> row;
  std::vector<TopnCand> cand;
  for (int q = 0; q < nq; ++q) {
    const uint32_t* o = &out[(size_t)q * per_q];
    const uint32_t above = st[(size_t)q].above, ties_total = st[(size_t)q].ties;
    const uint32_t ties_stored = std::min<uint32_t>(ties_total, (uint32_t)cap_ties);
    const uint32_t need_ties = above < (uint32_t)how_many ? (uint32_t)how_many - above : 0;
    cand.clear();
    if (above > (uint32_t)how_many || (ties_stored < ties_total && need_ties > 0)) {
      // more ties at the N-th score than the selection buffer holds (e.g. a block of identical items): which of
      // them have the lowest indices is not known from an unordered subset -- resolve this query from its score row
      row.resize((size_t)n_items);
      HIPCHK(h, hipMemcpy(row.data(), w->d_scores + (size_t)q * (size_t)n_items, sizeof(float) * (size_t)n_items, hipMemcpyDeviceToHost));
      const uint32_t ninf = score_key(-std::numeric_limits<float>::infinity());
      for (int64_t i = 0; i < n_items; ++i) {
        const uint32_t kk = score_key(row[(size_t)i]);
        if (kk > ninf && kk >= st[(size_t)q].prefix) cand.push_back({kk, i});
      }
    } else {
      for (uint32_t p = 0; p < above; ++p) cand.push_back({o[2 * p + 1], (int64_t)o[2 * p]});
      for (uint32_t p = 0; p < ties_stored; ++p) cand.push_back({o[2 * (how_many + p) + 1], (int64_t)o[2 * (how_many + p)]});
    }
    const TopnOut o_q = topn_out(rq, (size_t)(ps.q0 + q));
    if (!topn_emit(cand, how_many, o_q.items, o_q.scores, o_q.n))
      return fail(h, MALS_INVALID_ARG, "Bad recommendation value: a non-finite score (non-finite factors; RecommendIterator.java:105 throws IllegalStateException)");
  }
  return MALS_OK;
}

// ---- filter path ------------------------------------------------------------------------------------------------------
// query tiles per pass the LDS image of the split query operands allows (64 KB): S steps x NT tiles x 2 KB
int topn_max_tiles(int S) { return S == 1 ? 16 : S == 2 ? 15 : S == 3 ? 10 : 7; }

constexpr int TOPN_WAVE_CAP = 2048;  // hits a wave of the filter kernel can record (expected: a hundred)

// topn_stream_kernel: QT = query tiles per wave, 4 QT per workgroup.  MODE 0: *n_out = workgroups of the sample (16
// buckets each); MODE 1: *n_out = waves of the filter (one hit list each).
template <int S, int QT, int MODE>
int topn_launch_stream_QT(mals_handle h, TopnSlot& sl, const float* Y, int64_t n_items, int k, int nq, int tile_stride, int cap, int* n_out) {
  const int64_t tiles = (n_items + 16 * (int64_t)tile_stride - 1) / (16 * (int64_t)tile_stride);
  const int64_t stages = (tiles + 3) / 4;
  // resident workgroups per CU: what the instantiation's registers and LDS (two buffers of 4 (S + 1) KB) allow, asked once
  const int lm = k == 32 * S ? 1 : (k & 3) == 0 ? 2 : (k & 1) == 0 ? 3 : 0;  // topn_stream_kernel's load mode
  static int cached[4] = {0, 0, 0, 0};
  int& per_cu_cached = cached[lm];
  if (!per_cu_cached) {
    int nb = 0;
    const hipError_t e = lm == 1   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topn_stream_kernel<S, QT, MODE, 1>, 256, 0)
                         : lm == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topn_stream_kernel<S, QT, MODE, 2>, 256, 0)
                         : lm == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topn_stream_kernel<S, QT, MODE, 3>, 256, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topn_stream_kernel<S, QT, MODE, 0>, 256, 0);
    per_cu_cached = (e == hipSuccess && nb > 0) ? std::min(nb, 6) : 2;
  }
  int per_cu = per_cu_cached;
  if (const char* e = std::getenv("MALS_TOPN_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e));
  // the filter is a persistent grid; the sample's workgroups each keep 16 buckets per query
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(stages, MODE == 0 ? TOPN_SAMPLE_GROUPS : (int64_t)h->n_cu * per_cu));
  if (MODE == 1) {
    const size_t nw = (size_t)grid * 4;
    if (nw * TOPN_WAVE_CAP > sl.wh_cap) {
      HIPCHK(h, hipStreamSynchronize(sl.stream));
      free_dev(sl.d_whits);
      free_dev(sl.d_wcount);
      sl.wh_cap = 0;
      HIPCHK(h, hipMalloc(&sl.d_whits, sizeof(uint2) * nw * TOPN_WAVE_CAP));
      HIPCHK(h, hipMalloc(&sl.d_wcount, sizeof(unsigned) * (nw + 1)));
      sl.wh_cap = nw * TOPN_WAVE_CAP;
    }
    *n_out = (int)nw;
  } else {
    *n_out = (int)grid;
  }
#define MALS_TOPN_GO(LM)                                                                                                              \
  hipLaunchKernelGGL((topn_stream_kernel<S, QT, MODE, LM>), dim3(grid), dim3(256), 0, sl.stream, Y, n_items, k,                           \
                     static_cast<const bf16x8*>(sl.d_img), nq, tile_stride, sl.d_bmax, sl.d_bidx, sl.d_tau, TOPN_WAVE_CAP, sl.d_wcount,   \
                     sl.d_whits, cap, sl.d_count, sl.d_cand, sl.d_count + (size_t)TOPN_FILTER_QUERIES * TOPN_COUNT_STRIDE)
  if (lm == 1) MALS_TOPN_GO(1);
  else if (lm == 2) MALS_TOPN_GO(2);
  else if (lm == 3) MALS_TOPN_GO(3);
  else MALS_TOPN_GO(0);
#undef MALS_TOPN_GO
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
template <int MODE>
int topn_launch_stream(mals_handle h, TopnSlot& sl, int S, int nt, const float* Y, int64_t n_items, int k, int nq, int tile_stride, int cap, int* n_out) {
  const int qt = (nt + 3) / 4;
#define MALS_TOPN_STREAM(SS, QQ) return topn_launch_stream_QT<SS, QQ, MODE>(h, sl, Y, n_items, k, nq, tile_stride, cap, n_out)
  switch (S) {
    case 1:
      if (qt <= 1) MALS_TOPN_STREAM(1, 1);
      if (qt == 2) MALS_TOPN_STREAM(1, 2);
      if (qt == 3) MALS_TOPN_STREAM(1, 3);
      MALS_TOPN_STREAM(1, 4);
    case 2:
      if (qt <= 1) MALS_TOPN_STREAM(2, 1);
      if (qt == 2) MALS_TOPN_STREAM(2, 2);
      if (qt == 3) MALS_TOPN_STREAM(2, 3);
      MALS_TOPN_STREAM(2, 4);
    case 3:  // at most 10 query tiles per pass
      if (qt <= 1) MALS_TOPN_STREAM(3, 1);
      if (qt == 2) MALS_TOPN_STREAM(3, 2);
      MALS_TOPN_STREAM(3, 3);
    default:  // at most 7
      if (qt <= 1) MALS_TOPN_STREAM(4, 1);
      MALS_TOPN_STREAM(4, 2);
  }
#undef MALS_TOPN_STREAM
}

struct TopnFilterPlan {
  int S, cap, cap_pad, tile_stride;
  size_t stage_bytes;
};
TopnFilterPlan topn_plan(mals_handle h, int how_many) {
  TopnFilterPlan p;
  const int64_t n_items = h->side[MALS_SIDE_Y].n_total;
  p.S = (h->cfg.features + 31) / 32;
  p.cap = 48 * how_many + 2048;
  p.cap_pad = 1;
  while (p.cap_pad < p.cap) p.cap_pad <<= 1;
  // sample items: expected candidates per query = how_many x stride, spread like a negative binomial (sigma = mean / sqrt(N)).
  // An eighth of the items: the sample kernel keeps only bucket maxima, so it costs an eighth of a filter pass whatever the
  // catalogue, and every candidate less is less hit bookkeeping in the filter, less to scatter and rescore (16384 -> 131072 at
  // a million items: 1.4x the queries per second).  A sample that does NOT grow with the catalogue lets the candidates grow
  // with it instead: at 10M items a 131072-item sample put every fourth pass of 240 queries over its candidate buffers.
  int64_t target = std::max<int64_t>(512 * (int64_t)how_many, std::max<int64_t>(16384, n_items / 8));
  if (const char* e = std::getenv("MALS_TOPN_SAMPLE_ITEMS")) target = std::max<int64_t>(1024, std::atoll(e));  // tuning override
  p.tile_stride = (int)std::max<int64_t>(1, n_items / target);
  p.stage_bytes = (size_t)TOPN_FILTER_QUERIES * ((size_t)how_many * 8 + 8) + 16;  // pairs | counts | taus | overflow word
  return p;
}

// the kernels of one pass on the slot's stream (plain launches, or recorded into a graph while the stream is capturing)
int topn_pass_filter_launch(mals_handle h, TopnSlot& sl, const TopnRequest& rq, const TopnPass& ps, const TopnFilterPlan& p) {
  SideState& y = h->side[MALS_SIDE_Y];
  SideState& x = h->side[MALS_SIDE_X];
  const int k = h->cfg.features, nq = ps.nq, how_many = rq.how_many;
  const int64_t n_items = y.n_total;
  const int nt = (nq + 15) / 16;
  hipStream_t st = sl.stream;
  unsigned* d_overflow = sl.d_count + (size_t)TOPN_FILTER_QUERIES * TOPN_COUNT_STRIDE;
  const int64_t* d_rows = ps.have_rows ? sl.d_rows : nullptr;
  const int64_t* k_ptr = h->known_ptr ? h->known_ptr : x.row_ptr;   // knownItemIDs if the caller installed them, else the rows of R
  const int32_t* k_idx = h->known_ptr ? h->known_idx : x.col;
  const int64_t* d_eptr = ps.have_excl ? sl.d_excl_ptr : nullptr;
  const int64_t* d_eidx = ps.have_excl ? sl.d_excl_idx : nullptr;
  // 0. the queries as matrix operands (every tile an instantiation may touch: padding queries never produce a hit)
  hipLaunchKernelGGL(topn_prepare_kernel, dim3(16), dim3(256), 0, st, sl.d_vecs, sl.d_vrow, sl.d_vptr, nq, k, p.S,
                     static_cast<bf16x8*>(sl.d_img), sl.d_count, d_overflow);
  // 1. sample: bucket maxima of the lower bounds of every tile_stride-th tile; 2. threshold (buckets won by known items dropped)
  int n_groups = 0, n_fw = 0;
  if (int rc = topn_launch_stream<0>(h, sl, p.S, nt, y.F, n_items, k, nq, p.tile_stride, p.cap, &n_groups)) return rc;
  hipLaunchKernelGGL(topn_threshold_kernel, dim3((unsigned)nq), dim3(1024), 0, st, sl.d_bmax, sl.d_bidx, n_groups, how_many, k_ptr, k_idx, d_rows,
                     d_eptr, d_eidx, n_items, p.tile_stride, h->tag_bits, sl.d_tau);
  // 3. filter, 4. exact scores of the hits (known items dropped), 5. the N best -- written straight into the slot's pinned
  // block (device-visible host memory: no copy kernel, no copy call)
  // (the filter's waves scatter their own hits into the per-query candidate lists: no kernel in between)
  if (int rc = topn_launch_stream<1>(h, sl, p.S, nt, y.F, n_items, k, nq, 1, p.cap, &n_fw)) return rc;
  hipLaunchKernelGGL(topn_rescore_kernel, dim3(8, (unsigned)nq), dim3(64), sizeof(float) * 64 * (size_t)(k + 1), st, y.F, k, sl.d_vecs, sl.d_vrow, sl.d_vptr, sl.d_count, p.cap,
                     sl.d_cand, k_ptr, k_idx, d_rows, d_eptr, d_eidx, h->tag_bits, sl.d_pairs, d_overflow);
  uint8_t* o = sl.h_stage;
  const size_t o_cnt = sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many, o_tau = o_cnt + sizeof(unsigned) * TOPN_FILTER_QUERIES,
               o_ovf = o_tau + sizeof(float) * TOPN_FILTER_QUERIES;
  hipLaunchKernelGGL(topn_final_kernel, dim3((unsigned)nq), dim3(256), sizeof(uint64_t) * (size_t)p.cap, st, sl.d_pairs, sl.d_count, p.cap,
                     how_many, reinterpret_cast<uint64_t*>(o), reinterpret_cast<unsigned*>(o + o_cnt), sl.d_tau, reinterpret_cast<float*>(o + o_tau),
                     d_overflow, reinterpret_cast<unsigned*>(o + o_ovf));
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}

// enqueue one pass on its slot's stream; its results land in the slot's pinned block behind the slot's event
int topn_pass_filter_enqueue(mals_handle h, TopnSlot& sl, const TopnRequest& rq, const TopnPass& ps, const TopnFilterPlan& p) {
  hipStream_t st = sl.stream;
  if (!sl.d_tau) {
    HIPCHK(h, hipMalloc(&sl.d_tau, sizeof(float) * TOPN_FILTER_QUERIES));
    HIPCHK(h, hipMalloc(&sl.d_count, sizeof(unsigned) * (TOPN_FILTER_QUERIES * TOPN_COUNT_STRIDE + 1)));  // padded counters, then the overflow word
    HIPCHK(h, hipMalloc(&sl.d_img, (size_t)16 * 5 * 64 * 16));
    HIPCHK(h, hipMalloc(&sl.d_bmax, sizeof(float) * TOPN_FILTER_QUERIES * 16 * TOPN_SAMPLE_GROUPS));
    HIPCHK(h, hipMalloc(&sl.d_bidx, sizeof(uint32_t) * TOPN_FILTER_QUERIES * 16 * TOPN_SAMPLE_GROUPS));
  }
  if (int rc = topn_grow(h, st, sl.d_pairs, sl.pairs_cap, (size_t)TOPN_FILTER_QUERIES * (size_t)p.cap)) return rc;
  if (int rc = topn_grow(h, st, sl.d_cand, sl.cand_cap, (size_t)TOPN_FILTER_QUERIES * (size_t)p.cap)) return rc;
  // (one graph launch per pass instead of nine kernel launches was tried: no faster -- the device, not the host's launch
  // calls, sets the pace even at 64 queries per pass)
  if (int rc = topn_pass_filter_launch(h, sl, rq, ps, p)) return rc;
  HIPCHK(h, hipEventRecord(sl.ev, st));
  return MALS_OK;
}

// decode a finished pass; failed[q] != 0: query q of the pass overflowed its candidate buffer or had a thin sample (the dense
// path answers it); all of them if a wave's hit list overflowed (whose hits are missing is not known)
int topn_pass_filter_finish(mals_handle h, TopnSlot& sl, const TopnRequest& rq, const TopnPass& ps, const TopnFilterPlan& p,
                            std::vector<uint8_t>& failed, bool* any_failed) {
  HIPCHK(h, hipEventSynchronize(sl.ev));
  const int how_many = rq.how_many;
  const uint8_t* st = sl.h_stage;
  const uint64_t* outp = reinterpret_cast<const uint64_t*>(st);
  const unsigned* count = reinterpret_cast<const unsigned*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many);
  const float* tau = reinterpret_cast<const float*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many + sizeof(unsigned) * TOPN_FILTER_QUERIES);
  const unsigned wave_overflow = *reinterpret_cast<const unsigned*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many +
                                                                    (sizeof(unsigned) + sizeof(float)) * TOPN_FILTER_QUERIES);
  failed.assign((size_t)ps.nq, 0);
  *any_failed = false;
  for (int q = 0; q < ps.nq; ++q)
    if (wave_overflow != 0 || count[q] > (unsigned)p.cap || !(tau[q] > -std::numeric_limits<float>::infinity())) {
      failed[(size_t)q] = 1;
      *any_failed = true;
    }
  for (int q = 0; q < ps.nq; ++q) {
    if (failed[(size_t)q]) continue;
    const TopnOut o_q = topn_out(rq, (size_t)(ps.q0 + q));
    int n = 0;
    for (int j = 0; j < how_many; ++j) {
      const uint64_t pr = outp[(size_t)q * how_many + j];
      if (pr != 0) {
        o_q.items[j] = (int64_t)(0xffffffffu - (uint32_t)pr);
        o_q.scores[j] = key_score((uint32_t)(pr >> 32));
        ++n;
      } else {
        o_q.items[j] = -1;
        o_q.scores[j] = -std::numeric_limits<float>::infinity();
      }
    }
    if (o_q.n) *o_q.n = n;
  }
  return MALS_OK;
}

bool topn_dense_only(mals_handle h, int how_many) {
  const int64_t n_items = h->side[MALS_SIDE_Y].n_total;
  return n_items < 131072 || n_items >= 0xffffffffll || how_many > TOPN_FILTER_MAX_N || n_items / 16 < 64 * (int64_t)how_many ||
         std::getenv("MALS_TOPN_FULL");
}

// streams, events and pinned result blocks of the slots; every slot stream ordered after the work already on the handle's
int topn_prepare_slots(mals_handle h, TopnWorkspace* w, const TopnFilterPlan& p) {
  for (TopnSlot& sl : w->slot) {
    if (!sl.stream) HIPCHK(h, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    if (!sl.ev) HIPCHK(h, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.stage_cap < p.stage_bytes) {
      HIPCHK(h, hipStreamSynchronize(sl.stream));
      if (sl.h_stage) (void)hipHostFree(sl.h_stage);
      sl.h_stage = nullptr;
      sl.stage_cap = 0;
      HIPCHK(h, hipHostMalloc(&sl.h_stage, p.stage_bytes, hipHostMallocDefault));
      sl.stage_cap = p.stage_bytes;
    }
  }
  if (!w->ev_begin) HIPCHK(h, hipEventCreateWithFlags(&w->ev_begin, hipEventDisableTiming));
  return MALS_OK;
}

// decode the pass in slot s; the queries the filter could not answer (rare) go through the dense path, run by run (the
// slot's input block is free again: its pass has finished)
int topn_finish_slot(mals_handle h, TopnWorkspace* w, int s, const TopnRequest& rq, const TopnPass& done, const TopnFilterPlan& p,
                     std::vector<uint8_t>& failed) {
  bool any_failed = false;
  if (int rc = topn_pass_filter_finish(h, w->slot[s], rq, done, p, failed, &any_failed)) return rc;
  if (any_failed && std::getenv("MALS_TOPN_DEBUG")) {
    int nf = 0;
    for (uint8_t f : failed) nf += f;
    std::fprintf(stderr, "[mals top-N] pass at query %d (%d queries): %d to the dense path\n", done.q0, done.nq, nf);
  }
  if (any_failed) {
    for (int q = 0; q < done.nq;) {
      if (!failed[(size_t)q]) {
        ++q;
        continue;
      }
      int e = q;
      while (e < done.nq && failed[(size_t)e] && e - q < TOPN_MAX_QUERIES) ++e;
      TopnPass ps;
      ps.q0 = done.q0 + q;
      ps.nq = e - q;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (int rc = topn_upload_pass(h, w->slot[s], h->stream, rq, ps)) return rc;
      if (int rc = topn_pass_dense(h, w, w->slot[s], rq, ps)) return rc;
      q = e;
    }
  }
  return MALS_OK;
}

int topn_run(mals_handle h, const TopnRequest& rq) {
  if (!h->tn_ws) h->tn_ws = new TopnWorkspace();
  TopnWorkspace* w = static_cast<TopnWorkspace*>(h->tn_ws);
  if (topn_dense_only(h, rq.how_many)) {
    for (int q0 = 0; q0 < rq.n_queries; q0 += TOPN_MAX_QUERIES) {
      TopnPass ps;
      ps.q0 = q0;
      ps.nq = std::min(TOPN_MAX_QUERIES, rq.n_queries - q0);
      if (int rc = topn_upload_pass(h, w->slot[0], h->stream, rq, ps)) return rc;
      if (int rc = topn_pass_dense(h, w, w->slot[0], rq, ps)) return rc;
    }
    return MALS_OK;
  }
  const TopnFilterPlan p = topn_plan(h, rq.how_many);
  if (int rc = topn_prepare_slots(h, w, p)) return rc;
  // whatever the caller's stream still has to do (an iteration, a factor upload) comes first
  HIPCHK(h, hipEventRecord(w->ev_begin, h->stream));
  for (TopnSlot& sl : w->slot) HIPCHK(h, hipStreamWaitEvent(sl.stream, w->ev_begin, 0));
  int per_pass = 16 * topn_max_tiles(p.S);
  if (const char* e = std::getenv("MALS_TOPN_QUERIES_PER_PASS")) per_pass = std::max(16, std::min(per_pass, std::atoi(e) / 16 * 16));  // tuning override
  int n_slots = TOPN_SLOTS;
  if (const char* e = std::getenv("MALS_TOPN_SLOTS")) n_slots = std::max(1, std::min(TOPN_SLOTS, std::atoi(e)));  // tuning override
  // Passes go round the slots; pass i is decoded after pass i + n_slots - 1 has been enqueued, right before its slot is
  // needed again: the device never waits for the host between passes.
  TopnPass inflight[TOPN_SLOTS];
  bool busy[TOPN_SLOTS] = {};
  std::vector<uint8_t> failed;
  auto finish = [&](int s) -> int {
    if (!busy[s]) return MALS_OK;
    busy[s] = false;
    return topn_finish_slot(h, w, s, rq, inflight[s], p, failed);
  };
  int s = 0;
  int rc_all = MALS_OK;
  for (int q0 = 0; q0 < rq.n_queries && rc_all == MALS_OK; q0 += per_pass) {
    if ((rc_all = finish(s))) break;
    TopnPass ps;
    ps.q0 = q0;
    ps.nq = std::min(per_pass, rq.n_queries - q0);
    if ((rc_all = topn_upload_pass(h, w->slot[s], w->slot[s].stream, rq, ps))) break;
    if ((rc_all = topn_pass_filter_enqueue(h, w->slot[s], rq, ps, p))) break;
    inflight[s] = ps;
    busy[s] = true;
    s = (s + 1) % n_slots;
  }
  // drain in enqueue order (also on an error: nothing of this call may still be running when it returns)
  for (int i = 0; i < n_slots; ++i) {
    const int t = (s + i) % n_slots;
    if (rc_all != MALS_OK) {
      if (w->slot[t].stream) (void)hipStreamSynchronize(w->slot[t].stream);
      busy[t] = false;
    } else {
      rc_all = finish(t);
    }
  }
  return rc_all;
}

// ---- the serving front: many request threads, one handle ---------------------------------------------------------------
// The reference's top-N is entered by every request thread of the servlet container at once (ServerRecommender.java:359-441
// -> multithreadedTopN :443-508), one user per call.  Here a call is cheap only as part of a PASS (one read of Y answers up
// to 240 queries), so concurrent calls are folded into passes: a call becomes a ticket in the handle's queue; the first
// thread that finds no leader becomes the leader, packs whatever is queued into the next pass, enqueues it on a slot,
// decodes finished passes and wakes their callers.  When its own ticket is answered it hands leadership to a caller that is
// still waiting.  While a pass is on the device the tickets of the calls that arrive pile up and form the next one (group
// commit); `depth` passes are in flight at most (2: one streaming Y, one being prepared behind it; more would only split the
// same callers over more reads of Y).  Calls that are passes of their own already (>= TOPN_FRONT_BULK queries, caller's
// vectors, how_many the filter does not take) run exclusively, in queue order, through topn_run.
constexpr int TOPN_FRONT_BULK = 64;

struct TopnTicket {
  // a small by-user call ...
  const int64_t* user_idx = nullptr;
  int n = 0, how_many = 0;
  bool skip_known = false;
  int64_t* item_out = nullptr;
  float* score_out = nullptr;
  int32_t* n_out = nullptr;
  // ... or a small call with the caller's own vectors (recommendToAnonymous, recommendToMany: SR:366-441,561-606): n queries,
  // query q owns vectors [vec_ptr[q], vec_ptr[q+1]) (NULL: one each), optional exclusion lists
  const float* vectors = nullptr;
  const int64_t* vec_ptr = nullptr;
  const int64_t* excl_ptr = nullptr;
  const int64_t* excl_idx = nullptr;
  // ... or a whole request of its own
  const TopnRequest* bulk = nullptr;
  int rc = MALS_OK;
  std::string err;
  bool done = false;          // under the front's mutex
  bool is_leader = false;
  // How a waiting caller hears from the leader: `sig` (TOPN_SIG_DONE: the answer is in the caller's arrays; TOPN_SIG_LEAD:
  // take over as leader).  A caller first polls it (an answer takes 100-200 us: most arrive while it polls, and then the
  // leader's "wake-up" is one store, not a futex call per caller -- 33 of those per pass were half of a pass's host time at
  // 128 callers), then blocks on its own condition variable.  Everything the leader touches of a ticket it touches under
  // the ticket's mutex, and a caller takes that mutex once before it returns: the ticket lives on the caller's stack.
  std::atomic<int> sig{0};
  std::atomic<bool> blocked{false};
  std::mutex mu;
  std::condition_variable cv;
  // Waking the callers of a finished pass is a futex call each (they block): 28 of them were 110 us of the leader's time
  // per pass at 128 callers, more than the pass takes on the device.  The leader wakes ONE caller of the pass and leaves it
  // the list of the others (to_wake); that caller wakes them on its way out.  A ticket on such a list has visit_pending set
  // until its waker has let go of it, and does not return before.
  std::vector<TopnTicket*> to_wake;
  std::atomic<bool> visit_pending{false};
};
enum { TOPN_SIG_DONE = 1, TOPN_SIG_LEAD = 2 };

void topn_signal(TopnTicket* t, int bits) {
  std::lock_guard<std::mutex> lk(t->mu);
  t->sig.fetch_or(bits, std::memory_order_seq_cst);
  if (t->blocked.load(std::memory_order_seq_cst)) t->cv.notify_one();
}

// returns the signal bits seen (never 0)
int topn_wait_signal(TopnTicket* t, int spin_us) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0;; ++it) {
    const int sg = t->sig.load(std::memory_order_acquire);
    if (sg) return sg;
    if ((it & 15) == 15) {
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)spin_us) break;
      std::this_thread::yield();
    } else {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
      __builtin_ia32_pause();
#endif
    }
  }
  std::unique_lock<std::mutex> lk(t->mu);
  t->blocked.store(true, std::memory_order_seq_cst);
  while (!t->sig.load(std::memory_order_seq_cst)) t->cv.wait(lk);
  t->blocked.store(false, std::memory_order_seq_cst);
  return t->sig.load(std::memory_order_seq_cst);
}

struct TopnFrontPass {
  std::vector<TopnTicket*> tickets;
  std::vector<int64_t> users;
  std::vector<uint8_t> skip;
  std::vector<TopnOut> outs;
  // a pass of by-vector calls: the callers' vectors and exclusion lists back to back
  std::vector<float> vecs;
  std::vector<int64_t> vptr, eptr, eidx;
  TopnRequest rq;
  TopnPass ps;
  TopnFilterPlan plan;
};

struct TopnFront {
  std::mutex mu;
  std::deque<TopnTicket*> queue;
  bool leader = false;
  TopnFrontPass inflight[TOPN_SLOTS];
  bool busy[TOPN_SLOTS] = {};
  int n_busy = 0, next_slot = 0, oldest = 0;
  int depth = 2;
  int spin_us = 0;     // how long a waiting caller polls before it blocks (mals_recommend_set_spin_us; MALS_TOPN_FRONT_SPIN_US at
                       // mals_create).  0: block at once -- polling callers compete with the HIP runtime's own threads for cores
  std::vector<uint8_t> failed;
  // counters (mals_recommend_front_stats)
  uint64_t calls = 0, queries = 0, passes = 0, bulk_calls = 0;
};

TopnFront* topn_front(mals_handle h) { return static_cast<TopnFront*>(h->tn_front); }

// leader only, mutex NOT held: the pass in `fp` (already filled) onto slot s
int topn_front_enqueue(mals_handle h, TopnWorkspace* w, int s, TopnFrontPass& fp) {
  if (int rc = topn_prepare_slots(h, w, fp.plan)) return rc;
  HIPCHK(h, hipEventRecord(w->ev_begin, h->stream));
  HIPCHK(h, hipStreamWaitEvent(w->slot[s].stream, w->ev_begin, 0));
  if (int rc = topn_upload_pass(h, w->slot[s], w->slot[s].stream, fp.rq, fp.ps)) return rc;
  return topn_pass_filter_enqueue(h, w->slot[s], fp.rq, fp.ps, fp.plan);
}

void topn_front_complete(TopnFrontPass& fp, int rc, const std::string& err) {  // mutex held
  TopnTicket* waker = nullptr;
  size_t waiting = 0;
  for (TopnTicket* t : fp.tickets) {
    t->rc = rc;
    if (rc != MALS_OK) t->err = err;
    t->done = true;
    if (!t->is_leader) ++waiting;
  }
  if (waiting >= 16) {
    // groups of 8: the leader wakes the first caller of every group and leaves it the other seven (measured at 128 callers,
    // ~37 per pass: 2.4e5 -> 4.3e5 queries/s, p50 510 -> 260 us; below 16 per pass the leader's own wake-ups cost less than the
    // extra hop adds to the tail, so small passes are woken directly)
    size_t in_group = 0;
    std::vector<TopnTicket*> wakers;
    for (TopnTicket* t : fp.tickets) {
      if (t->is_leader) continue;
      if (in_group == 0) {
        waker = t;
        waker->to_wake.reserve(7);
        wakers.push_back(waker);
      } else {
        t->visit_pending.store(true, std::memory_order_seq_cst);   // before the answer becomes visible
        waker->to_wake.push_back(t);
        t->sig.fetch_or(TOPN_SIG_DONE, std::memory_order_seq_cst);  // (a caller that is polling sees it at once; one that blocks is woken by its group's first)
      }
      in_group = (in_group + 1) % 8;
    }
    for (TopnTicket* w : wakers) topn_signal(w, TOPN_SIG_DONE);
  } else {
    for (TopnTicket* t : fp.tickets)
      if (!t->is_leader) topn_signal(t, TOPN_SIG_DONE);   // (after this the ticket may be gone)
  }
  fp.tickets.clear();
}

// a caller on its way out: the pass-mates the leader left to it, then its own ticket's last visitors
void topn_ticket_leave(TopnTicket& me) {
  for (TopnTicket* m : me.to_wake) {
    {
      std::lock_guard<std::mutex> lk(m->mu);
      if (m->blocked.load(std::memory_order_seq_cst)) m->cv.notify_one();
    }
    m->visit_pending.store(false, std::memory_order_release);   // the last touch of m
  }
  me.to_wake.clear();
  while (me.visit_pending.load(std::memory_order_acquire)) std::this_thread::yield();
  { std::lock_guard<std::mutex> own(me.mu); }   // the leader has let go of the ticket
}

// The calling thread leads until its own ticket is answered.
void topn_front_lead(mals_handle h, TopnFront* f, std::unique_lock<std::mutex>& lk, TopnTicket* me) {
  if (!h->tn_ws) h->tn_ws = new TopnWorkspace();
  TopnWorkspace* w = static_cast<TopnWorkspace*>(h->tn_ws);
  const int k_tiles = topn_max_tiles((h->cfg.features + 31) / 32);
  auto finish_oldest = [&]() {
    const int s = f->oldest;
    TopnFrontPass& fp = f->inflight[s];
    lk.unlock();
    const int rc = topn_finish_slot(h, w, s, fp.rq, fp.ps, fp.plan, f->failed);
    const std::string err = rc != MALS_OK ? h->err : std::string();
    lk.lock();
    f->busy[s] = false;
    --f->n_busy;
    f->oldest = (s + 1) % f->depth;
    topn_front_complete(fp, rc, err);
  };
  while (!me->done) {
    TopnTicket* head = f->queue.empty() ? nullptr : f->queue.front();
    if (head && (head->bulk || topn_dense_only(h, head->how_many))) {
      // a request that is passes of its own (or a catalogue the dense path answers): alone on the workspace
      while (f->n_busy > 0) finish_oldest();
      if (me->done && head != me) break;   // (own answer arrived while draining: let the successor run it)
      f->queue.pop_front();
      TopnFrontPass one;
      one.tickets.push_back(head);
      if (head->bulk) {
        one.rq = *head->bulk;
        ++f->bulk_calls;
      } else {
        one.rq.n_queries = head->n;
        one.rq.how_many = head->how_many;
        if (head->vectors) {
          one.rq.vectors = head->vectors;
          one.rq.vec_ptr = head->vec_ptr;
          one.rq.excl_ptr = head->excl_ptr;
          one.rq.excl_idx = head->excl_idx;
        } else {
          one.rq.user_idx = head->user_idx;
          one.rq.skip_known = head->skip_known;
        }
        one.rq.item_idx_out = head->item_out;
        one.rq.score_out = head->score_out;
        one.rq.n_out = head->n_out;
      }
      lk.unlock();
      const int rc = topn_run(h, one.rq);
      const std::string err = rc != MALS_OK ? h->err : std::string();
      lk.lock();
      topn_front_complete(one, rc, err);
      continue;
    }
    if (head && f->n_busy < f->depth) {
      // the next pass: whole tickets, one how_many, up to the pass's query capacity
      const int s = f->next_slot;
      TopnFrontPass& fp = f->inflight[s];
      const int cap = 16 * k_tiles;
      const int kf = h->cfg.features;
      fp.tickets.clear(); fp.users.clear(); fp.skip.clear(); fp.outs.clear();
      fp.vecs.clear(); fp.vptr.assign(1, 0); fp.eptr.assign(1, 0); fp.eidx.clear();
      const int how_many = head->how_many;
      const bool by_vector = head->vectors != nullptr;   // a pass holds by-user calls or by-vector calls, not both (where its
                                                         // query vectors come from -- X on the device or an uploaded block -- is per pass)
      bool any_excl = false;
      while (!f->queue.empty()) {
        TopnTicket* t = f->queue.front();
        if (t->bulk || t->how_many != how_many || (t->vectors != nullptr) != by_vector || (int)fp.outs.size() + t->n > cap) break;
        f->queue.pop_front();
        fp.tickets.push_back(t);
        for (int q = 0; q < t->n; ++q) {
          if (by_vector) {
            const int64_t v0 = t->vec_ptr ? t->vec_ptr[q] : q, v1 = t->vec_ptr ? t->vec_ptr[q + 1] : q + 1;
            fp.vecs.insert(fp.vecs.end(), t->vectors + v0 * kf, t->vectors + v1 * kf);
            fp.vptr.push_back(fp.vptr.back() + (v1 - v0));
            if (t->excl_ptr && t->excl_idx) {
              fp.eidx.insert(fp.eidx.end(), t->excl_idx + t->excl_ptr[q], t->excl_idx + t->excl_ptr[q + 1]);
              any_excl = any_excl || t->excl_ptr[q + 1] > t->excl_ptr[q];
            }
            fp.eptr.push_back((int64_t)fp.eidx.size());
          } else {
            fp.users.push_back(t->user_idx[q]);
            fp.skip.push_back(t->skip_known ? 1 : 0);
          }
          fp.outs.push_back({t->item_out + (size_t)q * how_many, t->score_out + (size_t)q * how_many, t->n_out ? t->n_out + q : nullptr});
        }
      }
      fp.rq = TopnRequest();
      fp.rq.n_queries = (int)fp.outs.size();
      fp.rq.how_many = how_many;
      if (by_vector) {
        fp.rq.vectors = fp.vecs.data();
        fp.rq.vec_ptr = fp.vptr.data();
        if (any_excl) {
          fp.rq.excl_ptr = fp.eptr.data();
          fp.rq.excl_idx = fp.eidx.data();
        }
      } else {
        fp.rq.user_idx = fp.users.data();
        fp.rq.skip_known_q = fp.skip.data();
      }
      fp.rq.out_q = fp.outs.data();
      fp.ps = TopnPass();
      fp.ps.q0 = 0;
      fp.ps.nq = fp.rq.n_queries;
      fp.plan = topn_plan(h, how_many);
      f->busy[s] = true;
      ++f->n_busy;
      f->next_slot = (s + 1) % f->depth;
      ++f->passes;
      lk.unlock();
      const int rc = topn_front_enqueue(h, w, s, fp);
      const std::string err = rc != MALS_OK ? h->err : std::string();
      lk.lock();
      if (rc != MALS_OK) {  // nothing of the pass may still run when its callers return
        lk.unlock();
        if (w->slot[s].stream) (void)hipStreamSynchronize(w->slot[s].stream);
        lk.lock();
        f->busy[s] = false;
        --f->n_busy;
        // (slots complete in order: an enqueue failure of the newest pass leaves `oldest` where it was unless it was alone)
        if (f->n_busy == 0) f->oldest = f->next_slot;
        else f->next_slot = s;
        topn_front_complete(fp, rc, err);
      }
      continue;
    }
    if (f->n_busy > 0) {
      finish_oldest();
      continue;
    }
    break;  // (not reached: a leader's own ticket is queued, in flight or done)
  }
}

// every mals_recommend* call ends here
int topn_front_submit(mals_handle h, TopnTicket& me) {
  TopnFront* f = topn_front(h);
  std::unique_lock<std::mutex> lk(f->mu);
  ++f->calls;
  f->queries += (uint64_t)(me.bulk ? me.bulk->n_queries : me.n);
  f->queue.push_back(&me);
  for (;;) {
    if (me.done) break;
    if (!f->leader) {
      f->leader = true;
      me.is_leader = true;
      topn_front_lead(h, f, lk, &me);
      me.is_leader = false;
      f->leader = false;
      // successor: a caller whose answer is still out -- in the oldest pass in flight, else at the head of the queue
      TopnTicket* next = nullptr;
      if (f->n_busy > 0) {
        for (TopnTicket* t : f->inflight[f->oldest].tickets)
          if (!t->done) { next = t; break; }
      }
      if (!next && !f->queue.empty()) next = f->queue.front();
      if (next) topn_signal(next, TOPN_SIG_LEAD);
      continue;
    }
    // somebody leads: wait for the answer, or for the call to lead
    const int spin_us = f->spin_us;
    lk.unlock();
    const int sg = topn_wait_signal(&me, spin_us);
    if (sg & TOPN_SIG_LEAD) me.sig.fetch_and(~TOPN_SIG_LEAD, std::memory_order_seq_cst);
    lk.lock();
  }
  const int rc = me.rc;
  if (rc != MALS_OK) h->err = me.err;
  lk.unlock();
  topn_ticket_leave(me);
  return rc;
}
