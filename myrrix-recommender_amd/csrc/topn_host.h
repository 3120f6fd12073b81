// topn_host.h -- host side of top-N scoring (mals_recommend*, include/myrrix_als.h; kernels and the argument for
// exactness in topn_kernels.h).  Included by mals_api.hip inside its anonymous namespace, after mals_handle_s.
#pragma once

struct TopnWorkspace {
  // inputs of one pass: ONE device block, the image of the pinned block of the pass's slot (one copy); the pointers
  // below are views into it
  uint8_t* d_in = nullptr;
  size_t din_cap = 0;
  const float* d_vecs = nullptr;      // [n_vecs][k] (caller's vectors) or X itself (model users: rows d_vrow)
  const int64_t* d_vrow = nullptr;    // vector v = row d_vrow[v] of d_vecs; NULL: row v
  const int32_t* d_vptr = nullptr;    // [nq + 1]
  const int64_t* d_rows = nullptr;    // [nq]: local row of the query's user (known items), -1 = none
  const int64_t* d_excl_ptr = nullptr;
  const int64_t* d_excl_idx = nullptr;
  // filter path
  float *d_tau = nullptr, *d_lb = nullptr;
  unsigned* d_count = nullptr;
  uint32_t* d_cand = nullptr;
  uint64_t* d_pairs = nullptr;
  uint8_t* d_outp = nullptr;    // the pass's results: [nq][how_many] pairs | counts | taus | overflow word
  void* d_img = nullptr;        // the pass's queries as split bf16 MFMA operands (topn_image_kernel)
  unsigned* d_wcount = nullptr; // hits per wave of the filter kernel, [n_waves] + one overflow word
  uint2* d_whits = nullptr;     // [n_waves][TOPN_WAVE_CAP] (item, query)
  size_t wh_cap = 0;
  size_t lb_cap = 0, cand_cap = 0, pairs_cap = 0, outp_cap = 0;
  // dense path
  float* d_scores = nullptr;
  uint32_t* d_sel = nullptr;
  TopnState* d_state = nullptr;
  unsigned* d_hist = nullptr;
  size_t scores_cap = 0, sel_cap = 0;
  // pinned staging, two slots: pass p's results are decoded while pass p + 1 runs
  uint8_t* h_stage[2] = {nullptr, nullptr};
  size_t stage_cap = 0;
  uint8_t* h_in[2] = {nullptr, nullptr};   // pinned input blocks (offsets, rows, vectors, exclusion lists)
  size_t in_cap[2] = {0, 0};
  hipEvent_t ev[2] = {nullptr, nullptr};
};

void topn_free(mals_handle h) {
  TopnWorkspace* w = static_cast<TopnWorkspace*>(h->tn_ws);
  if (!w) return;
  free_dev(w->d_in);
  free_dev(w->d_tau); free_dev(w->d_lb); free_dev(w->d_count); free_dev(w->d_cand);
  free_dev(w->d_pairs); free_dev(w->d_outp); free_dev(w->d_img); free_dev(w->d_wcount); free_dev(w->d_whits); free_dev(w->d_scores); free_dev(w->d_sel); free_dev(w->d_state); free_dev(w->d_hist);
  for (int s = 0; s < 2; ++s) {
    if (w->h_stage[s]) (void)hipHostFree(w->h_stage[s]);
    if (w->h_in[s]) (void)hipHostFree(w->h_in[s]);
    if (w->ev[s]) (void)hipEventDestroy(w->ev[s]);
  }
  delete w;
  h->tn_ws = nullptr;
}

template <typename P>
int topn_grow(mals_handle h, P*& p, size_t& cap, size_t want) {
  if (want <= cap) return MALS_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));  // a pass still in flight may use the old buffer
  free_dev(p);
  cap = 0;
  HIPCHK(h, hipMalloc(&p, sizeof(P) * want));
  cap = want;
  return MALS_OK;
}

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
};

struct TopnPass {
  int q0 = 0, nq = 0;
  int64_t v0 = 0, n_vecs = 0;
  bool have_rows = false, have_excl = false;
};

struct TopnCand {
  uint32_t key;
  int64_t idx;
};
void topn_emit(std::vector<TopnCand>& cand, int how_many, int64_t* item_idx_out, float* score_out, int32_t* n_out) {
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
}

// The pass's vectors, offsets, known-item rows and exclusion lists on the device: assembled in the pinned input block
// of `slot` and sent with ONE asynchronous copy (the host never waits for the pass that is still running).
int topn_upload_pass(mals_handle h, TopnWorkspace* w, const TopnRequest& rq, TopnPass& ps, int slot) {
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
  if (total > w->in_cap[slot]) {
    HIPCHK(h, hipStreamSynchronize(h->stream));  // an earlier pass may still be reading the old block
    if (w->h_in[slot]) (void)hipHostFree(w->h_in[slot]);
    w->h_in[slot] = nullptr;
    w->in_cap[slot] = 0;
    HIPCHK(h, hipHostMalloc(&w->h_in[slot], total + total / 2, hipHostMallocDefault));
    w->in_cap[slot] = total + total / 2;
  }
  if (int rc = topn_grow(h, w->d_in, w->din_cap, total + total / 2)) return rc;
  uint8_t* in = w->h_in[slot];
  int32_t* vptr = reinterpret_cast<int32_t*>(in + o_vptr);
  for (int q = 0; q <= ps.nq; ++q) vptr[q] = (rq.user_idx || !rq.vec_ptr) ? q : (int32_t)(rq.vec_ptr[ps.q0 + q] - ps.v0);
  ps.have_rows = ps.have_excl = false;
  w->d_vptr = reinterpret_cast<const int32_t*>(w->d_in + o_vptr);
  w->d_rows = reinterpret_cast<const int64_t*>(w->d_in + o_rows);
  w->d_excl_ptr = reinterpret_cast<const int64_t*>(w->d_in + o_eptr);
  w->d_excl_idx = reinterpret_cast<const int64_t*>(w->d_in + o_eidx);
  if (rq.user_idx) {
    std::memcpy(in + o_uidx, rq.user_idx + ps.q0, sizeof(int64_t) * (size_t)ps.nq);
    w->d_vecs = x.F;
    w->d_vrow = reinterpret_cast<const int64_t*>(w->d_in + o_uidx);
    if (rq.skip_known) {
      int64_t* rows = reinterpret_cast<int64_t*>(in + o_rows);
      for (int q = 0; q < ps.nq; ++q) rows[q] = rq.user_idx[ps.q0 + q] - x.row_offset;
      ps.have_rows = true;
    }
  } else {
    std::memcpy(in + o_vecs, rq.vectors + ps.v0 * k, sizeof(float) * (size_t)ps.n_vecs * (size_t)k);
    w->d_vecs = reinterpret_cast<const float*>(w->d_in + o_vecs);
    w->d_vrow = nullptr;
  }
  if (n_ex > 0) {
    int64_t* eptr = reinterpret_cast<int64_t*>(in + o_eptr);
    for (int q = 0; q <= ps.nq; ++q) eptr[q] = rq.excl_ptr[ps.q0 + q] - rq.excl_ptr[ps.q0];
    std::memcpy(in + o_eidx, rq.excl_idx + rq.excl_ptr[ps.q0], sizeof(int64_t) * (size_t)n_ex);
    ps.have_excl = true;
  }
  HIPCHK(h, hipMemcpyAsync(w->d_in, in, total, hipMemcpyHostToDevice, h->stream));
  return MALS_OK;
}

// ---- dense path: exact scores of every item ------------------------------------------------------------------------
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

int topn_pass_dense(mals_handle h, TopnWorkspace* w, const TopnRequest& rq, const TopnPass& ps) {
  SideState& y = h->side[MALS_SIDE_Y];
  SideState& x = h->side[MALS_SIDE_X];
  const int k = h->cfg.features, nq = ps.nq, how_many = rq.how_many;
  const int64_t n_items = y.n_total;
  const int cap_ties = 1024;
  const size_t per_q = 2 * ((size_t)how_many + cap_ties);
  if (int rc = topn_grow(h, w->d_scores, w->scores_cap, (size_t)TOPN_MAX_QUERIES * (size_t)n_items)) return rc;
  if (int rc = topn_grow(h, w->d_sel, w->sel_cap, (size_t)TOPN_MAX_QUERIES * per_q)) return rc;
  if (!w->d_state) HIPCHK(h, hipMalloc(&w->d_state, sizeof(TopnState) * TOPN_MAX_QUERIES));
  if (!w->d_hist) HIPCHK(h, hipMalloc(&w->d_hist, sizeof(unsigned) * 256 * TOPN_MAX_QUERIES));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_items + 63) / 64, (int64_t)h->n_cu * 8));
  hipLaunchKernelGGL(topn_exact_dense_kernel, dim3(grid), dim3(256), sizeof(float) * 64 * (size_t)(k + 1), h->stream, y.F, n_items, k, w->d_vecs,
                     w->d_vrow, w->d_vptr, nq, w->d_scores);
  if (ps.have_rows)
    hipLaunchKernelGGL(topn_mask_kernel, dim3(64, (unsigned)nq), dim3(256), 0, h->stream, h->known_ptr ? h->known_ptr : x.row_ptr,
                       h->known_ptr ? h->known_idx : x.col, w->d_rows, nq, 1, n_items, w->d_scores);
  if (ps.have_excl)
    hipLaunchKernelGGL(topn_exclude_kernel, dim3(64, (unsigned)nq), dim3(256), 0, h->stream, w->d_excl_ptr, w->d_excl_idx, nq, n_items, 1, n_items,
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
    const size_t qq = (size_t)(ps.q0 + q);
    topn_emit(cand, how_many, rq.item_idx_out + qq * how_many, rq.score_out + qq * how_many, rq.n_out ? rq.n_out + qq : nullptr);
  }
  return MALS_OK;
}

// ---- filter path ------------------------------------------------------------------------------------------------------
// query tiles per pass the LDS image of the split query operands allows (64 KB): S steps x NT tiles x 2 KB
int topn_max_tiles(int S) { return S == 1 ? 16 : S == 2 ? 15 : S == 3 ? 10 : 7; }

constexpr int TOPN_WAVE_CAP = 2048;  // hits a wave of the filter kernel can record (expected: a hundred)

template <int S, int MODE>
int topn_launch_filter(mals_handle h, int nt, const float* Y, int64_t n_items, int k, TopnWorkspace* w, int nq, int tile_stride,
                       int64_t n_out, int* n_waves_out) {
  const int64_t tiles = (n_items + 16 * (int64_t)tile_stride - 1) / (16 * (int64_t)tile_stride);
  // NT = query tiles per workgroup, GY = workgroup rows over the query tiles.  The sample (MODE 0) is a thousand item tiles:
  // it is spread over the query tiles as well, four at a time.
#define MALS_TOPN_LAUNCH(NT, GY)                                                                                                      \
  do {                                                                                                                                \
    const size_t lds = (size_t)NT * (S + 1) * 64 * 16;                                                                                \
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / lds)); /* workgroups resident per CU */          \
    if (MODE == 1 && NT > 8) per_cu = std::min(per_cu, 2); /* ~250 registers: two waves per SIMD */                                   \
    if (const char* e = std::getenv("MALS_TOPN_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e));                                   \
    const int64_t groups = MODE == 1 ? (tiles + 1) / 2 : tiles;                                                                       \
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((groups + 3) / 4, (int64_t)h->n_cu * per_cu / (GY)));      \
    if (MODE == 1) {                                                                                                                  \
      const size_t nw = (size_t)grid * 4;                                                                                             \
      if (nw * TOPN_WAVE_CAP > w->wh_cap) {                                                                                           \
        HIPCHK(h, hipStreamSynchronize(h->stream));                                                                                   \
        free_dev(w->d_whits);                                                                                                         \
        free_dev(w->d_wcount);                                                                                                        \
        w->wh_cap = 0;                                                                                                                \
        HIPCHK(h, hipMalloc(&w->d_whits, sizeof(uint2) * nw * TOPN_WAVE_CAP));                                                        \
        HIPCHK(h, hipMalloc(&w->d_wcount, sizeof(unsigned) * (nw + 1)));                                                              \
        w->wh_cap = nw * TOPN_WAVE_CAP;                                                                                               \
      }                                                                                                                               \
      *n_waves_out = (int)nw;                                                                                                         \
    }                                                                                                                                 \
    if (k == 32 * S)                                                                                                                  \
      hipLaunchKernelGGL((topn_filter_kernel<S, NT, MODE, true>), dim3(grid, (unsigned)(GY)), dim3(256), lds, h->stream, Y, n_items, k, \
                         static_cast<const bf16x8*>(w->d_img), nq, tile_stride, n_out, w->d_lb, w->d_tau, TOPN_WAVE_CAP, w->d_wcount, \
                         w->d_whits);                                                                                                 \
    else                                                                                                                              \
      hipLaunchKernelGGL((topn_filter_kernel<S, NT, MODE, false>), dim3(grid, (unsigned)(GY)), dim3(256), lds, h->stream, Y, n_items, k, \
                         static_cast<const bf16x8*>(w->d_img), nq, tile_stride, n_out, w->d_lb, w->d_tau, TOPN_WAVE_CAP, w->d_wcount, \
                         w->d_whits);                                                                                                 \
  } while (0)
  constexpr int NTMAX = S == 1 ? 16 : S == 2 ? 15 : S == 3 ? 10 : 7;
  if constexpr (MODE == 0) {
    if (nt <= 1) MALS_TOPN_LAUNCH(1, 1);
    else MALS_TOPN_LAUNCH(4, (nt + 3) / 4);
  } else {
    if (nt <= 1) MALS_TOPN_LAUNCH(1, 1);
    else if (nt <= 4) MALS_TOPN_LAUNCH(4, 1);
    else if (nt <= 8 && NTMAX > 8) MALS_TOPN_LAUNCH(8, 1);
    else MALS_TOPN_LAUNCH(NTMAX, 1);
  }
#undef MALS_TOPN_LAUNCH
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
template <int MODE>
int topn_launch_filter_S(mals_handle h, int S, int nt, const float* Y, int64_t n_items, int k, TopnWorkspace* w, int nq, int tile_stride,
                         int64_t n_out, int* n_waves_out) {
  switch (S) {
    case 1: return topn_launch_filter<1, MODE>(h, nt, Y, n_items, k, w, nq, tile_stride, n_out, n_waves_out);
    case 2: return topn_launch_filter<2, MODE>(h, nt, Y, n_items, k, w, nq, tile_stride, n_out, n_waves_out);
    case 3: return topn_launch_filter<3, MODE>(h, nt, Y, n_items, k, w, nq, tile_stride, n_out, n_waves_out);
    default: return topn_launch_filter<4, MODE>(h, nt, Y, n_items, k, w, nq, tile_stride, n_out, n_waves_out);
  }
}

struct TopnFilterPlan {
  int S, cap, cap_pad, tile_stride;
  int64_t n_sample;
  size_t stage_bytes;
};
TopnFilterPlan topn_plan(mals_handle h, int how_many) {
  TopnFilterPlan p;
  const int64_t n_items = h->side[MALS_SIDE_Y].n_total;
  p.S = (h->cfg.features + 31) / 32;
  p.cap = 48 * how_many + 2048;
  p.cap_pad = 1;
  while (p.cap_pad < p.cap) p.cap_pad <<= 1;
  const int64_t target = std::max<int64_t>(512 * (int64_t)how_many, 16384);  // sample items: expected candidates = how_many x stride
  p.tile_stride = (int)std::max<int64_t>(1, n_items / target);
  p.n_sample = ((n_items + 16 * (int64_t)p.tile_stride - 1) / (16 * (int64_t)p.tile_stride)) * 16;
  p.stage_bytes = (size_t)TOPN_FILTER_QUERIES * ((size_t)how_many * 8 + 8) + 16;  // pairs | counts | taus | overflow word
  return p;
}

// enqueue one pass; its results land in pinned slot `slot` behind event ev[slot]
int topn_pass_filter_enqueue(mals_handle h, TopnWorkspace* w, const TopnRequest& rq, const TopnPass& ps, const TopnFilterPlan& p, int slot) {
  SideState& y = h->side[MALS_SIDE_Y];
  SideState& x = h->side[MALS_SIDE_X];
  const int k = h->cfg.features, nq = ps.nq, how_many = rq.how_many;
  const int64_t n_items = y.n_total;
  const int nt = (nq + 15) / 16;
  if (!w->d_tau) {
    HIPCHK(h, hipMalloc(&w->d_tau, sizeof(float) * TOPN_FILTER_QUERIES));
    HIPCHK(h, hipMalloc(&w->d_count, sizeof(unsigned) * (TOPN_FILTER_QUERIES * TOPN_COUNT_STRIDE + 1)));  // padded counters, then the overflow word
    HIPCHK(h, hipMalloc(&w->d_img, (size_t)16 * 5 * 64 * 16));
  }
  unsigned* d_overflow = w->d_count + (size_t)TOPN_FILTER_QUERIES * TOPN_COUNT_STRIDE;
  if (int rc = topn_grow(h, w->d_lb, w->lb_cap, (size_t)TOPN_FILTER_QUERIES * (size_t)p.n_sample)) return rc;
  if (int rc = topn_grow(h, w->d_pairs, w->pairs_cap, (size_t)TOPN_FILTER_QUERIES * (size_t)p.cap)) return rc;
  if (int rc = topn_grow(h, w->d_cand, w->cand_cap, (size_t)TOPN_FILTER_QUERIES * (size_t)p.cap)) return rc;
  if (int rc = topn_grow(h, w->d_outp, w->outp_cap, p.stage_bytes)) return rc;
  const int64_t* d_rows = ps.have_rows ? w->d_rows : nullptr;
  const int64_t* k_ptr = h->known_ptr ? h->known_ptr : x.row_ptr;   // knownItemIDs if the caller installed them, else the rows of R
  const int32_t* k_idx = h->known_ptr ? h->known_idx : x.col;
  const int64_t* d_eptr = ps.have_excl ? w->d_excl_ptr : nullptr;
  const int64_t* d_eidx = ps.have_excl ? w->d_excl_idx : nullptr;
  // 0. the queries as matrix operands (every tile an instantiation may touch: padding queries never produce a hit)
  hipLaunchKernelGGL(topn_prepare_kernel, dim3(16), dim3(256), 0, h->stream, w->d_vecs, w->d_vrow, w->d_vptr, nq, k, p.S,
                     static_cast<bf16x8*>(w->d_img), w->d_count, d_overflow);
  // 1. sample: lower bounds of every tile_stride-th tile; 2. threshold (known items out of the sample first)
  int n_fw = 0;
  if (int rc = topn_launch_filter_S<0>(h, p.S, nt, y.F, n_items, k, w, nq, p.tile_stride, p.n_sample, &n_fw)) return rc;
  hipLaunchKernelGGL(topn_threshold_kernel, dim3((unsigned)nq), dim3(1024), 0, h->stream, w->d_lb, p.n_sample, how_many, k_ptr, k_idx, d_rows, d_eptr,
                     d_eidx, n_items, p.tile_stride, w->d_tau);
  // 3. filter, 4. exact scores of the hits (known items dropped), 5. the N best
  if (int rc = topn_launch_filter_S<1>(h, p.S, nt, y.F, n_items, k, w, nq, 1, n_items, &n_fw)) return rc;
  hipLaunchKernelGGL(topn_scatter_kernel, dim3((unsigned)n_fw), dim3(256), 0, h->stream, w->d_wcount, w->d_whits, TOPN_WAVE_CAP, n_fw, p.cap,
                     w->d_count, w->d_cand, d_overflow);
  hipLaunchKernelGGL(topn_rescore_kernel, dim3(4, (unsigned)nq), dim3(256), 0, h->stream, y.F, k, w->d_vecs, w->d_vrow, w->d_vptr, w->d_count, p.cap,
                     w->d_cand, k_ptr, k_idx, d_rows, d_eptr, d_eidx, w->d_pairs);
  uint8_t* o = w->d_outp;
  const size_t o_cnt = sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many, o_tau = o_cnt + sizeof(unsigned) * TOPN_FILTER_QUERIES,
               o_ovf = o_tau + sizeof(float) * TOPN_FILTER_QUERIES;
  hipLaunchKernelGGL(topn_final_kernel, dim3((unsigned)nq), dim3(256), sizeof(uint64_t) * (size_t)p.cap, h->stream, w->d_pairs, w->d_count, p.cap,
                     how_many, reinterpret_cast<uint64_t*>(o), reinterpret_cast<unsigned*>(o + o_cnt), w->d_tau, reinterpret_cast<float*>(o + o_tau),
                     d_overflow, reinterpret_cast<unsigned*>(o + o_ovf));
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(w->h_stage[slot], o, p.stage_bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(w->ev[slot], h->stream));
  return MALS_OK;
}

// decode a finished pass; *ok = false: a query overflowed its candidate buffer or had a thin sample (the dense path answers)
int topn_pass_filter_finish(mals_handle h, TopnWorkspace* w, const TopnRequest& rq, const TopnPass& ps, const TopnFilterPlan& p, int slot, bool* ok) {
  HIPCHK(h, hipEventSynchronize(w->ev[slot]));
  const int how_many = rq.how_many;
  const uint8_t* st = w->h_stage[slot];
  const uint64_t* outp = reinterpret_cast<const uint64_t*>(st);
  const unsigned* count = reinterpret_cast<const unsigned*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many);
  const float* tau = reinterpret_cast<const float*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many + sizeof(unsigned) * TOPN_FILTER_QUERIES);
  const unsigned wave_overflow = *reinterpret_cast<const unsigned*>(st + sizeof(uint64_t) * (size_t)TOPN_FILTER_QUERIES * (size_t)how_many +
                                                                    (sizeof(unsigned) + sizeof(float)) * TOPN_FILTER_QUERIES);
  *ok = wave_overflow == 0;
  for (int q = 0; q < ps.nq; ++q)
    if (count[q] > (unsigned)p.cap || !(tau[q] > -std::numeric_limits<float>::infinity())) *ok = false;
  if (!*ok) return MALS_OK;
  for (int q = 0; q < ps.nq; ++q) {
    const size_t qq = (size_t)(ps.q0 + q);
    int n = 0;
    for (int j = 0; j < how_many; ++j) {
      const uint64_t pr = outp[(size_t)q * how_many + j];
      if (pr != 0) {
        rq.item_idx_out[qq * how_many + j] = (int64_t)(0xffffffffu - (uint32_t)pr);
        rq.score_out[qq * how_many + j] = key_score((uint32_t)(pr >> 32));
        ++n;
      } else {
        rq.item_idx_out[qq * how_many + j] = -1;
        rq.score_out[qq * how_many + j] = -std::numeric_limits<float>::infinity();
      }
    }
    if (rq.n_out) rq.n_out[qq] = n;
  }
  return MALS_OK;
}

int topn_run(mals_handle h, const TopnRequest& rq) {
  if (!h->tn_ws) h->tn_ws = new TopnWorkspace();
  TopnWorkspace* w = static_cast<TopnWorkspace*>(h->tn_ws);
  const int64_t n_items = h->side[MALS_SIDE_Y].n_total;
  const bool dense_only = n_items < 131072 || n_items >= 0xffffffffll || rq.how_many > TOPN_FILTER_MAX_N ||
                          n_items / 16 < 64 * (int64_t)rq.how_many || std::getenv("MALS_TOPN_FULL");
  if (dense_only) {
    for (int q0 = 0; q0 < rq.n_queries; q0 += TOPN_MAX_QUERIES) {
      TopnPass ps;
      ps.q0 = q0;
      ps.nq = std::min(TOPN_MAX_QUERIES, rq.n_queries - q0);
      if (int rc = topn_upload_pass(h, w, rq, ps, 0)) return rc;
      if (int rc = topn_pass_dense(h, w, rq, ps)) return rc;
    }
    return MALS_OK;
  }
  const TopnFilterPlan p = topn_plan(h, rq.how_many);
  if (w->stage_cap < p.stage_bytes) {
    for (int s = 0; s < 2; ++s) {
      if (w->h_stage[s]) (void)hipHostFree(w->h_stage[s]);
      w->h_stage[s] = nullptr;
    }
    w->stage_cap = 0;
    for (int s = 0; s < 2; ++s) HIPCHK(h, hipHostMalloc(&w->h_stage[s], p.stage_bytes, hipHostMallocDefault));
    w->stage_cap = p.stage_bytes;
  }
  for (int s = 0; s < 2; ++s)
    if (!w->ev[s]) HIPCHK(h, hipEventCreateWithFlags(&w->ev[s], hipEventDisableTiming));
  int per_pass = 16 * topn_max_tiles(p.S);
  if (const char* e = std::getenv("MALS_TOPN_QUERIES_PER_PASS")) per_pass = std::max(16, std::min(per_pass, std::atoi(e) / 16 * 16));  // tuning override
  // pass i + 1 is enqueued before pass i is decoded: the device does not wait for the host between passes
  TopnPass prev;
  bool have_prev = false;
  int prev_slot = 0, slot = 0;
  auto finish_prev = [&]() -> int {
    if (!have_prev) return MALS_OK;
    have_prev = false;
    bool ok = true;
    if (int rc = topn_pass_filter_finish(h, w, rq, prev, p, prev_slot, &ok)) return rc;
    if (!ok) {  // rare: answer the pass exactly the slow way (its own input block is free again: the pass has finished)
      for (int q0 = prev.q0; q0 < prev.q0 + prev.nq; q0 += TOPN_MAX_QUERIES) {
        TopnPass ps;
        ps.q0 = q0;
        ps.nq = std::min(TOPN_MAX_QUERIES, prev.q0 + prev.nq - q0);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (int rc = topn_upload_pass(h, w, rq, ps, prev_slot)) return rc;
        if (int rc = topn_pass_dense(h, w, rq, ps)) return rc;
      }
    }
    return MALS_OK;
  };
  for (int q0 = 0; q0 < rq.n_queries; q0 += per_pass) {
    TopnPass ps;
    ps.q0 = q0;
    ps.nq = std::min(per_pass, rq.n_queries - q0);
    if (int rc = topn_upload_pass(h, w, rq, ps, slot)) return rc;
    if (int rc = topn_pass_filter_enqueue(h, w, rq, ps, p, slot)) return rc;
    const int this_slot = slot;
    slot = 1 - slot;
    if (int rc = finish_prev()) return rc;  // the pass enqueued one iteration ago; its slots are free for the next one
    prev = ps;
    prev_slot = this_slot;
    have_prev = true;
  }
  return finish_prev();
}
