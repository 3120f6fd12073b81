// lds_gather.hip -- stand-alone check + rate of the LDS-staged k = 128 kernels (csrc/lds_kernels.h) before they go into the
// library: (1) MODE 1 partial slots and MODE 0 solved rows against a double-precision restatement on the host for rows of
// awkward lengths (1, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 500, 4096); (2) the gather rate on a table of a given
// size with rows of one length, next to als_persistent_kernel_h<8, MODE, true> on the same lists.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../myrrix-recommender_amd/csrc -o lds_gather lds_gather.hip
#include "lds_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace mals;

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      std::exit(2);                                                           \
    }                                                                         \
  } while (0)

static void perm_image_host(const std::vector<double>& G, int k, std::vector<float>& img, bool perm) {
  constexpr int T = 8;
  img.assign((size_t)tri(T) * 256, 0.f);
  for (int i = 0; i < T; ++i)
    for (int j = i; j < T; ++j)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int rr = 4 * (lane >> 4) + r, cc = lane & 15;
          const int row = perm ? ldsk_feature(i, rr) : 16 * i + rr, col = perm ? ldsk_feature(j, cc) : 16 * j + cc;
          img[((size_t)tidx(T, i, j) * 64 + lane) * 4 + r] = (float)G[(size_t)row * k + col];
        }
}

// x = W^-1 b, W SPD, double Cholesky
static bool chol_solve(std::vector<double> W, std::vector<double> b, int k, std::vector<double>& x) {
  for (int j = 0; j < k; ++j) {
    double d = W[(size_t)j * k + j];
    for (int t = 0; t < j; ++t) d -= W[(size_t)j * k + t] * W[(size_t)j * k + t];
    if (d <= 0) return false;
    d = std::sqrt(d);
    W[(size_t)j * k + j] = d;
    for (int i = j + 1; i < k; ++i) {
      double s = W[(size_t)i * k + j];
      for (int t = 0; t < j; ++t) s -= W[(size_t)i * k + t] * W[(size_t)j * k + t];
      W[(size_t)i * k + j] = s / d;
    }
  }
  x = b;
  for (int i = 0; i < k; ++i) {
    double s = x[i];
    for (int t = 0; t < i; ++t) s -= W[(size_t)i * k + t] * x[t];
    x[i] = s / W[(size_t)i * k + i];
  }
  for (int i = k - 1; i >= 0; --i) {
    double s = x[i];
    for (int t = i + 1; t < k; ++t) s -= W[(size_t)t * k + i] * x[t];
    x[i] = s / W[(size_t)i * k + i];
  }
  return true;
}

__global__ void fill_table_kernel(float* M, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u;
    x ^= x >> 15;
    M[i] = ((float)(x & 0xffff) - 32768.f) * (0.3f / 32768.f);
  }
}
__global__ void fill_entries_kernel(int32_t* col, float* val, int64_t n, uint32_t n_rows) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    col[i] = (int32_t)(x % n_rows);
    val[i] = (float)(1 + (x >> 40) % 5);
  }
}

struct Problem {
  int k = 128;
  int64_t n_table = 0;
  std::vector<float> M;           // host copy (small problems only)
  std::vector<int64_t> row_ptr;
  std::vector<int32_t> col;
  std::vector<float> val;
  std::vector<WorkItem> items;
  float *dM = nullptr, *dval = nullptr, *dout = nullptr, *dscratch = nullptr, *dGf = nullptr, *dGp = nullptr, *dz = nullptr;
  int32_t* dcol = nullptr;
  int64_t* drp = nullptr;
  WorkItem* ditems = nullptr;
  unsigned long long* dbad = nullptr;
  int* dmarked = nullptr;
  SolveParams p;
};

static void setup_params(Problem& P, int64_t n_rows) {
  std::memset(&P.p, 0, sizeof(P.p));
  SolveParams& p = P.p;
  p.row_ptr = P.drp;
  p.col = P.dcol;
  p.val = P.dval;
  p.M = P.dM;
  p.Gf = P.dGf;
  p.Gperm = P.dGp;
  p.out = P.dout;
  p.items = P.ditems;
  p.rowsC = nullptr;
  p.scratch = P.dscratch;
  p.bad_row = P.dbad;
  p.suspect = P.dbad + 1;
  p.any_marked = P.dmarked;
  p.refine_flag = nullptr;
  p.refine_limit = 0.f;
  p.gramian_weight = 0.25f;
  p.n_work = n_rows;
  p.k = 128;
  p.ldm = 128;
  p.flags = 0;
  p.alpha = 1.f;
  p.lambda_alpha = 0.1f;
  p.sing_threshold = 1e-5f;
  p.zscale = P.dz;
  p.trace = nullptr;
  p.trace_start = 0;
}

int main(int argc, char** argv) {
  const int k = 128;
  constexpr int T = 8;
  const int64_t SLOT = (tri(T) * 4 + T) * 64;
  int fails = 0;
  // ---------------- correctness ----------------
  {
    Problem P;
    P.n_table = 5000;
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    P.M.resize((size_t)P.n_table * k);
    for (auto& v : P.M) v = nd(rng) * 0.1f;
    const int lens[] = {4096, 500, 129, 128, 127, 96, 65, 64, 63, 33, 32, 31, 1, 700, 200, 100, 70, 40, 5, 2};
    const int n_len = sizeof(lens) / sizeof(lens[0]);
    std::vector<int> row_len;
    for (int rep = 0; rep < 40; ++rep)
      for (int i = 0; i < n_len; ++i) row_len.push_back(lens[i]);
    std::sort(row_len.begin(), row_len.end(), [](int a, int b) { return a > b; });
    const int64_t n_rows = (int64_t)row_len.size();
    P.row_ptr.assign(1, 0);
    for (int64_t r = 0; r < n_rows; ++r) {
      for (int e = 0; e < row_len[r]; ++e) {
        P.col.push_back((int32_t)(rng() % P.n_table));
        float v = (float)(1 + rng() % 5);
        if (rng() % 7 == 0) v = -v;
        P.val.push_back(v);
      }
      P.row_ptr.push_back((int64_t)P.col.size());
    }
    P.items.resize((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; ++r) {
      P.items[r].begin = P.row_ptr[r];
      P.items[r].len = row_len[r];
      P.items[r].id = (int32_t)r;
    }
    // Gramian of the table
    std::vector<double> G((size_t)k * k, 0.0);
    for (int64_t i = 0; i < P.n_table; ++i)
      for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) G[(size_t)a * k + b] += (double)P.M[(size_t)i * k + a] * (double)P.M[(size_t)i * k + b];
    std::vector<float> img, imgp;
    perm_image_host(G, k, img, false);
    perm_image_host(G, k, imgp, true);
    CK(hipMalloc(&P.dM, sizeof(float) * P.M.size()));
    CK(hipMemcpy(P.dM, P.M.data(), sizeof(float) * P.M.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dcol, sizeof(int32_t) * P.col.size()));
    CK(hipMemcpy(P.dcol, P.col.data(), sizeof(int32_t) * P.col.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dval, sizeof(float) * P.val.size()));
    CK(hipMemcpy(P.dval, P.val.data(), sizeof(float) * P.val.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.drp, sizeof(int64_t) * P.row_ptr.size()));
    CK(hipMemcpy(P.drp, P.row_ptr.data(), sizeof(int64_t) * P.row_ptr.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.ditems, sizeof(WorkItem) * P.items.size()));
    CK(hipMemcpy(P.ditems, P.items.data(), sizeof(WorkItem) * P.items.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dout, sizeof(float) * (size_t)n_rows * k));
    CK(hipMalloc(&P.dscratch, sizeof(float) * (size_t)n_rows * SLOT));
    CK(hipMalloc(&P.dGf, sizeof(float) * img.size()));
    CK(hipMemcpy(P.dGf, img.data(), sizeof(float) * img.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dGp, sizeof(float) * imgp.size()));
    double* dG = nullptr;
    CK(hipMalloc(&dG, sizeof(double) * G.size()));
    CK(hipMemcpy(dG, G.data(), sizeof(double) * G.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gramian_perm_kernel, dim3((tri(T) * 256 + 255) / 256), dim3(256), 0, 0, dG, k, P.dGp);
    {
      std::vector<float> back(imgp.size());
      CK(hipMemcpy(back.data(), P.dGp, sizeof(float) * back.size(), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < back.size(); ++i)
        if (back[i] != imgp[i]) {
          std::printf("FAIL gramian_perm_kernel element %zu: %g vs %g\n", i, back[i], imgp[i]);
          ++fails;
          break;
        }
    }
    const float S = 256.f;   // max |z| = sqrt(5) * 256 * ~0.5 << 2^14
    const float z4[4] = {S, 1.f / (S * S), 1.f, 0.f};
    CK(hipMalloc(&P.dz, sizeof(z4)));
    CK(hipMemcpy(P.dz, z4, sizeof(z4), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dbad, 2 * sizeof(unsigned long long)));
    CK(hipMemset(P.dbad, 0xff, 2 * sizeof(unsigned long long)));
    CK(hipMalloc(&P.dmarked, sizeof(int)));
    CK(hipMemset(P.dmarked, 0, sizeof(int)));
    setup_params(P, n_rows);
    for (unsigned grid : {7u, 64u, 800u}) {
      // MODE 1: partial slots
      CK(hipMemset(P.dscratch, 0xff, sizeof(float) * (size_t)n_rows * SLOT));
      hipLaunchKernelGGL((als_lds_kernel_h<1>), dim3(grid), dim3(64), 0, 0, P.p);
      CK(hipDeviceSynchronize());
      std::vector<float> slots((size_t)n_rows * SLOT);
      CK(hipMemcpy(slots.data(), P.dscratch, sizeof(float) * slots.size(), hipMemcpyDeviceToHost));
      // MODE 0: solved rows
      CK(hipMemset(P.dout, 0xff, sizeof(float) * (size_t)n_rows * k));
      hipLaunchKernelGGL((als_lds_kernel_h<0>), dim3(grid), dim3(64), 0, 0, P.p);
      CK(hipDeviceSynchronize());
      std::vector<float> out((size_t)n_rows * k);
      CK(hipMemcpy(out.data(), P.dout, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
      double worst_w = 0, worst_b = 0, worst_x = 0;
      int64_t worst_x_row = -1;
      for (int64_t r = 0; r < n_rows; r += (grid == 800u ? 1 : 7)) {
        std::vector<double> W((size_t)k * k, 0.0), b(k, 0.0);
        for (int64_t e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) {
          const float* y = &P.M[(size_t)P.col[e] * k];
          const double w = std::fabs((double)P.val[e]);
          const double cb = P.val[e] > 0 ? 1.0 + w : 0.0;
          for (int a = 0; a < k; ++a) {
            b[a] += cb * y[a];
            for (int c2 = 0; c2 < k; ++c2) W[(size_t)a * k + c2] += w * (double)y[a] * (double)y[c2];
          }
        }
        // slots: permuted tile layout
        const float* s = &slots[(size_t)r * SLOT];
        double wn = 0, wd = 0, bn = 0, bd = 0;
        for (int i = 0; i < T; ++i)
          for (int j = i; j < T; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int rg = 0; rg < 4; ++rg) {
                const int row = ldsk_feature(i, 4 * (lane >> 4) + rg), cc = ldsk_feature(j, lane & 15);
                const double got = s[((size_t)tidx(T, i, j) * 64 + lane) * 4 + rg], want = W[(size_t)row * k + cc];
                wn += (got - want) * (got - want);
                wd += want * want;
              }
        for (int v = 0; v < T; ++v)
          for (int lane = 0; lane < 64; ++lane) {
            const double got = s[((size_t)tri(T) * 4 + v) * 64 + lane], want = b[ldsk_feature(v, lane & 15)];
            bn += (got - want) * (got - want);
            bd += want * want;
          }
        worst_w = std::max(worst_w, std::sqrt(wn / (wd + 1e-300)));
        worst_b = std::max(worst_b, bd > 0 ? std::sqrt(bn / bd) : std::sqrt(bn));
        // solve
        const double ridge = 0.1 * (double)(P.row_ptr[r + 1] - P.row_ptr[r]);
        for (int a = 0; a < k; ++a) {
          for (int c2 = 0; c2 < k; ++c2) W[(size_t)a * k + c2] += G[(size_t)a * k + c2];
          W[(size_t)a * k + a] += ridge;
        }
        std::vector<double> x;
        chol_solve(W, b, k, x);
        double xn = 0, xd = 0;
        for (int a = 0; a < k; ++a) {
          const double got = out[(size_t)r * k + a];
          xn += (got - x[a]) * (got - x[a]);
          xd += x[a] * x[a];
        }
        const double ex = xd > 0 ? std::sqrt(xn / xd) : std::sqrt(xn);
        if (!(ex <= worst_x)) {
          worst_x = ex;
          worst_x_row = r;
        }
      }
      const bool ok = worst_w < 2e-6 && worst_b < 2e-6 && worst_x < 2e-5;
      std::printf("%s grid %u: worst relative error  row Gramian %.3g  rhs %.3g  x %.3g (row %lld, len %d)\n", ok ? "ok  " : "FAIL", grid,
                  worst_w, worst_b, worst_x, (long long)worst_x_row, worst_x_row >= 0 ? row_len[worst_x_row] : -1);
      if (!ok) ++fails;
    }
    hipFree(P.dM); hipFree(P.dcol); hipFree(P.dval); hipFree(P.drp); hipFree(P.ditems); hipFree(P.dout); hipFree(P.dscratch);
    hipFree(P.dGf); hipFree(P.dGp); hipFree(dG);
    if (argc > 1 && std::strcmp(argv[1], "check") == 0) return fails ? 1 : 0;
  }
  // ---------------- rates ----------------
  {
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    struct Case { int64_t n_table; int len; int64_t n_rows; };
    const Case cases[] = {{10000000, 500, 1250000}, {100000000, 500, 1250000}, {10000000, 128, 3200000}, {100000000, 4096, 160000}, {10000000, 96, 3200000}};
    for (const Case& cs : cases) {
      Problem P;
      P.n_table = cs.n_table;
      const int64_t nnz = cs.n_rows * cs.len;
      if (hipMalloc(&P.dM, sizeof(float) * (size_t)cs.n_table * k) != hipSuccess) { std::printf("table alloc failed\n"); continue; }
      CK(hipMalloc(&P.dcol, sizeof(int32_t) * nnz));
      CK(hipMalloc(&P.dval, sizeof(float) * nnz));
      hipLaunchKernelGGL(fill_table_kernel, dim3(8192), dim3(256), 0, 0, P.dM, cs.n_table * k);
      hipLaunchKernelGGL(fill_entries_kernel, dim3(8192), dim3(256), 0, 0, P.dcol, P.dval, nnz, (uint32_t)cs.n_table);
      CK(hipDeviceSynchronize());
      P.items.resize((size_t)cs.n_rows);
      for (int64_t r = 0; r < cs.n_rows; ++r) {
        P.items[(size_t)r].begin = r * cs.len;
        P.items[(size_t)r].len = cs.len;
        P.items[(size_t)r].id = (int32_t)r;
      }
      CK(hipMalloc(&P.ditems, sizeof(WorkItem) * P.items.size()));
      CK(hipMemcpy(P.ditems, P.items.data(), sizeof(WorkItem) * P.items.size(), hipMemcpyHostToDevice));
      CK(hipMalloc(&P.dout, sizeof(float) * (size_t)cs.n_rows * k));
      const int64_t n_slots = std::min<int64_t>(cs.n_rows, 200000);
      CK(hipMalloc(&P.dscratch, sizeof(float) * (size_t)n_slots * SLOT));
      std::vector<double> G((size_t)k * k, 0.0);
      for (int a = 0; a < k; ++a) G[(size_t)a * k + a] = 0.01 * (double)cs.n_table;
      std::vector<float> img, imgp;
      perm_image_host(G, k, img, false);
      perm_image_host(G, k, imgp, true);
      CK(hipMalloc(&P.dGf, sizeof(float) * img.size()));
      CK(hipMemcpy(P.dGf, img.data(), sizeof(float) * img.size(), hipMemcpyHostToDevice));
      CK(hipMalloc(&P.dGp, sizeof(float) * imgp.size()));
      CK(hipMemcpy(P.dGp, imgp.data(), sizeof(float) * imgp.size(), hipMemcpyHostToDevice));
      const float S = 256.f;
      const float z4[4] = {S, 1.f / (S * S), 1.f, 0.f};
      CK(hipMalloc(&P.dz, sizeof(z4)));
      CK(hipMemcpy(P.dz, z4, sizeof(z4), hipMemcpyHostToDevice));
      CK(hipMalloc(&P.dbad, 2 * sizeof(unsigned long long)));
      CK(hipMemset(P.dbad, 0xff, 2 * sizeof(unsigned long long)));
      CK(hipMalloc(&P.dmarked, sizeof(int)));
      setup_params(P, cs.n_rows);
      const double bytes = (double)nnz * 520.0 + (double)cs.n_rows * 520.0;
      for (int mode = 0; mode < 2; ++mode) {
        SolveParams p = P.p;
        if (mode == 1) {   // segments: slot ids wrap (rate only)
          std::vector<WorkItem> it2 = P.items;
          for (auto& w : it2) w.id = (int32_t)(w.id % n_slots);
          CK(hipMemcpy(P.ditems, it2.data(), sizeof(WorkItem) * it2.size(), hipMemcpyHostToDevice));
        }
        for (int which = 0; which < 2; ++which) {
          for (unsigned per_cu : {8u, 32u, 128u}) {
            if (which == 0 && per_cu != 32u) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
              CK(hipEventRecord(ea));
              if (which == 0) {
                const unsigned grid = (unsigned)std::min<int64_t>((cs.n_rows + 3) / 4, 256 * 2 * 16);
                if (mode == 0) hipLaunchKernelGGL((als_persistent_kernel_h<8, 0, true>), dim3(grid), dim3(256), 0, 0, p);
                else hipLaunchKernelGGL((als_persistent_kernel_h<8, 1, true>), dim3(grid), dim3(256), 0, 0, p);
              } else {
                const unsigned grid = (unsigned)std::min<int64_t>(cs.n_rows, 256 * (int64_t)per_cu);
                if (mode == 0) hipLaunchKernelGGL((als_lds_kernel_h<0>), dim3(grid), dim3(64), 0, 0, p);
                else hipLaunchKernelGGL((als_lds_kernel_h<1>), dim3(grid), dim3(64), 0, 0, p);
              }
              CK(hipEventRecord(eb));
              CK(hipEventSynchronize(eb));
              float ms = 0.f;
              CK(hipEventElapsedTime(&ms, ea, eb));
              best = std::min(best, ms);
            }
            std::printf("table %6.2f GB  rows %8lld x %4d  MODE %d  %-22s  %8.2f ms  %.2f TB/s algorithmic\n", (double)cs.n_table * 512 / 1e9,
                        (long long)cs.n_rows, cs.len, mode, which == 0 ? "register gather (r3)" : (per_cu == 8 ? "LDS gather  8 wg/CU" : per_cu == 32 ? "LDS gather 32 wg/CU" : "LDS gather 128 wg/CU"),
                        best, bytes / (best * 1e-3) / 1e12);
            std::fflush(stdout);
          }
        }
      }
      hipFree(P.dM); hipFree(P.dcol); hipFree(P.dval); hipFree(P.ditems); hipFree(P.dout); hipFree(P.dscratch); hipFree(P.dGf); hipFree(P.dGp);
      hipFree(P.dz); hipFree(P.dbad); hipFree(P.dmarked);
    }
  }
  return fails ? 1 : 0;
}
