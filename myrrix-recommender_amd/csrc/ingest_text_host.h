// ingest_text_host.h -- host side of the text half of the ingest path (mals_ingest_append_text / _read_file /
// _read_dir, include/myrrix_als.h).  Included by ingest_api.hip after mals_ingest_s and its helpers.
//
// The host does what java.io does for the reference -- list the directory (InputFilesReader.java:71-86), open and
// inflate files (FileLineIterator.java:92-102), hand the bytes on in blocks -- and keeps the two counters whose
// meaning is sequential (IFR:92-98: lines, badLines).  Splitting into lines, parsing every line and compacting the
// records happen on the device (ingest_text_kernels.h, text_parse.h).  A block boundary inside a line is healed by
// carrying the unterminated tail (device to device) in front of the next block.
#pragma once

#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

namespace {

constexpr size_t TEXT_PAD = 128;  // readable bytes behind the text (16-byte and 8-byte aligned window loads)

struct TextScratch {
  unsigned* block_counts = nullptr;  // line starts per LT_BLOCK_BYTES, then their exclusive scan
};

template <typename P>
int grow(mals_ingest g, P*& p, size_t& cap, size_t want, bool keep, size_t used = 0) {
  if (want <= cap) return MALS_OK;
  size_t ncap = std::max(want, cap + cap / 2);
  P* q = nullptr;
  ICHK(g, hipMalloc(&q, sizeof(P) * ncap));
  if (keep && used) {
    ICHK(g, hipMemcpyAsync(q, p, sizeof(P) * used, hipMemcpyDeviceToDevice, g->stream));
    ICHK(g, hipStreamSynchronize(g->stream));
  }
  dfree(p);
  p = q;
  cap = ncap;
  return MALS_OK;
}

int text_fail(mals_ingest g, int code, const std::string& msg) {
  g->text_failed = true;
  g->text_fail_code = code;
  g->text_fail_msg = msg;
  return fail(g, code, msg);
}

int ensure_record_capacity(mals_ingest g, int64_t extra) {
  if (g->n + extra <= g->cap) return MALS_OK;
  const int64_t cap = std::max<int64_t>(g->n + extra, g->cap + g->cap / 2);
  int64_t *u = nullptr, *it = nullptr;
  float* v = nullptr;
  ICHK(g, hipMalloc(&u, sizeof(int64_t) * (size_t)cap));
  ICHK(g, hipMalloc(&it, sizeof(int64_t) * (size_t)cap));
  ICHK(g, hipMalloc(&v, sizeof(float) * (size_t)cap));
  if (g->n) {
    ICHK(g, hipMemcpyAsync(u, g->d_user, sizeof(int64_t) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
    ICHK(g, hipMemcpyAsync(it, g->d_item, sizeof(int64_t) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
    ICHK(g, hipMemcpyAsync(v, g->d_value, sizeof(float) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
    ICHK(g, hipStreamSynchronize(g->stream));
  }
  dfree(g->d_user);
  dfree(g->d_item);
  dfree(g->d_value);
  g->d_user = u;
  g->d_item = it;
  g->d_value = v;
  g->cap = cap;
  return MALS_OK;
}

// One block: the bytes d_text[0, total) are on the device; lines are taken from [0, region) (region ends right after
// a line terminator, or at `total` for the last block of a file).
int process_text_region(mals_ingest g, size_t region) {
  if (region == 0) return MALS_OK;
  if (region >= 0xffffff00ull) return fail(g, MALS_INVALID_ARG, "text block too large");
  // HIP-event time of the kernels only: allocations (which cost milliseconds to seconds, depending on the box) sit
  // between the timed segments
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ICHK(g, hipEventCreate(&e0));
  ICHK(g, hipEventCreate(&e1));
  struct Ev {
    hipEvent_t &a, &b;
    ~Ev() {
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  } ev{e0, e1};
  auto seg_end = [&]() -> int {
    ICHK(g, hipEventRecord(e1, g->stream));
    ICHK(g, hipEventSynchronize(e1));
    float ms = 0.f;
    ICHK(g, hipEventElapsedTime(&ms, e0, e1));
    g->parse_ms += ms;
    return MALS_OK;
  };
  ICHK(g, hipEventRecord(e0, g->stream));
  // 1. line starts
  const int64_t n_blk = ((int64_t)region + LT_BLOCK_BYTES - 1) / LT_BLOCK_BYTES;
  const int64_t tiles = (n_blk + SC_TILE - 1) / SC_TILE;
  if (int rc = grow(g, g->t_block_counts, g->t_block_counts_cap, (size_t)n_blk, false)) return rc;
  if (int rc = grow(g, g->t_tile_sums, g->t_tile_sums_cap, (size_t)tiles + 2, false)) return rc;
  Scratch s;
  s.tile_sums = g->t_tile_sums;
  s.total = g->t_tile_sums + tiles;  // one word behind the tile sums
  hipLaunchKernelGGL(line_count_kernel, dim3((unsigned)n_blk), dim3(256), 0, g->stream, g->d_text, (int64_t)region, g->t_block_counts);
  ICHK(g, hipGetLastError());
  unsigned n_lines = 0;
  if (int rc = scan_u32(g, s, g->t_block_counts, g->t_block_counts, n_blk, &n_lines)) return rc;
  if (int rc = seg_end()) return rc;
  if (n_lines == 0) return fail(g, MALS_HIP_ERROR, "internal: a non-empty text region without a line");
  // sequential rule IFR:96-98: the line that follows the 101st bad line throws
  if (g->abort_armed) return text_fail(g, MALS_IO_ERROR, "Too many bad lines; aborting");
  // 2. per-line arrays
  const size_t L = n_lines;
  if (L > g->t_line_cap) {
    dfree(g->t_starts); dfree(g->t_status); dfree(g->t_user); dfree(g->t_item); dfree(g->t_value);
    dfree(g->t_flag); dfree(g->t_flag_scan); dfree(g->t_defer);
    g->t_line_cap = 0;
    const size_t cap = L + L / 4;
    ICHK(g, hipMalloc(&g->t_starts, sizeof(unsigned) * cap));
    ICHK(g, hipMalloc(&g->t_status, cap + 8));
    ICHK(g, hipMalloc(&g->t_user, sizeof(int64_t) * cap));
    ICHK(g, hipMalloc(&g->t_item, sizeof(int64_t) * cap));
    ICHK(g, hipMalloc(&g->t_value, sizeof(uint32_t) * cap));
    ICHK(g, hipMalloc(&g->t_flag, sizeof(unsigned) * cap));
    ICHK(g, hipMalloc(&g->t_defer, sizeof(unsigned) * cap));
    g->t_line_cap = cap;
  }
  {
    const int64_t lt = ((int64_t)L + SC_TILE - 1) / SC_TILE;
    if (int rc = grow(g, g->t_tile_sums, g->t_tile_sums_cap, (size_t)std::max(lt, tiles) + 2, false)) return rc;
    s.tile_sums = g->t_tile_sums;
    s.total = g->t_tile_sums + std::max(lt, tiles);
  }
  if (!g->t_counters) ICHK(g, hipMalloc(&g->t_counters, sizeof(TextCounters) + 2 * sizeof(unsigned)));
  ICHK(g, hipEventRecord(e0, g->stream));
  ICHK(g, hipMemsetAsync(g->t_counters, 0, sizeof(TextCounters) + 2 * sizeof(unsigned), g->stream));
  hipLaunchKernelGGL(line_starts_kernel, dim3((unsigned)n_blk), dim3(256), 0, g->stream, g->d_text, (int64_t)region, g->t_block_counts,
                     g->t_starts);
  const unsigned lgrid = (unsigned)((L + 255) / 256);
  const int first = g->lines == 0 ? 1 : 0;
  // 3. parse: the bulk, then whatever it handed on
  hipLaunchKernelGGL(parse_lines_kernel, dim3(lgrid), dim3(256), 0, g->stream, g->d_text, g->t_starts, (int64_t)L, (unsigned)region, first,
                     g->t_status, g->t_user, g->t_item, g->t_value, g->t_defer, g->t_counters);
  hipLaunchKernelGGL(parse_deferred_kernel, dim3((unsigned)std::min<size_t>((L + 63) / 64, 4096)), dim3(64), 0, g->stream, g->d_text,
                     g->t_starts, (int64_t)L, (unsigned)region, first, g->t_status, g->t_user, g->t_item, g->t_value, g->t_defer,
                     g->t_counters);
  const int64_t ct = ((int64_t)L + CT_TILE - 1) / CT_TILE;  // t_flag: records per tile, then their exclusive scan
  hipLaunchKernelGGL(line_summary_kernel, dim3((unsigned)ct), dim3(256), 0, g->stream, g->t_status, (int64_t)L, g->t_flag, g->t_counters);
  unsigned n_records = 0;
  if (int rc = scan_u32(g, s, g->t_flag, g->t_flag, ct, nullptr)) return rc;
  ICHK(g, hipMemcpyAsync(&n_records, s.total, sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
  ICHK(g, hipGetLastError());
  TextCounters c;
  ICHK(g, hipMemcpyAsync(&c, g->t_counters, sizeof(c), hipMemcpyDeviceToHost, g->stream));
  if (int rc = seg_end()) return rc;
  c.records = n_records;
  // 4. the sequential part of the contract
  if (c.fatal || g->bad_lines + (int64_t)c.bad > 100) {
    std::vector<uint8_t> st(L);
    ICHK(g, hipMemcpy(st.data(), g->t_status, L, hipMemcpyDeviceToHost));
    int64_t bad = g->bad_lines;
    for (size_t i = 0; i < L; ++i) {
      if (bad > 100) return text_fail(g, MALS_IO_ERROR, "Too many bad lines; aborting");
      const int k = st[i] & 15;
      if (k == text::ST_FATAL)
        return text_fail(g, MALS_INVALID_ARG,
                         "line " + std::to_string(g->lines + (int64_t)i + 1) +
                             ": a token that is a lone '\"' (the reference throws StringIndexOutOfBoundsException here)");
      if (k == text::ST_BAD) ++bad;
    }
  }
  if (g->finished) free_results(g);   // any accepted text (also lines that yield no record) makes the last finish stale
  g->bad_lines += c.bad;
  g->abort_armed = g->bad_lines > 100;
  g->lines += (int64_t)L;
  g->header_lines += c.header;
  g->skipped_lines += c.skipped;
  g->slow_lines += c.deferred;
  // 5. records, in file order
  if (c.records) {
    if (g->n + (int64_t)c.records >= MALS_INGEST_MAX_RECORDS) return text_fail(g, MALS_INVALID_ARG, "at most 2^36 records per ingest");
    if (int rc = ensure_record_capacity(g, c.records)) return rc;
    ICHK(g, hipEventRecord(e0, g->stream));
    hipLaunchKernelGGL(compact_records_kernel, dim3((unsigned)ct), dim3(256), 0, g->stream, g->t_status, g->t_flag, (int64_t)L, g->t_user,
                       g->t_item, g->t_value, g->d_user + g->n, g->d_item + g->n, g->d_value + g->n);
    ICHK(g, hipGetLastError());
    if (int rc = seg_end()) return rc;
    g->n += c.records;
  }
  if (c.user_tags || c.item_tags) {
    if (int rc = grow(g, g->d_tags[0], g->tag_cap[0], g->n_tags_raw[0] + c.user_tags, true, g->n_tags_raw[0])) return rc;
    if (int rc = grow(g, g->d_tags[1], g->tag_cap[1], g->n_tags_raw[1] + c.item_tags, true, g->n_tags_raw[1])) return rc;
    unsigned* cursors = reinterpret_cast<unsigned*>(g->t_counters + 1);
    hipLaunchKernelGGL(collect_tags_kernel, dim3(blocks_for((int64_t)L)), dim3(256), 0, g->stream, g->t_status, (int64_t)L, g->t_user, g->t_item,
                       g->d_tags[0] + g->n_tags_raw[0], g->d_tags[1] + g->n_tags_raw[1], cursors);
    ICHK(g, hipGetLastError());
    g->n_tags_raw[0] += c.user_tags;
    g->n_tags_raw[1] += c.item_tags;
  }
  g->text_bytes += (int64_t)region;
  return MALS_OK;
}

// position just behind the last line terminator of bytes[0, m) that can be recognised without looking past m
size_t last_terminator_end(const uint8_t* b, size_t m) {
  for (size_t q = m; q-- > 0;) {
    if (b[q] == '\n') return q + 1;
    if (b[q] == '\r' && q + 1 < m) return q + 1;
  }
  return 0;
}

int append_text_impl(mals_ingest g, const uint8_t* bytes, int64_t n_bytes, int mem_kind, bool eof) {
  const size_t block = g->text_block_bytes;
  int64_t off = 0;
  bool first_pass = true;
  while (off < n_bytes || (first_pass && eof && g->carry_len > 0)) {
    first_pass = false;
    const size_t m = (size_t)std::min<int64_t>(n_bytes - off, (int64_t)block);
    const bool last = eof && off + (int64_t)m == n_bytes;
    const size_t total = g->carry_len + m;
    if (total + TEXT_PAD > g->text_cap) {
      dfree(g->d_text);
      g->text_cap = 0;
      const size_t cap = ((std::max(total, block) + TEXT_PAD + 4095) / 4096) * 4096;
      ICHK(g, hipMalloc(&g->d_text, cap));
      g->text_cap = cap;
    }
    if (!g->t_ev[0]) {
      ICHK(g, hipEventCreate(&g->t_ev[0]));
      ICHK(g, hipEventCreate(&g->t_ev[1]));
    }
    ICHK(g, hipEventRecord(g->t_ev[0], g->stream));
    if (g->carry_len) ICHK(g, hipMemcpyAsync(g->d_text, g->d_carry, g->carry_len, hipMemcpyDeviceToDevice, g->stream));
    if (m)
      ICHK(g, hipMemcpyAsync(g->d_text + g->carry_len, bytes + off, m, mem_kind == MALS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                             g->stream));
    ICHK(g, hipMemsetAsync(g->d_text + total, 0, TEXT_PAD, g->stream));
    ICHK(g, hipEventRecord(g->t_ev[1], g->stream));
    // where the complete lines end
    size_t region;
    uint8_t last_byte = 0;
    if (last) {
      region = total;
    } else {
      size_t cut = 0;  // within the new bytes
      if (mem_kind == MALS_MEM_HOST) {
        cut = last_terminator_end(bytes + off, m);
        if (m) last_byte = bytes[off + (int64_t)m - 1];
      } else {
        std::vector<uint8_t> tail;
        size_t look = std::min<size_t>(m, 1 << 16);
        for (;;) {
          tail.resize(look);
          ICHK(g, hipMemcpy(tail.data(), bytes + off + (int64_t)(m - look), look, hipMemcpyDeviceToHost));
          const size_t c = last_terminator_end(tail.data(), look);
          if (c || look == m) {
            cut = c ? (m - look) + c : 0;
            break;
          }
          look = std::min<size_t>(m, look * 16);
        }
        if (m) last_byte = tail[look - 1];
      }
      if (cut) region = g->carry_len + cut;
      else region = (g->carry_ends_cr && m) ? g->carry_len : 0;  // the carried '\r' turned out to be a terminator of its own
    }
    if (int rc = process_text_region(g, region)) return rc;
    {
      float ms = 0.f;
      ICHK(g, hipEventSynchronize(g->t_ev[1]));
      ICHK(g, hipEventElapsedTime(&ms, g->t_ev[0], g->t_ev[1]));
      g->stage_ms += ms;  // bringing the block in front of the kernels: PCIe for host bytes, a device copy otherwise
    }
    // the tail waits for the next block
    const size_t rest = total - region;
    if (rest) {
      if (rest > g->carry_cap) {
        dfree(g->d_carry);
        g->carry_cap = 0;
        ICHK(g, hipMalloc(&g->d_carry, rest + rest / 2 + 4096));
        g->carry_cap = rest + rest / 2 + 4096;
      }
      ICHK(g, hipMemcpyAsync(g->d_carry, g->d_text + region, rest, hipMemcpyDeviceToDevice, g->stream));
      ICHK(g, hipStreamSynchronize(g->stream));
    } else {
      ICHK(g, hipStreamSynchronize(g->stream));
    }
    g->carry_len = rest;
    if (m) g->carry_ends_cr = rest > 0 && last_byte == '\r';
    else if (!rest) g->carry_ends_cr = false;
    off += (int64_t)m;
  }
  if (eof) {
    g->carry_len = 0;
    g->carry_ends_cr = false;
  }
  return MALS_OK;
}

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// PatternFilenameFilter(".+\\.csv(\\.(zip|gz))?")  (IFR:71)
bool is_input_file_name(const std::string& name) {
  std::string stem = name;
  if (ends_with(stem, ".zip")) stem.resize(stem.size() - 4);
  else if (ends_with(stem, ".gz")) stem.resize(stem.size() - 3);
  return ends_with(stem, ".csv") && stem.size() > 4;
}

}  // namespace

extern "C" {

int mals_ingest_set_option(mals_ingest g, int32_t option, int64_t value) {
  if (!g) return MALS_INVALID_ARG;
  switch (option) {
    case MALS_INGEST_OPT_KNOWN_ITEMS:
      g->want_known = value != 0;
      return MALS_OK;
    case MALS_INGEST_OPT_RESERVE_RECORDS: {
      if (value < 0 || value >= MALS_INGEST_MAX_RECORDS) return fail(g, MALS_INVALID_ARG, "at most 2^36 records per ingest");
      ICHK(g, hipSetDevice(g->device));
      return ensure_record_capacity(g, std::max<int64_t>(0, value - g->n));
    }
    case MALS_INGEST_OPT_PARTITION_RECORDS:
      if (value != 0 && (value < 64 || value > MALS_INGEST_ONE_SHOT_MAX)) return fail(g, MALS_INVALID_ARG, "partition records: 0 (default) or 64 .. 2^31 - 256");
      g->part_cap = value;
      return MALS_OK;
    case MALS_INGEST_OPT_TEXT_BLOCK_BYTES:
      if (value < 1 || value > (int64_t)1 << 31) return fail(g, MALS_INVALID_ARG, "text block: 1 byte .. 2 GiB");
      g->text_block_bytes = (size_t)value;
      return MALS_OK;
    default:
      return fail(g, MALS_INVALID_ARG, "unknown ingest option");
  }
}

int mals_ingest_append_text(mals_ingest g, const void* bytes, int64_t n_bytes, int mem_kind, int32_t end_of_file) {
  if (!g) return MALS_INVALID_ARG;
  if (g->text_failed) return fail(g, g->text_fail_code, g->text_fail_msg);
  if (n_bytes < 0 || (n_bytes > 0 && !bytes)) return fail(g, MALS_INVALID_ARG, "bad text block");
  if (mem_kind != MALS_MEM_HOST && mem_kind != MALS_MEM_DEVICE) return fail(g, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  ICHK(g, hipSetDevice(g->device));
  return append_text_impl(g, static_cast<const uint8_t*>(bytes), n_bytes, mem_kind, end_of_file != 0);
}

int mals_ingest_read_file(mals_ingest g, const char* path) {
  if (!g || !path) return MALS_INVALID_ARG;
  if (g->text_failed) return fail(g, g->text_fail_code, g->text_fail_msg);
  ICHK(g, hipSetDevice(g->device));
  const std::string p(path);
  if (ends_with(p, ".zip")) {
    // FileLineIterator.java:98-99 wraps the file in a ZipInputStream and never calls getNextEntry(): such a stream reads
    // as empty, so the reference takes no line from a .zip input.  The file must still exist (FileInputStream).
    FILE* f = fopen(path, "rb");
    if (!f) return text_fail(g, MALS_IO_ERROR, std::string("cannot open ") + path);
    fclose(f);
    return append_text_impl(g, nullptr, 0, MALS_MEM_HOST, true);
  }
  const size_t buf_bytes = std::min<size_t>(g->text_block_bytes, (size_t)64 << 20);
  if (buf_bytes > g->pinned_cap) {
    if (g->h_pinned) (void)hipHostFree(g->h_pinned);
    g->h_pinned = nullptr;
    g->pinned_cap = 0;
    ICHK(g, hipHostMalloc(&g->h_pinned, buf_bytes, hipHostMallocDefault));
    g->pinned_cap = buf_bytes;
  }
  uint8_t* buf = static_cast<uint8_t*>(g->h_pinned);
  if (ends_with(p, ".gz")) {
    gzFile f = gzopen(path, "rb");
    if (!f) return text_fail(g, MALS_IO_ERROR, std::string("cannot open ") + path);
    (void)gzbuffer(f, 1 << 20);
    bool any = false;
    for (;;) {
      const int got = gzread(f, buf, (unsigned)std::min<size_t>(buf_bytes, 1u << 30));
      if (got < 0 || (!any && gzdirect(f))) {  // GZIPInputStream: "Not in GZIP format" / corrupt member -> IOException
        gzclose(f);
        return text_fail(g, MALS_IO_ERROR, std::string("not a readable gzip file: ") + path);
      }
      any = true;
      const bool eof = got == 0 || gzeof(f);
      if (int rc = append_text_impl(g, buf, got, MALS_MEM_HOST, eof)) {
        gzclose(f);
        return rc;
      }
      if (eof) break;
    }
    gzclose(f);
    return MALS_OK;
  }
  FILE* f = fopen(path, "rb");
  if (!f) return text_fail(g, MALS_IO_ERROR, std::string("cannot open ") + path);
  for (;;) {
    const size_t got = fread(buf, 1, buf_bytes, f);
    if (ferror(f)) {
      fclose(f);
      return text_fail(g, MALS_IO_ERROR, std::string("read error: ") + path);
    }
    const bool eof = got < buf_bytes;  // a short read without an error is the end of a regular file
    if (int rc = append_text_impl(g, buf, (int64_t)got, MALS_MEM_HOST, eof)) {
      fclose(f);
      return rc;
    }
    if (eof) break;
  }
  fclose(f);
  return MALS_OK;
}

int mals_ingest_read_dir(mals_ingest g, const char* input_dir, int32_t* n_files_read) {
  if (!g || !input_dir) return MALS_INVALID_ARG;
  if (n_files_read) *n_files_read = 0;
  if (g->text_failed) return fail(g, g->text_fail_code, g->text_fail_msg);
  DIR* d = opendir(input_dir);
  if (!d) return MALS_OK;  // listFiles() == null: "No input files", not an error (IFR:80-83)
  struct Entry {
    std::string name;
    int64_t mtime_ms;
  };
  std::vector<Entry> files;
  while (dirent* e = readdir(d)) {
    const std::string name(e->d_name);
    if (name == "." || name == ".." || !is_input_file_name(name)) continue;
    struct stat st;
    const std::string full = std::string(input_dir) + "/" + name;
    if (stat(full.c_str(), &st) != 0) continue;
    files.push_back({name, (int64_t)st.st_mtim.tv_sec * 1000 + st.st_mtim.tv_nsec / 1000000});  // File.lastModified(): ms
  }
  closedir(d);
  // ByLastModifiedComparator (ascending); Arrays.sort is stable over listFiles()'s unspecified order: by name here
  std::sort(files.begin(), files.end(), [](const Entry& a, const Entry& b) { return a.name < b.name; });
  std::stable_sort(files.begin(), files.end(), [](const Entry& a, const Entry& b) { return a.mtime_ms < b.mtime_ms; });
  for (const Entry& f : files) {
    const std::string full = std::string(input_dir) + "/" + f.name;
    struct stat st;
    if (stat(full.c_str(), &st) != 0 || !S_ISREG(st.st_mode))
      return text_fail(g, MALS_IO_ERROR, full + " is not a readable file (FileInputStream would throw)");
    if (int rc = mals_ingest_read_file(g, full.c_str())) return rc;
    if (n_files_read) ++*n_files_read;
  }
  return MALS_OK;
}

int mals_ingest_text_info(mals_ingest g, mals_ingest_text_info_t* out) {
  if (!g || !out || out->struct_size < (int32_t)sizeof(mals_ingest_text_info_t)) return MALS_INVALID_ARG;
  out->lines = g->lines;
  out->bad_lines = g->bad_lines;
  out->header_lines = g->header_lines;
  out->skipped_lines = g->skipped_lines;
  out->full_parser_lines = g->slow_lines;
  out->text_bytes = g->text_bytes;
  out->records = g->n;
  out->parse_ms = g->parse_ms;
  out->stage_ms = g->stage_ms;
  out->n_item_tag_ids = g->finished ? g->n_tag_ids[0] : -1;
  out->n_user_tag_ids = g->finished ? g->n_tag_ids[1] : -1;
  out->n_known_items = (g->finished && g->known_ptr) ? g->n_known : -1;
  return MALS_OK;
}

int mals_ingest_get_tag_ids(mals_ingest g, int32_t which, int64_t* host_ids_out) {
  if (!g) return MALS_INVALID_ARG;
  if (which != MALS_ITEM_TAG_IDS && which != MALS_USER_TAG_IDS) return fail(g, MALS_INVALID_ARG, "which: MALS_ITEM_TAG_IDS or MALS_USER_TAG_IDS");
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  if (g->n_tag_ids[which] && !host_ids_out) return MALS_INVALID_ARG;
  ICHK(g, hipSetDevice(g->device));
  if (g->n_tag_ids[which])
    ICHK(g, hipMemcpy(host_ids_out, g->tag_ids[which], sizeof(int64_t) * (size_t)g->n_tag_ids[which], hipMemcpyDeviceToHost));
  return MALS_OK;
}

int mals_ingest_get_known_items(mals_ingest g, int64_t* host_ptr, int32_t* host_item_idx) {
  if (!g) return MALS_INVALID_ARG;
  if (!g->finished || !g->known_ptr) return fail(g, MALS_INVALID_ARG, "no known items: set MALS_INGEST_OPT_KNOWN_ITEMS before mals_ingest_finish");
  ICHK(g, hipSetDevice(g->device));
  if (host_ptr) ICHK(g, hipMemcpy(host_ptr, g->known_ptr, sizeof(int64_t) * (size_t)(g->n_users + 1), hipMemcpyDeviceToHost));
  if (host_item_idx && g->n_known)
    ICHK(g, hipMemcpy(host_item_idx, g->known_idx, sizeof(int32_t) * (size_t)g->n_known, hipMemcpyDeviceToHost));
  return MALS_OK;
}

int mals_ingest_device_known_items(mals_ingest g, const int64_t** ptr, const int32_t** item_idx, int64_t* n_known) {
  if (!g) return MALS_INVALID_ARG;
  if (!g->finished || !g->known_ptr) return fail(g, MALS_INVALID_ARG, "no known items: set MALS_INGEST_OPT_KNOWN_ITEMS before mals_ingest_finish");
  if (ptr) *ptr = g->known_ptr;
  if (item_idx) *item_idx = g->known_idx;
  if (n_known) *n_known = g->n_known;
  return MALS_OK;
}

}  // extern "C"
