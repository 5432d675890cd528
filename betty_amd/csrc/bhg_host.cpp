// bhg_host.cpp — host-only part of libbhg: version, error string, layout construction.
// No GPU is touched here, so these entry points also work on a CPU-only box (the loader /
// symbol tests rely on that).
#include <stdarg.h>

#include "bhg_common.hpp"

namespace bhg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

}  // namespace bhg

using namespace bhg;

extern "C" {

int bhg_version(void) { return BHG_VERSION; }

const char* bhg_last_error(void) { return g_err; }

size_t bhg_workspace_bytes(int T) { return ws_bytes(T); }

int64_t bhg_layout_flat_size(const int64_t* numel, int T) {
  if (T < 0 || (T > 0 && !numel)) return -1;
  int64_t off = 0;
  for (int t = 0; t < T; ++t) {
    if (numel[t] < 0) return -1;
    off = round_up(off, BHG_FLAT_ALIGN) + numel[t];
  }
  return round_up(off, BHG_FLAT_ALIGN);
}

int64_t bhg_layout_num_chunks(const int64_t* numel, int T) {
  if (T < 0 || (T > 0 && !numel)) return -1;
  int64_t n = 0;
  for (int t = 0; t < T; ++t) {
    if (numel[t] < 0) return -1;
    n += (numel[t] + BHG_CHUNK_ELEMS - 1) / BHG_CHUNK_ELEMS;
  }
  return n;
}

int bhg_layout_build(const int64_t* numel, int T, int64_t* starts, bhg_chunk* chunks) {
  BHG_REQUIRE(T >= 0, "negative tensor count");
  BHG_REQUIRE(T == 0 || (numel && starts), "NULL array");
  int64_t off = 0, c = 0;
  for (int t = 0; t < T; ++t) {
    BHG_REQUIRE(numel[t] >= 0, "negative numel");
    off = round_up(off, BHG_FLAT_ALIGN);
    starts[t] = off;
    for (int64_t s = 0; s < numel[t]; s += BHG_CHUNK_ELEMS) {
      BHG_REQUIRE(chunks, "NULL chunk array");
      const int64_t len = (numel[t] - s) < BHG_CHUNK_ELEMS ? (numel[t] - s) : BHG_CHUNK_ELEMS;
      chunks[c].flat_off = off + s;
      chunks[c].src_off = s;
      chunks[c].tensor = t;
      chunks[c].len = (int32_t)len;
      ++c;
    }
    off += numel[t];
  }
  return BHG_OK;
}

}  // extern "C"
