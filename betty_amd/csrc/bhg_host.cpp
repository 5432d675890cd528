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

#ifdef BHG_AB
int g_dbg[DBG_COUNT] = {
#define BHG_DBG_INIT(n) kDbgUnset,
    BHG_DBG_KEYS(BHG_DBG_INIT)
#undef BHG_DBG_INIT
};
#endif
static const char* const kDbgNames[DBG_COUNT] = {
#define BHG_DBG_NAME(n) #n,
    BHG_DBG_KEYS(BHG_DBG_NAME)
#undef BHG_DBG_NAME
};
static int dbg_find(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < DBG_COUNT; ++i)
    if (strcmp(kDbgNames[i], key) == 0) return i;
  return -1;
}

}  // namespace bhg

using namespace bhg;

extern "C" {

int bhg_version(void) { return BHG_VERSION; }

const char* bhg_last_error(void) { return g_err; }

size_t bhg_workspace_bytes(int T) { return ws_bytes(T); }

#ifdef BHG_AB
int bhg_debug_set(const char* key, int value) {
  const int i = dbg_find(key);
  BHG_REQUIRE(i >= 0, "unknown debug key");
  BHG_REQUIRE(value != kDbgUnset, "INT32_MIN is the 'unset' marker");
  g_dbg[i] = value;
  return BHG_OK;
}
int bhg_debug_unset(const char* key) {
  const int i = dbg_find(key);
  BHG_REQUIRE(i >= 0, "unknown debug key");
  g_dbg[i] = kDbgUnset;
  return BHG_OK;
}
void bhg_debug_reset(void) {
  for (int i = 0; i < DBG_COUNT; ++i) g_dbg[i] = kDbgUnset;
}
int bhg_debug_key_count(void) { return DBG_COUNT; }
const char* bhg_debug_key_name(int i) { return (i >= 0 && i < DBG_COUNT) ? kDbgNames[i] : nullptr; }
int bhg_is_ab_build(void) { return 1; }
#else
// the product carries no measurement arm (bhg_common.hpp): the calls exist so that one binding serves both builds, and say so
int bhg_debug_set(const char* key, int) {
  (void)dbg_find(key);
  BHG_REQUIRE(false, "this is the product build of libbhg: measurement / test arms live in libbhg_ab.so (make -C betty_amd/csrc ab)");
}
int bhg_debug_unset(const char* key) {
  (void)key;
  return BHG_OK;   // (nothing is ever set)
}
void bhg_debug_reset(void) {}
int bhg_debug_key_count(void) { return 0; }
const char* bhg_debug_key_name(int) { return nullptr; }
int bhg_is_ab_build(void) { return 0; }
#endif

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

// Strided block copy on the device (one hipMemcpy2DAsync): what betty_amd/hypergradient/_mlp_hip.py uses to keep the zero-padded twin of
// a network whose widths are not multiples of 32 (and to un-pad results) — see include/bhg.h.
int bhg_copy2d(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int64_t cols, void* stream) {
  BHG_REQUIRE(rows >= 0 && cols >= 0 && ldd >= cols && lds >= cols, "bad block shape");
  if (rows == 0 || cols == 0) return BHG_OK;
  BHG_REQUIRE(dst && src, "NULL pointer");
  BHG_HIP_CHECK(hipMemcpy2DAsync(dst, sizeof(float) * (size_t)ldd, src, sizeof(float) * (size_t)lds, sizeof(float) * (size_t)cols,
                                 (size_t)rows, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return BHG_OK;
}

}  // extern "C"
