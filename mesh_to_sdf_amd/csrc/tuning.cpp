// tuning.cpp — the knob table of tuning.h: defaults, names, the one place the environment is read.
#include "tuning.h"

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace m2s {

namespace {

enum Kind { K_INT, K_U32, K_F32, K_F64 };
struct Entry {
  const char* name;
  Kind kind;
  size_t offset;
};
#define M2S_KNOB(NAME, KIND, FIELD) {NAME, KIND, offsetof(Tuning, FIELD)}
const Entry TABLE[] = {
    M2S_KNOB("M2S_STATS", K_INT, stats),
    M2S_KNOB("M2S_HOST_TIMES", K_INT, host_times),
    M2S_KNOB("M2S_LANE_WALK", K_INT, lane_walk),
    M2S_KNOB("M2S_TREELETS", K_INT, treelets),
    M2S_KNOB("M2S_SORT_TILE", K_INT, sort_tile),
    M2S_KNOB("M2S_LEAF_MAX", K_U32, leaf_max),
    M2S_KNOB("M2S_LANE_RATIO", K_F64, lane_ratio),
    M2S_KNOB("M2S_LANE_RATIO_SPLIT", K_F64, lane_ratio_split),
    M2S_KNOB("M2S_QUERY_LANE_COEFF", K_F64, query_lane_coeff),
    M2S_KNOB("M2S_BRUTE_MAX", K_F64, brute_max),
    M2S_KNOB("M2S_CUT_MIN_PACKETS", K_U32, cut_min_packets),
    M2S_KNOB("M2S_QUERY_CUT_MIN", K_U32, query_cut_min),
    M2S_KNOB("M2S_QUERY_LAUNCH_TIGHT", K_INT, query_launch_tight),
    M2S_KNOB("M2S_GROUP", K_INT, group),
    M2S_KNOB("M2S_GROUP_TARGET_WAVES", K_U32, group_target_waves),
    M2S_KNOB("M2S_GROUP_MIN_RATIO", K_F64, group_min_ratio),
    M2S_KNOB("M2S_CUT_NEAR", K_F32, cut_near),
    M2S_KNOB("M2S_CUT_FAR", K_F32, cut_far),
    M2S_KNOB("M2S_CUT_COARSE", K_INT, cut_coarse),
    M2S_KNOB("M2S_CUT_COARSE_MIN_WAVES", K_U32, cut_coarse_min_waves),
    M2S_KNOB("M2S_CUT_COARSE_CAP", K_U32, cut_coarse_cap),
    M2S_KNOB("M2S_CUT_WAVE_CAP", K_U32, cut_wave_cap),
    M2S_KNOB("M2S_SPLIT", K_INT, split),
    M2S_KNOB("M2S_SPLIT_BUDGET", K_U32, split_budget),
    M2S_KNOB("M2S_SPLIT_PATIENCE", K_F64, split_patience),
    M2S_KNOB("M2S_SPLIT_MIN_RECORDS", K_U32, split_min_records),
    M2S_KNOB("M2S_SPLIT_MAX_RECORDS", K_U32, split_max_records),
    M2S_KNOB("M2S_SPLIT_ROUNDS", K_U32, split_rounds),
    M2S_KNOB("M2S_SPLIT_REPORT", K_INT, split_report),
    M2S_KNOB("M2S_DEFER", K_INT, defer),
    M2S_KNOB("M2S_HOST_PIECE_MB", K_U32, host_piece_mb),
    M2S_KNOB("M2S_PUSH_PIECES", K_U32, push_pieces),
    M2S_KNOB("M2S_PUSH_BLOCKS", K_U32, push_blocks),
};
#undef M2S_KNOB

Tuning g_tuning;
std::once_flag g_once;
std::mutex g_set_mu;

bool assign(Tuning& t, const Entry& e, const char* text) {
  char* end = nullptr;
  char* field = reinterpret_cast<char*>(&t) + e.offset;
  switch (e.kind) {
    // the whole text must be one number in the field's range ("12abc", 2^32 for a 32-bit field: refused, not wrapped)
    case K_INT: { const long v = strtol(text, &end, 10); if (end == text || *end || v < INT32_MIN || v > INT32_MAX) return false; *reinterpret_cast<int*>(field) = (int)v; break; }
    case K_U32: { const long long v = strtoll(text, &end, 10); if (end == text || *end || v < 0 || v > (long long)UINT32_MAX) return false; *reinterpret_cast<uint32_t*>(field) = (uint32_t)v; break; }
    case K_F32: { const double v = strtod(text, &end); if (end == text || *end) return false; *reinterpret_cast<float*>(field) = (float)v; break; }
    case K_F64: { const double v = strtod(text, &end); if (end == text || *end) return false; *reinterpret_cast<double*>(field) = v; break; }
  }
  return true;
}

void copy_field(Tuning& dst, const Tuning& src, const Entry& e) {
  const size_t bytes = e.kind == K_F64 ? 8 : 4;
  memcpy(reinterpret_cast<char*>(&dst) + e.offset, reinterpret_cast<const char*>(&src) + e.offset, bytes);
}

void sanitise(Tuning& t) {
  if (t.split_rounds < 1) t.split_rounds = 1;
  if (t.split_rounds > 6) t.split_rounds = 6;
  if (t.push_pieces < 1) t.push_pieces = 1;
  if (t.host_piece_mb < 1) t.host_piece_mb = 1;
  if (t.leaf_max > 16) t.leaf_max = 16;   // the tree's limit (bvh.hip): a larger wish would differ from every resident tree's size for ever
}

void load_from_environment() {
  for (const Entry& e : TABLE) {
    const char* v = getenv(e.name);   // the library's only look at the environment for its knobs
    if (v && *v && !assign(g_tuning, e, v)) fprintf(stderr, "[m2s] %s=%s ignored: not a number\n", e.name, v);
  }
  sanitise(g_tuning);
}

}  // namespace

const Tuning& tuning() {
  std::call_once(g_once, load_from_environment);
  return g_tuning;
}

int tuning_set(const char* name, const char* value) {
  (void)tuning();
  if (!name) return -1;
  std::lock_guard<std::mutex> lk(g_set_mu);
  for (const Entry& e : TABLE) {
    if (strcmp(e.name, name) != 0) continue;
    if (value == nullptr || *value == 0) {
      const Tuning defaults;
      copy_field(g_tuning, defaults, e);
    } else if (!assign(g_tuning, e, value)) {
      return -1;
    }
    sanitise(g_tuning);
    return 0;
  }
  return -1;
}

int tuning_describe(char* buf, int cap) {
  const Tuning& t = tuning();
  int need = 0, copied = 0;
  for (const Entry& e : TABLE) {
    char line[96];
    const char* field = reinterpret_cast<const char*>(&t) + e.offset;
    int n = 0;
    switch (e.kind) {
      case K_INT: n = snprintf(line, sizeof(line), "%s=%d\n", e.name, *reinterpret_cast<const int*>(field)); break;
      case K_U32: n = snprintf(line, sizeof(line), "%s=%u\n", e.name, *reinterpret_cast<const uint32_t*>(field)); break;
      case K_F32: n = snprintf(line, sizeof(line), "%s=%.9g\n", e.name, (double)*reinterpret_cast<const float*>(field)); break;
      case K_F64: n = snprintf(line, sizeof(line), "%s=%.17g\n", e.name, *reinterpret_cast<const double*>(field)); break;
    }
    if (buf && copied == need && need + n < cap) { memcpy(buf + need, line, (size_t)n); copied += n; }   // whole lines, and none after the first that did not fit
    need += n;
  }
  if (buf && cap > 0) buf[copied] = 0;
  return need;
}

}  // namespace m2s
