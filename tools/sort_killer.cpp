// sort_killer -- search for key arrays (keys 1..15) that drive libstdc++'s std::sort, as MaximizeCell / UpperBound call it
// (ref: src/protocolStack/mac/packet-scheduler/downlink-transport-scheduler.cpp:223-246, 351-376; libstdc++ 11 bits/stl_algo.h
// __introsort_loop: depth limit 2*floor(log2 n), __partial_sort when it runs out), into its HEAP-SORT fallback on a range longer
// than 16 -- the one branch of the sort that random CQI grids never take (0 of 80 000 sorts in the survey's probe).
//
// Hill climb over the product's own emulation (radiosaber_amd/csrc/rs_sort_emul.h, the code the gfx950 kernel runs): a step changes
// up to three keys, the score rewards long sub-ranges on deep recursion levels.  Every array found is checked against the REAL
// std::sort before it is printed (same permutation), and classified by the device site it reaches in the level-synchronous form of
// the loop (radiosaber_amd/csrc/rs_sort_device.h):
//     wg    a sub-range longer than 64 (or more than kFinishMax * waves sub-ranges) is alive on every level down to depth 0:
//           the workgroup-level fallback (register form, introsort_levels_reg; the LDS form always falls back at this level)
//     wave  the last levels were handed to single waves before the depth ran out: the fallback inside finish_subranges_on_wave
//
//   g++ -O2 -std=c++17 -o /tmp/sort_killer tools/sort_killer.cpp
//   /tmp/sort_killer <n> <mode: any|wg|wave> <seed> [max_steps] [waves=8] [ept=0: ceil(n / (64 waves))] [run=0: start from a uniform array]
// prints one line "<n> <mode> <site at that workgroup size> <heap calls> <steps>" and one line with the n keys, or "none".
//   /tmp/sort_killer check [waves] [ept] < keys       classifies an array (tests/test_sort_killers.py)
// tools/make_sort_killers.py drives it and writes tests/golden/sort_killers.npz.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "../radiosaber_amd/csrc/rs_sort_emul.h"

struct Trace {
  int depth_limit = 0, heap_calls = 0, heap_longest = 0;
  std::vector<int> n_alive, n_long, max_len;  // per recursion level: sub-ranges longer than 16 / longer than 64 / the longest
};

static void walk(std::vector<uint32_t>& v, int first, int last, int depth, int level, Trace& t) {
  while (last - first > 16) {
    if ((int)t.n_alive.size() <= level) { t.n_alive.resize(level + 1, 0); t.n_long.resize(level + 1, 0); t.max_len.resize(level + 1, 0); }
    t.n_alive[level]++;
    if (last - first > 64) t.n_long[level]++;
    t.max_len[level] = std::max(t.max_len[level], last - first);
    if (depth == 0) {
      t.heap_calls++;
      t.heap_longest = std::max(t.heap_longest, last - first);
      rs_sort::heap_sort(v, first, last);
      return;
    }
    --depth;
    const int mid = first + (last - first) / 2;
    rs_sort::median_to_first(v, first, first + 1, mid, last - 1);
    const int cut = rs_sort::unguarded_partition(v, first + 1, last, first);
    walk(v, cut, last, depth, level + 1, t);
    last = cut;
    ++level;
  }
}

static Trace trace_of(const std::vector<int>& keys, std::vector<uint32_t>* out = nullptr) {
  const int n = (int)keys.size();
  std::vector<uint32_t> v(n);
  for (int i = 0; i < n; ++i) v[i] = ((uint32_t)keys[i] << 16) | (uint32_t)i;
  Trace t;
  t.depth_limit = 2 * rs_sort::floor_log2(n > 1 ? n : 1);
  walk(v, 0, n, t.depth_limit, 0, t);
  if (out) {
    std::stable_sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });  // __final_insertion_sort
    *out = v;
  }
  return t;
}

// which fallback site the register form of the device loop reaches (rs_sort_device.h, introsort_levels_reg)
static const char* device_site(const Trace& t, int waves, int ept) {
  const int finish_max = ept == 1 ? 2 : 4;
  for (int level = 0; level < (int)t.n_alive.size(); ++level) {
    const int depth = t.depth_limit - level;
    const long alive = t.n_alive[level] + ((long)t.n_long[level] << 16);
    if (alive == 0) break;
    if (depth != 0 && alive <= finish_max * waves && (level > 0 || ept > 1 || alive == 1)) return t.heap_calls ? "wave" : "-";
    if (depth == 0) return "wg";
  }
  return "-";
}

static long score_of(const Trace& t, const std::string& mode, int waves, int ept) {
  // "any": the deepest level that still holds a sub-range longer than 16, then that sub-range's length -- wide plateaus, which the
  // climb crosses by neutral moves.  "wg": the same with "longer than 64".  "wave": the hand-off to single waves must come before
  // the depth runs out, so nothing longer than 64 may live on the last levels.
  const int levels = (int)t.max_len.size();
  long s = 0;
  if (mode == "wg") {
    int deep = 0;
    while (deep < levels && t.n_long[deep] > 0) ++deep;
    s = 4096L * deep + (deep < levels ? std::min(t.max_len[deep], 64) : 0);
  } else {
    s = 4096L * levels + std::min(levels ? t.max_len[levels - 1] : 0, 48);
    if (mode == "wave")
      for (int level = std::max(1, t.depth_limit - 4); level < levels; ++level) s -= 3L * 4096L * t.n_long[level];
  }
  (void)waves; (void)ept;
  return s;
}

static bool reached(const Trace& t, const std::string& mode, int waves, int ept) {
  if (t.heap_calls == 0) return false;
  if (mode == "any") return true;
  return mode == device_site(t, waves, ept);
}

static bool same_as_std_sort(const std::vector<int>& keys) {
  typedef std::pair<std::pair<int, int>, double> elem_t;  // the reference's coord_cqi_t
  const int n = (int)keys.size();
  std::vector<elem_t> ref(n);
  for (int i = 0; i < n; ++i) ref[i] = elem_t(std::make_pair(i, 0), 0.125 * keys[i]);
  std::sort(ref.begin(), ref.end(), [](elem_t a, elem_t b) { return a.second > b.second; });
  std::vector<uint32_t> got;
  trace_of(keys, &got);
  for (int i = 0; i < n; ++i)
    if ((int)(got[i] & 0xFFFF) != ref[i].first.first) return false;
  return true;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "check") == 0) {
    /* check [waves] [ept] < keys: "<site> <heap calls> <longest heap-sorted range> <1: the emulation's permutation is std::sort's>" */
    std::vector<int> keys;
    for (int k; scanf("%d", &k) == 1;) keys.push_back(k);
    const int n = (int)keys.size();
    const int waves = argc > 2 ? atoi(argv[2]) : 8;
    const int ept = (argc > 3 && atoi(argv[3]) > 0) ? atoi(argv[3]) : (n + 64 * waves - 1) / (64 * waves);
    const Trace t = trace_of(keys);
    printf("%s %d %d %d\n", device_site(t, waves, ept), t.heap_calls, t.heap_longest, same_as_std_sort(keys) ? 1 : 0);
    return 0;
  }
  if (argc < 4) {
    fprintf(stderr, "usage: %s <n> <any|wg|wave> <seed> [max_steps] [waves] [ept] [runs] | %s check [waves] [ept] < keys\n", argv[0], argv[0]);
    return 2;
  }
  const int n = atoi(argv[1]);
  const std::string mode = argv[2];
  const unsigned seed = (unsigned)strtoul(argv[3], nullptr, 10);
  const long max_steps = argc > 4 ? atol(argv[4]) : 20000000L;
  const int waves = argc > 5 ? atoi(argv[5]) : 8;
  const int ept = (argc > 6 && atoi(argv[6]) > 0) ? atoi(argv[6]) : (n + 64 * waves - 1) / (64 * waves);
  std::mt19937 rng(seed);
  std::vector<int> keys(n, 1 + (int)(rng() % 15)), trial;
  if (argc > 7 && strcmp(argv[7], "0") != 0) {
    /* a structured start: a pivot equal to the range's smallest key, met by a run of that key at the range's right end, peels about
     * half of the run per level and leaves the same picture behind -- log2(run) levels per key value.  argv[7] = the run lengths of
     * keys 1, 2, ... from the right end ("62" = that length for every key while the array's right two thirds last) */
    std::fill(keys.begin(), keys.end(), 15);
    keys[0] = keys[1] = 1;
    int pos = n;
    if (strchr(argv[7], ',')) {
      int k = 1;
      for (char* tok = strtok(argv[7], ","); tok && k <= 14; tok = strtok(nullptr, ","), ++k)
        for (int j = atoi(tok); j > 0 && pos > 2; --j) keys[--pos] = k;
    } else {
      const int run = atoi(argv[7]);
      for (int k = 1; k <= 14 && pos - run > n / 3; ++k)
        for (int j = 0; j < run; ++j) keys[--pos] = k;
    }
  }
  Trace t = trace_of(keys);
  long best = score_of(t, mode, waves, ept);
  long step = 0;
  if (getenv("SORT_KILLER_VERBOSE")) {
    fprintf(stderr, "start: %d levels of %d, heap calls %d, site %s; longest per level:", (int)t.max_len.size(), t.depth_limit, t.heap_calls, device_site(t, waves, ept));
    for (size_t l = 0; l < t.max_len.size(); ++l) fprintf(stderr, " %d(%d)", t.max_len[l], t.n_alive[l]);
    fprintf(stderr, "\n");
  }
  for (; step < max_steps && !reached(t, mode, waves, ept); ++step) {
    trial = keys;
    const int changes = 1 + (int)(rng() % 3);
    for (int c = 0; c < changes; ++c) trial[rng() % n] = 1 + (int)(rng() % 15);
    const Trace tt = trace_of(trial);
    const long s = score_of(tt, mode, waves, ept);
    if (s >= best) {
      best = s;
      keys.swap(trial);
      t = tt;
    }
  }
  if (!reached(t, mode, waves, ept)) {
    printf("none\n");
    return 1;
  }
  if (!same_as_std_sort(keys)) {
    fprintf(stderr, "emulation and std::sort disagree on the array found\n");
    return 3;
  }
  printf("%d %s %s %d %ld\n", n, mode.c_str(), device_site(t, waves, ept), t.heap_calls, step);
  for (int i = 0; i < n; ++i) printf("%d%c", keys[i], i + 1 < n ? ' ' : '\n');
  return 0;
}
