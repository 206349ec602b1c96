/*
 * rs_sort_emul.h -- exact emulation of the element order libstdc++'s std::sort leaves behind for
 * MaximizeCell's call
 *     std::sort(v.begin(), v.end(), [](coord_cqi_t a, coord_cqi_t b){ return a.second > b.second; });
 * (reference: src/protocolStack/mac/packet-scheduler/downlink-transport-scheduler.cpp:354-361).
 *
 * std::sort is not stable and the keys take only 16 values, so MaximizeCell's greedy scan depends on
 * the exact permutation.  The algorithm restated here is the published one of libstdc++ 11
 * (bits/stl_algo.h std::__sort: __introsort_loop with _S_threshold 16 and depth limit 2*floor(log2 n),
 * __move_median_to_first, __unguarded_partition, __partial_sort heap fallback, then
 * __final_insertion_sort; bits/stl_heap.h __adjust_heap/__push_heap/__pop_heap), written from its
 * specification, on packed 32-bit elements whose sort key is the top 16 bits.
 *
 * The final insertion sort is a stable sort of the array the introsort loop leaves, and a stable
 * order is unique, so it is done as a 16-bucket stable counting sort (descending key).
 *
 * Host+device: the kernel uses median-of-3 / before() / heap_sort() from here (and the explicit-stack
 * loop in the RS_SERIAL_SORT debug build); tests/test_sort_emul.py runs the same code on the CPU against
 * the real std::sort and std::partial_sort (tests/csrc/sort_emul_check.cpp).
 */
#ifndef RS_SORT_EMUL_H_
#define RS_SORT_EMUL_H_

#ifndef __HIPCC_RTC__
#include <stdint.h>
#elif !defined(RS_RTC_STDINT)
#define RS_RTC_STDINT /* hiprtc keeps its fixed-width types in a namespace */
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define RS_HD __host__ __device__ __forceinline__
#else
#define RS_HD inline
#endif

namespace rs_sort {

/* comparator of the reference lambda on packed elements: a "less" b  <=>  key(a) > key(b) */
RS_HD bool before(uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }

template <typename A>
RS_HD void swap_at(A& v, int i, int j) {
  uint32_t t = v[i];
  v[i] = v[j];
  v[j] = t;
}

/* std::__move_median_to_first(result, a, b, c) */
template <typename A>
RS_HD void median_to_first(A& v, int result, int a, int b, int c) {
  uint32_t va = v[a], vb = v[b], vc = v[c];
  if (before(va, vb)) {
    if (before(vb, vc)) swap_at(v, result, b);
    else if (before(va, vc)) swap_at(v, result, c);
    else swap_at(v, result, a);
  } else if (before(va, vc)) swap_at(v, result, a);
  else if (before(vb, vc)) swap_at(v, result, c);
  else swap_at(v, result, b);
}

/* std::__unguarded_partition(first, last, pivot) with pivot an element of the array */
template <typename A>
RS_HD int unguarded_partition(A& v, int first, int last, int pivot) {
  uint32_t pv = v[pivot];
  while (true) {
    while (before(v[first], pv)) ++first;
    --last;
    while (before(pv, v[last])) --last;
    if (!(first < last)) return first;
    swap_at(v, first, last);
    ++first;
  }
}

/* std::__adjust_heap + std::__push_heap on the sub-array starting at `base` */
template <typename A>
RS_HD void adjust_heap(A& v, int base, int hole, int len, uint32_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (before(v[base + child], v[base + child - 1])) child--;
    v[base + hole] = v[base + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[base + hole] = v[base + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && before(v[base + parent], value)) {
    v[base + hole] = v[base + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[base + hole] = value;
}

/* std::__partial_sort(first, last, last): __heap_select (= make_heap, empty tail loop) + __sort_heap */
template <typename A>
RS_HD void heap_sort(A& v, int first, int last) {
  int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      uint32_t value = v[first + parent];
      adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    uint32_t value = v[last];
    v[last] = v[first];
    adjust_heap(v, first, 0, last - first, value);
  }
}

RS_HD int floor_log2(int n) {
  int l = 0;
  while (n > 1) { n >>= 1; l++; }
  return l;
}

/* std::__introsort_loop over [0, n): recursion on the right part is replaced by an explicit stack
 * of (first, last, depth) triples in `stk` (capacity >= 3 * (2*floor(log2 n) + 2) ints).
 * depth_limit < 0 selects the library's 2*floor(log2 n). */
template <typename A, typename S>
RS_HD void introsort_loop(A& v, int n, S& stk, int depth_limit = -1) {
  if (n <= 1) return;
  int sp = 0;
  int first = 0, last = n, depth = depth_limit < 0 ? 2 * floor_log2(n) : depth_limit;
  while (true) {
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort(v, first, last);
        break;
      }
      --depth;
      int mid = first + (last - first) / 2;
      median_to_first(v, first, first + 1, mid, last - 1);
      int cut = unguarded_partition(v, first + 1, last, first);
      /* recurse on [cut, last) first (the library's recursive call), then continue with [first, cut) */
      stk[sp++] = first;
      stk[sp++] = cut;
      stk[sp++] = depth;
      first = cut;
    }
    if (sp == 0) break;
    depth = stk[--sp];
    last = stk[--sp];
    first = stk[--sp];
  }
}

}  // namespace rs_sort

#endif /* RS_SORT_EMUL_H_ */
