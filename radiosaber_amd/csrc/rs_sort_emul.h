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

/* ---- Iteration order of libstdc++'s std::unordered_map<int, int> (GCC 11, <bits/hashtable.h> + _Prime_rehash_policy), restated
 * for host and device code.
 *
 * The reference's SubOpt (downlink-transport-scheduler.cpp:296-306, 315) fills `slice_fewer` with the slices below their
 * quota in ascending slice order and afterwards iterates it with strict `<` comparisons, so ties between equal
 * efficiency losses go to the slice the hashtable happens to yield first.  Later `erase` calls unlink nodes without moving
 * the others, hence the order of the survivors is the order right after the insertions, which this file computes:
 *
 *  - one singly linked list of all nodes; the nodes of a bucket are contiguous; a bucket remembers the node BEFORE its first;
 *  - insert into a non-empty bucket: right behind that "before" node (front of the bucket's group); into an empty bucket:
 *    at the front of the whole list (and the bucket of the former first node now has the new node as its "before");
 *  - identity hash, bucket = key % bucket_count; bucket counts 13, 29, 59, 127, grown when the element count would exceed
 *    the bucket count (max_load_factor 1; the first insertion allocates 13 buckets);
 *  - a rehash walks the old list front to back and re-inserts each node by the same two rules.
 *
 * tests/test_sort_emul.py::test_unordered_map_order_matches_libstdcxx checks this against the real container on every
 * subset it draws; the oracle calls the real container.
 */
#define RS_UMAP_NIL 0xFF
#define RS_UMAP_HEAD 64 /* index of the list's before-begin node in nxt[] */
#ifndef RS_UMAP_SCRATCH_BYTES
#define RS_UMAP_SCRATCH_BYTES 272 /* nxt u8[68] | bkt u8[128] | ord u8[64] (+ padding) */
#endif

/* keys: bit k set = key k inserted (ascending); nxt[65], bkt[127]: scratch; ord[64]: the keys in iteration order.
 * Returns the number of keys. */
RS_HD int rs_umap_order(uint64_t keys, uint8_t* nxt, uint8_t* bkt, uint8_t* ord) {
  int n_bkt = 0, count = 0;
  nxt[RS_UMAP_HEAD] = RS_UMAP_NIL;
  for (int key = 0; key < 64; ++key) {
    if (!((keys >> key) & 1ull)) continue;
    if (count + 1 > n_bkt) { /* _M_need_rehash: 0 -> 13 -> 29 -> 59 -> 127 */
      n_bkt = n_bkt == 0 ? 13 : (n_bkt == 13 ? 29 : (n_bkt == 29 ? 59 : 127));
      for (int b = 0; b < n_bkt; ++b) bkt[b] = RS_UMAP_NIL;
      int p = nxt[RS_UMAP_HEAD], front_bkt = 0;
      nxt[RS_UMAP_HEAD] = RS_UMAP_NIL;
      while (p != RS_UMAP_NIL) { /* _M_rehash_aux(unique keys) */
        const int next = nxt[p], b = p % n_bkt;
        if (bkt[b] == RS_UMAP_NIL) {
          nxt[p] = nxt[RS_UMAP_HEAD];
          nxt[RS_UMAP_HEAD] = (uint8_t)p;
          bkt[b] = RS_UMAP_HEAD;
          if (nxt[p] != RS_UMAP_NIL) bkt[front_bkt] = (uint8_t)p;
          front_bkt = b;
        } else {
          nxt[p] = nxt[bkt[b]];
          nxt[bkt[b]] = (uint8_t)p;
        }
        p = next;
      }
    }
    const int b = key % n_bkt; /* _M_insert_bucket_begin */
    if (bkt[b] != RS_UMAP_NIL) {
      nxt[key] = nxt[bkt[b]];
      nxt[bkt[b]] = (uint8_t)key;
    } else {
      nxt[key] = nxt[RS_UMAP_HEAD];
      nxt[RS_UMAP_HEAD] = (uint8_t)key;
      if (nxt[key] != RS_UMAP_NIL) bkt[nxt[key] % n_bkt] = (uint8_t)key;
      bkt[b] = RS_UMAP_HEAD;
    }
    ++count;
  }
  int m = 0;
  for (int p = nxt[RS_UMAP_HEAD]; p != RS_UMAP_NIL; p = nxt[p]) ord[m++] = (uint8_t)p;
  return m;
}

#endif /* RS_SORT_EMUL_H_ */
