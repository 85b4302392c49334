// introsort_emul.h — which element does libstdc++'s std::sort leave at a given position?
//
// cluster2MovingObject (scene_flow_clusterer/src/clusterer_nodelet.cpp:168-174) sorts a cluster's points by ||v|| with
// std::sort and reports the element at size/2.  std::sort is unstable: when several members tie on ||v|| but carry
// different vectors, which of them lands at size/2 is decided by the exact moves of libstdc++'s introsort
// (bits/stl_algo.h: __introsort_loop, __move_median_to_first, __unguarded_partition, __partial_sort on depth exhaustion,
// __final_insertion_sort; threshold 16, depth limit 2*floor(log2 n)).  The reference's third-party dependency here is
// libstdc++ (gcc 7 in the reference's docker image; the algorithm is unchanged through gcc 11, which the oracle uses).
//
// Only the sub-range that contains the wanted position has to be followed:
//   * both halves of a partition are sorted independently, so the other half never influences this one;
//   * the final insertion sort never moves an element across a partition boundary (everything left of a cut is <= the
//     pivot <= everything right of it in the comparator's order), and inside a finished (<= 16 element) range it is a
//     stable insertion sort.
// The functions below are sequential and shared by the host test (tests/cpp/introsort_emul_test.cpp, fuzzed against
// std::sort) and the device code; the GPU kernel k_median_ties replaces the partition by a block-parallel equivalent.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define IE_HD __host__ __device__ __forceinline__
#else
#define IE_HD inline
#endif

namespace introsort_emul {

// The sort is by ||v|| descending: comp(a, b) = key[a] > key[b] on the F32 bit patterns (all norms finite, >= 0).
struct View {
  uint32_t *key;   // sort keys, permuted in place
  uint32_t *val;   // payload (pixel index), permuted alongside
  IE_HD bool comp(int a, int b) const { return key[a] > key[b]; }
  IE_HD void swap(int a, int b) const {
    const uint32_t k = key[a], v = val[a];
    key[a] = key[b]; val[a] = val[b];
    key[b] = k; val[b] = v;
  }
};

IE_HD int floor_log2(int n) { int l = 0; while (n > 1) { n >>= 1; l++; } return l; }

// __move_median_to_first(result, a, b, c)
IE_HD void move_median_to_first(const View &v, int result, int a, int b, int c) {
  if (v.comp(a, b)) {
    if (v.comp(b, c)) v.swap(result, b);
    else if (v.comp(a, c)) v.swap(result, c);
    else v.swap(result, a);
  } else if (v.comp(a, c)) v.swap(result, a);
  else if (v.comp(b, c)) v.swap(result, c);
  else v.swap(result, b);
}

// __unguarded_partition(first, last, pivot) — sequential form
IE_HD int unguarded_partition(const View &v, int first, int last, int pivot) {
  while (true) {
    while (v.comp(first, pivot)) ++first;
    --last;
    while (v.comp(pivot, last)) --last;
    if (!(first < last)) return first;
    v.swap(first, last);
    ++first;
  }
}

// ---- heap sort of [first, last): __partial_sort(first, last, last) = __make_heap + __sort_heap ----
IE_HD void push_heap_(const View &v, int first, int hole, int top, uint32_t vk, uint32_t vv) {
  int parent = (hole - 1) / 2;
  while (hole > top && v.key[first + parent] > vk) {     // comp(first + parent, value)
    v.key[first + hole] = v.key[first + parent]; v.val[first + hole] = v.val[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v.key[first + hole] = vk; v.val[first + hole] = vv;
}
IE_HD void adjust_heap_(const View &v, int first, int hole, int len, uint32_t vk, uint32_t vv) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (v.comp(first + child, first + (child - 1))) child--;
    v.key[first + hole] = v.key[first + child]; v.val[first + hole] = v.val[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v.key[first + hole] = v.key[first + (child - 1)]; v.val[first + hole] = v.val[first + (child - 1)];
    hole = child - 1;
  }
  push_heap_(v, first, hole, top, vk, vv);
}
IE_HD void heap_sort(const View &v, int first, int last) {
  const int len = last - first;
  if (len >= 2) {                                           // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const uint32_t vk = v.key[first + parent], vv = v.val[first + parent];
      adjust_heap_(v, first, parent, len, vk, vv);
      if (parent == 0) break;
      parent--;
    }
  }
  int end = last;
  while (end - first > 1) {                                 // __sort_heap: __pop_heap(first, end - 1, end - 1)
    --end;
    const uint32_t vk = v.key[end], vv = v.val[end];
    v.key[end] = v.key[first]; v.val[end] = v.val[first];
    adjust_heap_(v, first, 0, end - first, vk, vv);
  }
}

// stable insertion sort of a finished range (what __final_insertion_sort does to it)
IE_HD void insertion_sort(const View &v, int first, int last) {
  for (int i = first + 1; i < last; i++) {
    const uint32_t vk = v.key[i], vv = v.val[i];
    int j = i;
    while (j > first && vk > v.key[j - 1]) { v.key[j] = v.key[j - 1]; v.val[j] = v.val[j - 1]; j--; }   // comp(val, j-1)
    v.key[j] = vk; v.val[j] = vv;
  }
}

// Payload of the element std::sort(keys by descending key) leaves at position `want` of n elements given in `v`'s order.
// Sequential reference of the whole procedure; `v` is permuted.
IE_HD uint32_t element_at_sequential(const View &v, int n, int want) {
  int first = 0, last = n, depth = 2 * floor_log2(n);
  while (last - first > 16) {
    if (depth == 0) { heap_sort(v, first, last); return v.val[want]; }
    --depth;
    const int mid = first + (last - first) / 2;
    move_median_to_first(v, first, first + 1, mid, last - 1);
    const int cut = unguarded_partition(v, first + 1, last, first);
    if (want >= cut) first = cut; else last = cut;
  }
  insertion_sort(v, first, last);
  return v.val[want];
}

}  // namespace introsort_emul
