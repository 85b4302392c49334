// introsort_emul_test.cpp — CPU fuzz of moving_object_detector_amd/csrc/introsort_emul.h (product code, host build)
// against the real libstdc++ std::sort / std::make_heap / std::sort_heap with the reference's comparator
// (clusterer_nodelet.cpp:168-172: descending ||v||).  Exit code 0 = all agree.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

#include "../../moving_object_detector_amd/csrc/introsort_emul.h"

using Pair = std::pair<uint32_t, uint32_t>;   // (key, original index)
static bool cmp(const Pair &a, const Pair &b) { return a.first > b.first; }

int main() {
  std::mt19937 rng(12345);
  long checks = 0;
  for (int trial = 0; trial < 3000; trial++) {
    const int n = 1 + (int)(rng() % (trial % 10 == 0 ? 60000 : 3000));
    const int distinct = 1 + (int)(rng() % (trial % 3 == 0 ? 4 : (trial % 3 == 1 ? 50 : 100000)));   // heavy ties .. none
    std::vector<uint32_t> keys(n);
    for (auto &k : keys) k = 0x3f000000u + (rng() % distinct) * 977u;
    if (trial % 7 == 0) std::sort(keys.begin(), keys.end());                       // pre-sorted ascending
    if (trial % 11 == 0) std::sort(keys.begin(), keys.end(), std::greater<uint32_t>());
    std::vector<Pair> ref(n);
    for (int i = 0; i < n; i++) ref[i] = {keys[i], (uint32_t)i};
    std::sort(ref.begin(), ref.end(), cmp);
    const int wants[4] = {n / 2, 0, n - 1, (int)(rng() % n)};
    for (int w : wants) {
      std::vector<uint32_t> k2 = keys, v2(n);
      for (int i = 0; i < n; i++) v2[i] = (uint32_t)i;
      introsort_emul::View v{k2.data(), v2.data()};
      const uint32_t got = introsort_emul::element_at_sequential(v, n, w);
      if (got != ref[w].second) { printf("MISMATCH trial %d n %d want %d: got %u expected %u\n", trial, n, w, got, ref[w].second); return 1; }
      checks++;
    }
    // heap-sort restatement (the depth-exhaustion branch of introsort): arrangement must equal make_heap + sort_heap
    {
      std::vector<Pair> h(n);
      for (int i = 0; i < n; i++) h[i] = {keys[i], (uint32_t)i};
      const int a = (int)(rng() % n), b = a + 1 + (int)(rng() % (n - a));
      std::make_heap(h.begin() + a, h.begin() + b, cmp);
      std::sort_heap(h.begin() + a, h.begin() + b, cmp);
      std::vector<uint32_t> k2 = keys, v2(n);
      for (int i = 0; i < n; i++) v2[i] = (uint32_t)i;
      introsort_emul::View v{k2.data(), v2.data()};
      introsort_emul::heap_sort(v, a, b);
      for (int i = 0; i < n; i++)
        if (k2[i] != h[i].first || v2[i] != h[i].second) { printf("HEAP MISMATCH trial %d n %d at %d\n", trial, n, i); return 2; }
      checks++;
    }
  }
  printf("ok %ld checks\n", checks);
  return 0;
}
