// TEST INFRASTRUCTURE: the C++ library's own std::make_heap / std::pop_heap over records ordered by probability only (as the
// reference's RatedTarget / RatingsHeapItem, PqaCore/Interface/PqaCommon.h:58-60, PqaCore/RatingsHeap.h:18-20), behind a C ABI, so
// that tests/test_oracle.py can hold oracle/pqa_oracle.c's restated heap steps to libstdc++'s.  Compiled by the test with g++.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct Rec {
  double prob;
  int64_t id;
  bool operator<(const Rec &o) const { return prob < o.prob; }
};
std::vector<Rec> gather(const double *prob, const int64_t *id, int64_t n) {
  std::vector<Rec> v((size_t)n);
  for (int64_t i = 0; i < n; i++) v[(size_t)i] = Rec{prob[i], id[i]};
  return v;
}
void scatter(const std::vector<Rec> &v, double *prob, int64_t *id) {
  for (size_t i = 0; i < v.size(); i++) { prob[i] = v[i].prob; id[i] = v[i].id; }
}
}  // namespace

extern "C" void std_heap_make(double *prob, int64_t *id, int64_t n) {
  std::vector<Rec> v = gather(prob, id, n);
  std::make_heap(v.begin(), v.end());
  scatter(v, prob, id);
}
extern "C" void std_heap_pop(double *prob, int64_t *id, int64_t n) {
  std::vector<Rec> v = gather(prob, id, n);
  std::pop_heap(v.begin(), v.end());
  scatter(v, prob, id);
}
