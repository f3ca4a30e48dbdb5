// The host-side tables of the chain's cyclic reduction (pycollo_amd/csrc/pc_kkt_cr.hpp) under AddressSanitizer + UBSan.
// usage: kkt_cr_sanitize nb unknowns_per_node  n_1 first_1 last_1  n_2 first_2 last_2 ...   (one triple per chain segment)
// prints every table, one line each, for the test to compare with the library's (pc_kkt_cr_plan).
#include <cstdio>
#include <cstdlib>
#include "../../pycollo_amd/csrc/pc_kkt_cr.hpp"

template <class V>
static void line(const char* name, const V& v, size_t n) {
  std::printf("%s", name);
  for (size_t i = 0; i < n; ++i) std::printf(" %lld", (long long)v[i]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 3 || (argc - 3) % 3 != 0) return 2;
  const int64_t nb = std::atoll(argv[1]), per_node = std::atoll(argv[2]);
  std::vector<int64_t> seg_ptr{0};
  std::vector<uint8_t> exp;
  for (int i = 3; i < argc; i += 3) {
    const int64_t n = std::atoll(argv[i]);
    seg_ptr.push_back(seg_ptr.back() + n);
    for (int64_t p = 0; p < n; ++p) exp.push_back((p == 0 && std::atoi(argv[i + 1])) || (p == n - 1 && std::atoi(argv[i + 2])));
  }
  const int64_t nc = seg_ptr.back();
  std::vector<int64_t> chain_ptr((size_t)nc + 1);
  for (int64_t c = 0; c <= nc; ++c) chain_ptr[(size_t)c] = c * per_node;
  CrPlan P;
  try {
    cr_build(nc, (int64_t)seg_ptr.size() - 1, seg_ptr.data(), chain_ptr.data(), nb, exp.data(), P);
  } catch (const std::exception& e) {
    std::printf("error %s\n", e.what());
    return 0;
  }
  line("a", P.ca, (size_t)nc); line("b", P.cb, (size_t)nc); line("mid_a", P.mida, (size_t)nc); line("mid_b", P.midb, (size_t)nc);
  line("level", P.lvl, (size_t)nc); line("pull_ptr", P.pull_ptr, (size_t)nc + 1); line("pull_e", P.pull_e, (size_t)P.n_pull);
  std::printf("buf_len %lld ldsmax %lld max_pull %lld lmax %d\n", (long long)P.buf_len, (long long)P.ldsmax, (long long)P.max_pull, P.lmax);
  return 0;
}
