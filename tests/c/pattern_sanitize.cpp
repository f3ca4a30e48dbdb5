// CPU-only sanitizer harness for the host side of the C ABI (TEST INFRASTRUCTURE).
//
// Compiles the SAME headers the library's pc_create runs -- pc_desc.hpp (descriptor -> pcp::Problem, LDS sizing) and
// pc_pattern.hpp (mesh prefix tables, tiles, NLP layout, CSR patterns of G and H, producer-slot tables: ~600 lines of
// index arithmetic) -- with g++ -fsanitize=address,undefined, reads a problem description as text, builds everything
// and prints the index arrays.  tests/test_sanitized_host.py feeds it the descriptors the library gets and compares
// its output with the library's (sanitizer builds on the CPU only; GPU ASan / XNACK are not
// available on the pool).
//
//   usage: pattern_sanitize <in.txt> <out.txt>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../pycollo_amd/csrc/pc_desc.hpp"

namespace {

template <class T>
std::vector<T> read_vec(std::istream& in, long n) {
  std::vector<T> v((size_t)(n > 0 ? n : 0));
  for (auto& e : v)
    if (!(in >> e)) throw std::runtime_error("input truncated");
  return v;
}

struct PhaseIn {
  std::vector<int32_t> n_k, jr, jc, hr, hc, wk, wi, fk, fo;
  std::vector<double> h_k;
};

template <class T>
void put(std::ostream& out, const char* name, const std::vector<T>& v) {
  out << name << ' ' << v.size();
  for (const auto& e : v) out << ' ' << (long long)e;
  out << '\n';
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <in.txt> <out.txt>\n", argv[0]);
    return 2;
  }
  try {
    std::ifstream in(argv[1]);
    if (!in) throw std::runtime_error("cannot open input");
    pc_problem_desc d{};
    int tile_nodes = 0, qa_total = 0, qw_total = 0;
    in >> d.n_phases >> d.n_s >> d.n_point >> d.n_b >> d.n_jgrad >> d.n_bjac >> d.n_pthess >> tile_nodes >> qa_total >> qw_total;
    if (!in || d.n_phases < 1 || d.n_phases > PC_MAX_PHASES) throw std::runtime_error("bad header");
    std::vector<pc_phase_desc> phases((size_t)d.n_phases);
    std::vector<PhaseIn> store((size_t)d.n_phases);
    for (int ip = 0; ip < d.n_phases; ++ip) {
      pc_phase_desc& s = phases[ip];
      PhaseIn& a = store[ip];
      s = pc_phase_desc{};
      in >> s.n_y >> s.n_u >> s.n_q >> s.n_p >> s.t0_free >> s.tF_free >> s.K >> s.n_jac >> s.n_hess >> s.n_w >> s.compiled_order;
      // mixed build: the orders with a tile body; a caller's tile table (pc_phase_desc::fixed_tile_k0 / fixed_tile_order)
      in >> s.n_spec >> s.spec_orders[0] >> s.spec_orders[1] >> s.spec_orders[2] >> s.spec_orders[3] >> s.n_fixed_tiles;
      if (!in) throw std::runtime_error("bad phase header");
      a.n_k = read_vec<int32_t>(in, s.K);
      a.h_k = read_vec<double>(in, s.K);
      a.jr = read_vec<int32_t>(in, s.n_jac);
      a.jc = read_vec<int32_t>(in, s.n_jac);
      a.hr = read_vec<int32_t>(in, s.n_hess);
      a.hc = read_vec<int32_t>(in, s.n_hess);
      a.wk = read_vec<int32_t>(in, s.n_w);
      a.wi = read_vec<int32_t>(in, s.n_w);
      a.fk = read_vec<int32_t>(in, s.n_fixed_tiles > 0 ? s.n_fixed_tiles + 1 : 0);
      a.fo = read_vec<int32_t>(in, s.n_fixed_tiles > 0 ? s.n_fixed_tiles : 0);
      s.fixed_tile_k0 = s.n_fixed_tiles > 0 ? a.fk.data() : nullptr;
      s.fixed_tile_order = s.n_fixed_tiles > 0 ? a.fo.data() : nullptr;
      s.n_k = a.n_k.data(); s.h_k = a.h_k.data();
      s.jac_row = a.jr.data(); s.jac_col = a.jc.data();
      s.hess_row = a.hr.data(); s.hess_col = a.hc.data();
      s.w_kind = s.n_w > 0 ? a.wk.data() : nullptr;
      s.w_idx = s.n_w > 0 ? a.wi.data() : nullptr;
      s.bulk_kernel = "pc_bulk";
    }
    d.phases = phases.data();
    auto pp = read_vec<int32_t>(in, d.n_point), pk = read_vec<int32_t>(in, d.n_point), pi = read_vec<int32_t>(in, d.n_point);
    auto jg = read_vec<int32_t>(in, d.n_jgrad);
    auto br = read_vec<int32_t>(in, d.n_bjac), bc = read_vec<int32_t>(in, d.n_bjac);
    auto phr = read_vec<int32_t>(in, d.n_pthess), phc = read_vec<int32_t>(in, d.n_pthess);
    d.point_phase = pp.data(); d.point_kind = pk.data(); d.point_idx = pi.data();
    d.jgrad_col = jg.data(); d.bjac_row = br.data(); d.bjac_col = bc.data();
    d.pthess_row = phr.data(); d.pthess_col = phc.data();
    d.device = -1;

    // what pc_create does on a structure-only handle
    pcp::Problem Q;
    pcp::from_desc(d, Q);
    for (auto& P : Q.ph) pcp::finalize_phase_tables(P, Q.n_s);
    for (auto& P : Q.ph)   // mixed build: the row caps pc_create sets for one wave per tile (64 KiB of LDS)
      if (!P.spec_orders.empty()) {
        pcp::phase_set_caps(P, tile_nodes, 1, qa_total, qw_total, 64 * 1024);
        P.mix_cap_rows = std::min(P.mix_cap_rows, 32);
      }
    pcp::build_all(Q, tile_nodes);

    std::ofstream out(argv[2]);
    out << "sizes " << Q.num_x << ' ' << Q.num_c << ' ' << Q.g_row.size() << ' ' << Q.h_row.size() << '\n';
    put(out, "g_row", Q.g_row); put(out, "g_col", Q.g_col); put(out, "h_row", Q.h_row); put(out, "h_col", Q.h_col);
    put(out, "g_indptr", Q.g_indptr); put(out, "h_indptr", Q.h_indptr);
    put(out, "tail_owned", Q.tail_owned);
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      const auto& P = Q.ph[ip];
      put(out, "tile_k0", P.tile_k0);
      put(out, "tile_order", P.tile_order);
      if (!P.spec_orders.empty()) {   // the exact staging size of the tiles as cut, row groups included, in both forms
        bool ap = false, ag = false;
        std::vector<int> m = {pcp::phase_lds_out_tiles(P, &ap, &ag), pcp::phase_lds_out_tiles(P, nullptr, nullptr, false), (int)ap, (int)ag};
        for (int n = 2; n <= PC_MAX_ORDER; ++n) m.push_back(pc_row_passes(n, pcp::phase_max_row_len(P, n)));
        put(out, "mixed", m);
      }
      put(out, "goff", P.goff); put(out, "hoff", P.hoff); put(out, "hslot0", P.hslot0); put(out, "hslotN", P.hslotN);
      const int rows = pcp::phase_max_tile_rows(P);
      const int lds_out = pcp::phase_lds_out(P, rows, std::min(tile_nodes, rows + 1));
      std::vector<int> lds = {rows, lds_out};
      for (int W : {1, 2, 4}) lds.push_back(pcp::phase_lds_bytes(P, 64, qa_total, qw_total, lds_out * W, P.compiled_order == 0));
      put(out, "lds", lds);
      // every slot table entry must address the pattern
      for (auto v : P.hslot0) if (v < 0 || v >= (int64_t)Q.h_row.size()) throw std::runtime_error("hslot0 out of range");
      for (auto v : P.hslotN) if (v < 0 || v >= (int64_t)Q.h_row.size()) throw std::runtime_error("hslotN out of range");
    }
    out << "ok\n";
    return out ? 0 : 3;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "pattern_sanitize: %s\n", e.what());
    return 1;
  }
}
