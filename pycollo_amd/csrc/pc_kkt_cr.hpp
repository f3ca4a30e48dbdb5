// Host-side tables of the chain's cyclic reduction (pure C++, no device code: compiled into pc_kkt.hip, and on its own under
// AddressSanitizer / UBSan by tests/c/kkt_cr_sanitize.cpp).
//
// A segment (a phase's chain, or with cuts a part of one) of n nodes, positions 0 .. n-1: at level l the positions
// (2k+1) 2^(l-1) are eliminated against their neighbours at distance 2^(l-1); position 0 goes last.  A node marked in
// chain_export is NOT eliminated (a rank's part of a factorisation cut across ranks: the node it shares with its neighbour
// rank, pycollo_amd/kkt_sharded.py) -- only the two ends of a segment can be: an exported last node is the right separator
// of every node whose own would lie beyond it ("anchor"), an exported first node keeps position 0's place in the order.
// Their assembled panels [D | K(first, last) | F] are what the rank adds to the reduced system.
//
// Per node: its separators (cr_a, cr_b; -1 none), the node whose Schur block holds its coupling to each (mid; -1 = they
// are neighbours in the chain: the assembled entries), its level, and the eliminated nodes it is a separator of, level by
// level (pull list: node << 1 | 1 if it is that node's LEFT separator).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <utility>
#include <vector>

constexpr int CR_MAX_PULL = 64;   // a wave's lanes look the pull list up: two entries per level below a node's own

struct CrPlan {
  std::vector<int64_t> ca, cb, mida, midb, pull_ptr, oP, oS, oG;
  std::vector<int32_t> pull_e;
  std::vector<int> lvl;
  std::vector<uint8_t> first, exported;
  int lmax = 1;
  int64_t max_pull = 0, n_pull = 0, buf_len = 0, ldsmax = 0;
  bool any_export = false;
};

inline void cr_build(int64_t nc, int64_t n_phase, const int64_t* chain_phase_ptr, const int64_t* chain_ptr, int64_t nb,
                     const uint8_t* chain_export, CrPlan& T) {
  T.ca.assign((size_t)nc, -1); T.cb.assign((size_t)nc, -1); T.mida.assign((size_t)nc, -1); T.midb.assign((size_t)nc, -1);
  T.oP.assign((size_t)nc, 0); T.oS.assign((size_t)nc, 0); T.oG.assign((size_t)nc, 0);
  T.lvl.assign((size_t)nc, 1);
  T.first.assign((size_t)nc, 0); T.exported.assign((size_t)nc, 0);
  auto &ca = T.ca, &cb = T.cb, &mida = T.mida, &midb = T.midb;
  auto &lvl = T.lvl;
  auto &exported = T.exported;
  if (chain_export)
    for (int64_t c = 0; c < nc; ++c) T.any_export |= (exported[(size_t)c] = chain_export[c] != 0) != 0;
  if (nc >= ((int64_t)1 << 30)) throw std::runtime_error("chain too long for the pull tables");
  struct Owe { int64_t sep; int32_t code; };
  std::vector<Owe> owes;                     // (node that pulls, eliminated node << 1 | pulls as the LEFT separator)
  owes.reserve((size_t)2 * (size_t)nc);
  T.lmax = 1;
  for (int64_t ph = 0; ph < n_phase; ++ph) {
    const int64_t c0 = chain_phase_ptr[ph], n = chain_phase_ptr[ph + 1] - c0;
    if (n <= 0) continue;
    if (c0 < 0 || c0 + n > nc) throw std::runtime_error("chain segment out of range");
    T.first[(size_t)c0] = 1;
    for (int64_t p = 1; p + 1 < n; ++p)
      if (exported[(size_t)(c0 + p)]) throw std::runtime_error("only the first and the last node of a chain segment can be exported");
    const bool expN = n >= 2 && exported[(size_t)(c0 + n - 1)];
    const int64_t anchor = expN ? c0 + n - 1 : -1, n1 = expN ? n - 1 : n;
    int levels = 0;
    while (((int64_t)1 << levels) <= n1 - 1) ++levels;      // number of odd-even levels
    for (int64_t p = 0; p < n1; ++p) {
      if (p == 0) { lvl[(size_t)c0] = levels + 1; cb[(size_t)c0] = anchor; continue; }
      const int l = __builtin_ctzll((unsigned long long)p) + 1;
      const int64_t h = (int64_t)1 << (l - 1);
      lvl[(size_t)(c0 + p)] = l;
      ca[(size_t)(c0 + p)] = c0 + p - h;
      cb[(size_t)(c0 + p)] = p + h <= n1 - 1 ? c0 + p + h : anchor;
    }
    if (expN) lvl[(size_t)anchor] = levels + 2;             // after position 0, which may still be eliminated against it
    T.lmax = std::max(T.lmax, levels + (expN ? 2 : 1));
    // in level order: where a node's couplings to its separators come from (the node eliminated between them last,
    // -1 = they are neighbours in the chain: the assembled entries), and what an eliminated node owes its separators
    std::vector<std::vector<int64_t>> by_level((size_t)levels + 3);
    for (int64_t p = 0; p < n; ++p) by_level[(size_t)lvl[(size_t)(c0 + p)]].push_back(c0 + p);
    std::map<std::pair<int64_t, int64_t>, int64_t> link;
    for (size_t L = 1; L < by_level.size(); ++L)
      for (int64_t c : by_level[L]) {
        if (ca[(size_t)c] >= 0) {
          auto it = link.find({ca[(size_t)c], c});
          mida[(size_t)c] = it != link.end() ? it->second : -1;
          if (mida[(size_t)c] < 0 && ca[(size_t)c] != c - 1) throw std::runtime_error("cyclic reduction: a separator without a coupling (internal)");
        }
        if (cb[(size_t)c] >= 0) {
          auto it = link.find({c, cb[(size_t)c]});
          midb[(size_t)c] = it != link.end() ? it->second : -1;
          if (midb[(size_t)c] < 0 && cb[(size_t)c] != c + 1) throw std::runtime_error("cyclic reduction: a separator without a coupling (internal)");
        }
        if (exported[(size_t)c]) continue;
        if (cb[(size_t)c] >= 0) owes.push_back({cb[(size_t)c], (int32_t)(c << 1)});          // (the right separator's term first, as the
        if (ca[(size_t)c] >= 0) owes.push_back({ca[(size_t)c], (int32_t)((c << 1) | 1)});   //  levels were summed before these tables)
        if (ca[(size_t)c] >= 0 && cb[(size_t)c] >= 0) link[{ca[(size_t)c], cb[(size_t)c]}] = c;
      }
  }
  // CSR by pulling node, stable in the order collected (level by level; inside a level by position: for a separator c
  // the node c - h comes before c + h)
  T.pull_ptr.assign((size_t)nc + 1, 0);
  for (const Owe& o : owes) ++T.pull_ptr[(size_t)o.sep + 1];
  T.max_pull = 0;
  for (int64_t c = 0; c < nc; ++c) {
    T.max_pull = std::max(T.max_pull, T.pull_ptr[(size_t)c + 1]);
    T.pull_ptr[(size_t)c + 1] += T.pull_ptr[(size_t)c];
  }
  T.n_pull = (int64_t)owes.size();
  T.pull_e.assign(owes.size() ? owes.size() : 1, 0);
  {
    std::vector<int64_t> fill(T.pull_ptr.begin(), T.pull_ptr.end() - 1);
    for (const Owe& o : owes) T.pull_e[(size_t)fill[(size_t)o.sep]++] = o.code;
  }
  int64_t off = 0;
  T.ldsmax = 0;
  auto nzof = [&](int64_t c) { return chain_ptr[c + 1] - chain_ptr[c]; };
  for (int64_t c = 0; c < nc; ++c) {
    const int64_t nz = nzof(c), w = (ca[(size_t)c] >= 0 ? nzof(ca[(size_t)c]) : 0) + (cb[(size_t)c] >= 0 ? nzof(cb[(size_t)c]) : 0) + nb;
    T.oP[(size_t)c] = off; off += nz * (nz + w);
    T.oS[(size_t)c] = off; off += w * w;
    T.oG[(size_t)c] = off; off += w;
    T.ldsmax = std::max(T.ldsmax, 8 * (2 * nz + w + nz * (nz + w) + w * w + 2));
    if (nz > 64 || w > 64) T.ldsmax = (int64_t)1 << 30;
  }
  T.buf_len = off;
}
