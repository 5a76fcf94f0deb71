// cluster.h -- the host half of make_new_grids (grids.hip): Berger-Rigoutsos clustering of a lattice of tagged blocks and the merge of the boxes it returns.
// Plain C++ (no HIP): tests/test_cluster_cpu.py compiles it with g++ and holds the merge pass against the start-again-from-(0,1) rule it replaces.
// make_new_grids is FBoxLib's and not in the reference tree (call sites: src/initialize.f90:247-248, src/regrid.f90:148-149); the procedure is ours.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>
namespace vdn_cluster {
struct IBox { int lo[3], hi[3]; };
struct Lattice;
inline bool all_allowed(const Lattice &G, const IBox &b);
struct Lattice {
  int n[3]; std::vector<unsigned char> t, ok;                 // t: tagged blocks; ok: blocks that lie in the nesting region as a whole
  unsigned char at(int i, int j, int k) const { return t[(size_t)i + (size_t)n[0] * ((size_t)j + (size_t)n[1] * (size_t)k)]; }
  unsigned char allowed(int i, int j, int k) const { return ok[(size_t)i + (size_t)n[0] * ((size_t)j + (size_t)n[1] * (size_t)k)]; }
};
inline bool all_allowed(const Lattice &G, const IBox &b) {
  for (int k = b.lo[2]; k <= b.hi[2]; k++) for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) if (!G.allowed(i, j, k)) return false;
  return true;
}
inline long count_tags(const Lattice &G, const IBox &b) {
  long c = 0;
  for (int k = b.lo[2]; k <= b.hi[2]; k++) for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) c += G.at(i, j, k);
  return c;
}
inline bool shrink_to_tags(const Lattice &G, IBox &b) {
  int lo[3] = { b.hi[0] + 1, b.hi[1] + 1, b.hi[2] + 1 }, hi[3] = { b.lo[0] - 1, b.lo[1] - 1, b.lo[2] - 1 };
  for (int k = b.lo[2]; k <= b.hi[2]; k++) for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++)
    if (G.at(i, j, k)) { const int q[3] = { i, j, k }; for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], q[d]); hi[d] = std::max(hi[d], q[d]); } }
  if (lo[0] > hi[0]) return false;
  for (int d = 0; d < 3; d++) { b.lo[d] = lo[d]; b.hi[d] = hi[d]; }
  return true;
}
inline void cluster(const Lattice &G, IBox b, double min_eff, int min_width, std::vector<IBox> &out) {
  if (!shrink_to_tags(G, b)) return;
  const long vol = (long)(b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1) * (b.hi[2] - b.lo[2] + 1);
  const long ntag = count_tags(G, b);
  // a box is acceptable only if it stays inside the nesting region (an efficient box may still contain an untagged block outside it)
  const bool nested = all_allowed(G, b);
  if (nested && (double)ntag >= min_eff * (double)vol) { out.push_back(b); return; }
  // signatures
  std::vector<long> sig[3];
  for (int d = 0; d < 3; d++) sig[d].assign(b.hi[d] - b.lo[d] + 1, 0);
  for (int k = b.lo[2]; k <= b.hi[2]; k++) for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++)
    if (G.at(i, j, k)) { sig[0][i - b.lo[0]]++; sig[1][j - b.lo[1]]++; sig[2][k - b.lo[2]]++; }
  int cut_d = -1, cut_at = -1;                 // the box is cut between cut_at-1 and cut_at (index relative to b.lo)
  // (a) a hole in a signature; the one closest to the middle of its side, longest side first
  int order[3] = { 0, 1, 2 };
  std::sort(order, order + 3, [&](int x, int y) { return sig[x].size() > sig[y].size() || (sig[x].size() == sig[y].size() && x < y); });
  for (int o = 0; o < 3 && cut_d < 0; o++) {
    const int d = order[o], len = (int)sig[d].size();
    int best = -1;
    for (int c = min_width; c <= len - min_width; c++) if (sig[d][c] == 0 || sig[d][c - 1] == 0) { if (best < 0 || std::abs(2 * c - len) < std::abs(2 * best - len)) best = c; }
    if (best >= 0) { cut_d = d; cut_at = best; }
  }
  // (b) the strongest inflection of the second difference of a signature
  if (cut_d < 0) {
    long best_jump = 0;
    for (int o = 0; o < 3; o++) {
      const int d = order[o], len = (int)sig[d].size();
      if (len < 2 * min_width || len < 4) continue;
      std::vector<long> lap(len, 0);
      for (int c = 1; c < len - 1; c++) lap[c] = sig[d][c - 1] - 2 * sig[d][c] + sig[d][c + 1];
      for (int c = std::max(min_width, 2); c <= std::min(len - min_width, len - 2); c++) {
        if ((lap[c - 1] < 0) != (lap[c] < 0) || (lap[c - 1] == 0) != (lap[c] == 0)) {
          const long jump = std::labs(lap[c] - lap[c - 1]);
          if (jump > best_jump || (jump == best_jump && cut_d == d && std::abs(2 * c - len) < std::abs(2 * cut_at - len))) { best_jump = jump; cut_d = d; cut_at = c; }
        }
      }
    }
  }
  // (c) the middle of the longest side
  if (cut_d < 0) {
    const int d = order[0], len = (int)sig[d].size();
    if (len >= 2 * min_width) { cut_d = d; cut_at = len / 2; }
  }
  if (cut_d < 0 && !nested) {                          // too small for the usual rules but not nested: halve the longest side anyway
    const int d = order[0], len = (int)sig[d].size();
    if (len >= 2) { cut_d = d; cut_at = len / 2; }
  }
  if (cut_d < 0) { out.push_back(b); return; }       // cannot be cut: accept
  IBox l = b, r = b;
  l.hi[cut_d] = b.lo[cut_d] + cut_at - 1; r.lo[cut_d] = b.lo[cut_d] + cut_at;
  cluster(G, l, min_eff, min_width, out);
  cluster(G, r, min_eff, min_width, out);
}

inline int mergeable(const IBox &p, const IBox &q) {          // the direction in which the union of p and q is a box, or -1
  for (int d = 0; d < 3; d++) {
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    if (p.lo[t1] != q.lo[t1] || p.hi[t1] != q.hi[t1] || p.lo[t2] != q.lo[t2] || p.hi[t2] != q.hi[t2]) continue;
    if (p.hi[d] + 1 == q.lo[d] || q.hi[d] + 1 == p.lo[d]) return d;
  }
  return -1;
}
// merge neighbours whose union is again a box (the recursion cuts more than the final box set needs, e.g. where a cut for the nesting region or a hole left two
// boxes of equal cross-section side by side): fewer, larger boxes for the same cells.
// The rule: take the first pair (a, b), a < b, in lexicographic order whose union is a box, replace a by the union, drop b, start again.  Starting again
// from (0, 1) each time is cubic in the number of boxes (a thousand on a 512^3 level); the same sequence of merges comes out of keeping the row: after a
// merge into row a only pairs WITH the changed box can have become mergeable -- (x, a) for x < a, in increasing x, then row a from a + 1 on.
inline void merge_boxes(std::vector<IBox> &cl) {
  auto absorb = [&](size_t a, size_t b) {                  // cl[a] = cl[a] u cl[b]; b goes
    const int d = mergeable(cl[a], cl[b]);
    cl[a].lo[d] = std::min(cl[a].lo[d], cl[b].lo[d]); cl[a].hi[d] = std::max(cl[a].hi[d], cl[b].hi[d]);
    cl.erase(cl.begin() + (long)b);
  };
  for (size_t row = 0; row < cl.size();) {
    size_t b = row + 1;
    while (b < cl.size() && mergeable(cl[row], cl[b]) < 0) b++;
    if (b == cl.size()) { row++; continue; }
    absorb(row, b);
    size_t dirty = row;
    for (bool again = true; again;) {
      again = false;
      for (size_t x = 0; x < dirty; x++) if (mergeable(cl[x], cl[dirty]) >= 0) { absorb(x, dirty); dirty = x; again = true; break; }
    }
    row = dirty;
  }
}
}  // namespace vdn_cluster
