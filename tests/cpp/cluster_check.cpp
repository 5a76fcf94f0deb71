// tests/test_cluster_cpu.py builds and runs this: the host half of make_new_grids (varden_amd/csrc/cluster.h) without a GPU.
//  1. merge_boxes (one pass, keeping the row) against the rule it replaces -- find the first mergeable pair in lexicographic order, merge, START AGAIN -- on random
//     box sets and on the sets the clustering itself produces: the same boxes in the same order.
//  2. cluster + merge on random tag lattices with a nesting mask: the boxes are disjoint, cover every tagged block, hold only allowed blocks, and every box is either
//     efficient enough or could not be cut.
#include "cluster.h"
#include <cstdio>
#include <cstdint>
#include <cstring>
using namespace vdn_cluster;
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
static void merge_naive(std::vector<IBox> &cl) {
  for (bool merged = true; merged;) {
    merged = false;
    for (size_t a = 0; a < cl.size() && !merged; a++)
      for (size_t b = a + 1; b < cl.size() && !merged; b++) {
        const int d = mergeable(cl[a], cl[b]);
        if (d < 0) continue;
        cl[a].lo[d] = std::min(cl[a].lo[d], cl[b].lo[d]); cl[a].hi[d] = std::max(cl[a].hi[d], cl[b].hi[d]);
        cl.erase(cl.begin() + (long)b); merged = true;
      }
  }
}
static bool same(const std::vector<IBox> &a, const std::vector<IBox> &b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &b[i], sizeof(IBox)) != 0) return false;
  return true;
}
// a random partition of an n^3 lattice into boxes (recursive cuts), shuffled: plenty of mergeable neighbours, chains of merges
static void partition(IBox b, int depth, std::vector<IBox> &out) {
  int len[3]; for (int d = 0; d < 3; d++) len[d] = b.hi[d] - b.lo[d] + 1;
  if (depth == 0 || (len[0] == 1 && len[1] == 1 && len[2] == 1) || rnd() % 7 == 0) { out.push_back(b); return; }
  int d = (int)(rnd() % 3); for (int t = 0; t < 3 && len[d] == 1; t++) d = (d + 1) % 3;
  const int cut = 1 + (int)(rnd() % (uint32_t)(len[d] - 1));
  IBox l = b, r = b; l.hi[d] = b.lo[d] + cut - 1; r.lo[d] = b.lo[d] + cut;
  partition(l, depth - 1, out); partition(r, depth - 1, out);
}
int main() {
  int cases = 0;
  for (int trial = 0; trial < 400; trial++) {
    const int n = 2 + (int)(rnd() % 7);
    IBox whole; for (int d = 0; d < 3; d++) { whole.lo[d] = 0; whole.hi[d] = n - 1; }
    std::vector<IBox> v; partition(whole, 3 + (int)(rnd() % 8), v);
    for (size_t i = v.size(); i > 1; i--) std::swap(v[i - 1], v[rnd() % i]);
    if (rnd() % 2) for (size_t i = 0; i < v.size() / 3; i++) v.erase(v.begin() + (long)(rnd() % v.size()));      // holes: merges that stop short
    std::vector<IBox> a = v, b = v;
    merge_boxes(a); merge_naive(b);
    if (!same(a, b)) { printf("FAIL: merge_boxes differs from the start-again rule (trial %d, %zu boxes in, %zu / %zu out)\n", trial, v.size(), a.size(), b.size()); return 1; }
    cases++;
  }
  for (int trial = 0; trial < 120; trial++) {
    Lattice G; for (int d = 0; d < 3; d++) G.n[d] = 4 + (int)(rnd() % 21);
    if (trial % 5 == 0) G.n[2] = 1;                                                 // (the 2-D lattice of dm = 2)
    const size_t nb = (size_t)G.n[0] * G.n[1] * G.n[2];
    G.t.assign(nb, 0); G.ok.assign(nb, 1);
    // tags: a few blobs; allowed: everything but a slab or a corner (the nesting region)
    const int blobs = 1 + (int)(rnd() % 4);
    for (int q = 0; q < blobs; q++) {
      int c[3], r[3]; for (int d = 0; d < 3; d++) { c[d] = (int)(rnd() % (uint32_t)G.n[d]); r[d] = 1 + (int)(rnd() % 5); }
      for (int k = 0; k < G.n[2]; k++) for (int j = 0; j < G.n[1]; j++) for (int i = 0; i < G.n[0]; i++) {
        const double x = (double)(i - c[0]) / r[0], y = (double)(j - c[1]) / r[1], z = (double)(k - c[2]) / r[2];
        if (x * x + y * y + z * z <= 1.0) G.t[(size_t)i + (size_t)G.n[0] * ((size_t)j + (size_t)G.n[1] * (size_t)k)] = 1;
      }
    }
    if (trial % 3 == 0) { const int cutx = (int)(rnd() % (uint32_t)G.n[0]); for (int k = 0; k < G.n[2]; k++) for (int j = 0; j < G.n[1]; j++) for (int i = cutx; i < G.n[0]; i++) if ((i + j) % 3 == 0 || i > cutx + 1) G.ok[(size_t)i + (size_t)G.n[0] * ((size_t)j + (size_t)G.n[1] * (size_t)k)] = 0; }
    for (size_t q = 0; q < nb; q++) if (!G.ok[q]) G.t[q] = 0;                        // (a block takes part only if it is allowed: grids.hip's kk_block_lattice)
    const double min_eff = 0.5 + 0.1 * (rnd() % 5); const int min_width = 1 + (int)(rnd() % 2);
    IBox whole; for (int d = 0; d < 3; d++) { whole.lo[d] = 0; whole.hi[d] = G.n[d] - 1; }
    std::vector<IBox> cl; cluster(G, whole, min_eff, min_width, cl);
    std::vector<IBox> a = cl, b = cl; merge_boxes(a); merge_naive(b);
    if (!same(a, b)) { printf("FAIL: merge differs on clustered boxes (trial %d)\n", trial); return 1; }
    std::vector<int> cover(nb, 0);
    for (const IBox &bx : a) for (int k = bx.lo[2]; k <= bx.hi[2]; k++) for (int j = bx.lo[1]; j <= bx.hi[1]; j++) for (int i = bx.lo[0]; i <= bx.hi[0]; i++) cover[(size_t)i + (size_t)G.n[0] * ((size_t)j + (size_t)G.n[1] * (size_t)k)]++;
    for (size_t q = 0; q < nb; q++) {
      if (cover[q] > 1) { printf("FAIL: boxes overlap (trial %d)\n", trial); return 1; }
      if (G.t[q] && !cover[q]) { printf("FAIL: a tagged block is not covered (trial %d)\n", trial); return 1; }
      if (cover[q] && !G.ok[q]) { printf("FAIL: a box holds a block outside the nesting region (trial %d)\n", trial); return 1; }
    }
    for (const IBox &bx : cl) {                                                      // before the merge: efficient, or too small to cut
      const long vol = (long)(bx.hi[0] - bx.lo[0] + 1) * (bx.hi[1] - bx.lo[1] + 1) * (bx.hi[2] - bx.lo[2] + 1);
      const long nt = count_tags(G, bx);
      int longest = 0; for (int d = 0; d < 3; d++) longest = std::max(longest, bx.hi[d] - bx.lo[d] + 1);
      if ((double)nt < min_eff * (double)vol && longest >= 2 * min_width) { printf("FAIL: an inefficient box that could be cut was accepted (trial %d: %ld of %ld, longest side %d)\n", trial, nt, vol, longest); return 1; }
    }
    cases++;
  }
  printf("OK %d cases\n", cases);
  return 0;
}
