// grids.hip -- the grid generation that sits in front of the AMR hot path: tag_boxes (src/tag_boxes.f90:17-216) on the device and
// FBoxLib's make_new_grids (src/initialize.f90:247-248, src/regrid.f90:148-149): the per-cell maps on the device, the clustering of the block lattice
// (1/blocking^3 of the cells) on the host.
//
// make_new_grids is not in the reference tree (FBoxLib); the call sites and the probin parameters fix what goes in and out, the
// procedure below is ours:
//   1. tags = tag_boxes(first component of `s`, level) on the valid cells of the level (the reference's thresholds);
//   2. the tags are grown by amr_buf_width cells (probin.template:147-154) and clipped to the nesting region: the cells of the
//      level that keep `nest` cells of the level between themselves and any cell outside it (domain boundaries do not count; a periodic face is no boundary: the maps wrap) --
//      the coarse-fine interpolation of the finer level then always finds its parents on this level;
//   3. Berger-Rigoutsos clustering on the lattice of cluster_blocking_factor^3 blocks: take the bounding box of the tagged
//      blocks; accept it when tagged/total >= cluster_min_eff or it cannot be cut; otherwise cut it at a hole of the tag
//      signature, else at the strongest inflection of the signature's second difference, else in the middle of the longest side,
//      and recurse;
//   4. the boxes are refined by ref_ratio and chopped to max_grid_size.
// Output: boxes of level+1 in its own index space, disjoint, every tagged cell covered, blocking-factor aligned.
#include "vdn_dev.h"
#include "cluster.h"
#include <vector>
#include <algorithm>

__global__ void kk_tag(FV s, unsigned char *tags, int n0, int n1, int d0, int d1, int d2, int rule, double tlo, double thi, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double v = fv_get(s, i, j, k);
  const bool t = rule == 0 ? (v > tlo) : (v > tlo && v < thi);
  tags[(size_t)(i - d0) + (size_t)n0 * ((size_t)(j - d1) + (size_t)n1 * (size_t)(k - d2))] = t ? 1 : 0;
}

// the byte maps of make_new_grids on the device: one byte per cell of the level's domain, x fastest
__global__ void kk_count_bytes(const unsigned char *a, size_t n, unsigned long long *out) {
  unsigned long long c = 0;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) c += a[q];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ void kk_fill_bytes(unsigned char *a, int n0, int n1, Range3 r) {
  THREAD_IJK(r)
  if (in_range) a[(size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * (size_t)k)] = 1;
}
// box dilation (OR) or erosion (AND) by `width` cells in direction d; cells outside the domain do not take part, except across a periodic face, where the
// map wraps (a refined box at a periodic face needs its parents' periodic images underneath it: without the wrap the nesting region ignored them and
// inputs_RayleighTaylor_2d met an improperly nested level at its 31st regrid)
__global__ void kk_sweep_bytes(const unsigned char *in, unsigned char *out, int n0, int n1, int n2, int d, int width, int dilate, int periodic) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), j = (int)(blockIdx.y * blockDim.y + threadIdx.y), k = (int)(blockIdx.z * blockDim.z + threadIdx.z);
  if (i >= n0 || j >= n1 || k >= n2) return;
  const size_t c = (size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * (size_t)k);
  const size_t stride = d == 0 ? 1 : (d == 1 ? (size_t)n0 : (size_t)n0 * n1);
  const int q = d == 0 ? i : (d == 1 ? j : k), nq = d == 0 ? n0 : (d == 1 ? n1 : n2);
  unsigned char v = dilate ? 0 : 1;
  for (int w = -width; w <= width; w++) {
    int ww = w;
    if (q + w < 0 || q + w >= nq) { if (!periodic) continue; ww = ((q + w) % nq + nq) % nq - q; }
    const unsigned char x = in[(long)c + (long)ww * (long)stride];
    if (dilate) v |= x; else v &= x;
  }
  out[c] = v;
}
// per block of blocking^dm cells: ok = every cell inside the nesting region, t = ok and any cell tagged
__global__ void kk_block_lattice(const unsigned char *tags, const unsigned char *inside, unsigned char *t, unsigned char *ok, int n0, int n1, int g0, int g1,
                                 int blocking, int bz, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  bool any = false, all_in = true;
  for (int c = 0; c < bz; c++) for (int b = 0; b < blocking; b++) for (int a = 0; a < blocking; a++) {
    const size_t q = (size_t)(i * blocking + a) + (size_t)n0 * ((size_t)(j * blocking + b) + (size_t)n1 * (size_t)(k * bz + c));
    any = any || tags[q]; all_in = all_in && inside[q];
  }
  t[(size_t)i + (size_t)g0 * ((size_t)j + (size_t)g1 * (size_t)k)] = (any && all_in) ? 1 : 0;
  ok[(size_t)i + (size_t)g0 * ((size_t)j + (size_t)g1 * (size_t)k)] = all_in ? 1 : 0;
}

using namespace vdn_cluster;

// tag_boxes (src/tag_boxes.f90:17-39 driver, 41-127 2-D, 128-216 3-D) of one level: a byte per cell of the level's DOMAIN (x fastest), 1 where
// the first component of s exceeds the level's threshold; cells outside the level's boxes stay 0; all ranks end with the same bitmap.
// Returns a device buffer the caller frees.
static unsigned char *tag_level(const vdn_multifab *s, int lev1) {
  const vdn_box &pd = s->la->pd[s->lev];
  int n[3]; for (int d = 0; d < 3; d++) n[d] = pd.hi[d] - pd.lo[d] + 1;
  const int pt = ctx().prm.prob_type;
  REQUIRE(pt == 1 || pt == 2 || pt == 3, "tag_boxes: unsupported prob_type %d (tag_boxes.f90:212)", pt);
  int rule = 0; double tlo = 0.0, thi = 0.0;
  if (pt == 3) { rule = 1; tlo = 1.2; thi = 1.8; } else { tlo = lev1 == 1 ? 1.01 : (lev1 == 2 ? 1.1 : 1.5); }
  const size_t ncell = (size_t)n[0] * n[1] * n[2];
  unsigned char *d_tags; HIPCHK(hipMalloc((void **)&d_tags, ncell));
  HIPCHK(hipMemsetAsync(d_tags, 0, ncell, ctx().stream));
  for (int b = 0; b < s->nfabs(); b++) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = s->vbox[b].lo[d]; r.hi[d] = s->vbox[b].hi[d]; }
    hipLaunchKernelGGL(kk_tag, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, s->fabs[b], d_tags, n[0], n[1], pd.lo[0], pd.lo[1], pd.lo[2], rule, tlo, thi, r);
  }
  comm_allreduce_max_u8_dev(d_tags, ncell);                // the tags of the other ranks' boxes: every rank sees the same bitmap
  return d_tags;
}
// the tag bitmap alone (the parity test against the restated tag_boxes_3d): tags_host holds one byte per cell of the level's domain
extern "C" int vdn_tag_boxes(const vdn_multifab *s, int lev1, unsigned char *tags_host) {
  VDN_TRY
  REQUIRE(s && tags_host, "vdn_tag_boxes: null argument");
  const vdn_box &pd = s->la->pd[s->lev];
  size_t ncell = 1; for (int d = 0; d < 3; d++) ncell *= (size_t)(pd.hi[d] - pd.lo[d] + 1);
  unsigned char *d_tags = tag_level(s, lev1);
  HIPCHK(hipMemcpyAsync(tags_host, d_tags, ncell, hipMemcpyDeviceToHost, ctx().stream));
  HIPCHK(hipStreamSynchronize(ctx().stream));
  HIPCHK(hipFree(d_tags));
  VDN_CATCH
}

// s: the state of ONE level (valid cells of its local boxes; single rank).  boxes_out: boxes of the next finer level in ITS index space
extern "C" int vdn_make_new_grids(const vdn_multifab *s, int lev1, int buf_wid, int nest, double min_eff, int min_width, int blocking,
                                  int max_grid_size, int maxboxes, vdn_box *boxes_out, int *nboxes_out, long *ntagged) {
  VDN_TRY
  REQUIRE(s && boxes_out && nboxes_out, "vdn_make_new_grids: null argument");
  REQUIRE(blocking >= 1 && min_width >= 1 && max_grid_size >= 2 * blocking, "vdn_make_new_grids: bad clustering parameters");
  const vdn_layout *la = s->la;
  const vdn_box &pd = la->pd[s->lev];
  int n[3]; for (int d = 0; d < 3; d++) n[d] = pd.hi[d] - pd.lo[d] + 1;
  const int dm = ctx().prm.dm;
  for (int d = 0; d < dm; d++) REQUIRE(n[d] % blocking == 0, "vdn_make_new_grids: the domain extent %d is not a multiple of the blocking factor %d", n[d], blocking);
  // 1. tags on the device (tag_boxes.f90:142-210: thresholds by level, prob_type)
  const size_t ncell = (size_t)n[0] * n[1] * n[2];
  unsigned char *d_tags = tag_level(s, lev1);
  // 2. on the device (a 512^3 level is 134 M cells: the host loops of rounds 2-5 took 5 s of a 0.5 s step, tools/probes/regrid_profile_probe.py): count the tags,
  // grow them by buf_wid (box dilation, one direction after the other), inside[] = 1 on the cells of the level -- every box, on any rank --, shrunk by `nest`
  // the same way (cells outside the domain count as inside: a level may touch the domain boundary)
  hipStream_t st = ctx().stream;
  unsigned char *d_in = nullptr, *d_tmp = nullptr; unsigned long long *d_cnt = nullptr;
  HIPCHK(hipMalloc((void **)&d_in, ncell)); HIPCHK(hipMalloc((void **)&d_tmp, ncell)); HIPCHK(hipMalloc((void **)&d_cnt, sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(kk_count_bytes, dim3((unsigned)std::min<size_t>((ncell + 255) / 256, 4096)), dim3(256), 0, st, d_tags, ncell, d_cnt);
  unsigned long long nt = 0;
  HIPCHK(hipMemcpyAsync(&nt, d_cnt, sizeof(nt), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (ntagged) *ntagged = (long)nt;
  *nboxes_out = 0;
  if (nt == 0) { HIPCHK(hipFree(d_tags)); HIPCHK(hipFree(d_in)); HIPCHK(hipFree(d_tmp)); HIPCHK(hipFree(d_cnt)); return 0; }
  Range3 whole_cells; for (int d = 0; d < 3; d++) { whole_cells.lo[d] = 0; whole_cells.hi[d] = n[d] - 1; }
  auto sweep = [&](unsigned char *&a, unsigned char *&tmp, int width, int dilate) {
    if (width <= 0) return;
    for (int d = 0; d < dm; d++) {
      hipLaunchKernelGGL(kk_sweep_bytes, grid_for(whole_cells), dim3(64, 4, 1), 0, st, (const unsigned char *)a, tmp, n[0], n[1], n[2], d, width, dilate, la->pmask[d] ? 1 : 0);
      std::swap(a, tmp);
    }
  };
  HIPCHK(hipMemsetAsync(d_in, 0, ncell, st));
  for (const vdn_box &gb : la->boxes[s->lev]) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = gb.lo[d] - pd.lo[d]; r.hi[d] = gb.hi[d] - pd.lo[d]; }
    hipLaunchKernelGGL(kk_fill_bytes, grid_for(r), dim3(64, 4, 1), 0, st, d_in, n[0], n[1], r);
  }
  sweep(d_tags, d_tmp, buf_wid, 1);
  sweep(d_in, d_tmp, nest, 0);
  // 3. cluster on the lattice of blocks; a block takes part only if it lies in the nesting region as a whole (a tagged cell outside the nesting region
  // cannot tag a block: the block is then not inside as a whole either)
  Lattice G; for (int d = 0; d < 3; d++) G.n[d] = d < dm ? n[d] / blocking : 1;
  const size_t nblk = (size_t)G.n[0] * G.n[1] * G.n[2];
  G.t.assign(nblk, 0); G.ok.assign(nblk, 0);
  const int bz = dm == 3 ? blocking : 1;
  {
    unsigned char *d_lat = nullptr; HIPCHK(hipMalloc((void **)&d_lat, 2 * nblk));
    Range3 rb; for (int d = 0; d < 3; d++) { rb.lo[d] = 0; rb.hi[d] = G.n[d] - 1; }
    hipLaunchKernelGGL(kk_block_lattice, grid_for(rb), dim3(64, 4, 1), 0, st, (const unsigned char *)d_tags, (const unsigned char *)d_in, d_lat, d_lat + nblk,
                       n[0], n[1], G.n[0], G.n[1], blocking, bz, rb);
    HIPCHK(hipMemcpyAsync(G.t.data(), d_lat, nblk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(G.ok.data(), d_lat + nblk, nblk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipFree(d_lat));
  }
  HIPCHK(hipFree(d_tags)); HIPCHK(hipFree(d_in)); HIPCHK(hipFree(d_tmp)); HIPCHK(hipFree(d_cnt));
  std::vector<IBox> cl;
  IBox whole; for (int d = 0; d < 3; d++) { whole.lo[d] = 0; whole.hi[d] = G.n[d] - 1; }
  cluster(G, whole, min_eff, std::max(1, (min_width + blocking - 1) / blocking), cl);
  // 3b. merge neighbours whose union is again a box (cluster.h)
  merge_boxes(cl);
  // 4. refine (blocks -> cells of this level -> cells of the finer level) and chop to max_grid_size
  std::vector<vdn_box> out;
  for (const IBox &b : cl) {
    int flo[3], fhi[3], cnt[3], len[3];
    for (int d = 0; d < 3; d++) {
      const int bl = d < dm ? blocking : 1, rr = d < dm ? 2 : 1;
      flo[d] = (pd.lo[d] + b.lo[d] * bl) * rr; fhi[d] = (pd.lo[d] + (b.hi[d] + 1) * bl) * rr - 1;
      len[d] = fhi[d] - flo[d] + 1;
      cnt[d] = (len[d] + max_grid_size - 1) / max_grid_size;
    }
    for (int c = 0; c < cnt[2]; c++) for (int bb = 0; bb < cnt[1]; bb++) for (int a = 0; a < cnt[0]; a++) {
      const int q[3] = { a, bb, c };
      vdn_box o;
      for (int d = 0; d < 3; d++) {
        // equal pieces, multiples of the (refined) blocking factor
        const int unit = d < dm ? 2 * blocking : 1, units = len[d] / unit;
        const int u0 = (int)((long)units * q[d] / cnt[d]), u1 = (int)((long)units * (q[d] + 1) / cnt[d]);
        o.lo[d] = flo[d] + u0 * unit; o.hi[d] = flo[d] + u1 * unit - 1;
      }
      out.push_back(o);
    }
  }
  REQUIRE((int)out.size() <= maxboxes, "vdn_make_new_grids: %d boxes, room for %d", (int)out.size(), maxboxes);
  for (size_t i = 0; i < out.size(); i++) boxes_out[i] = out[i];
  *nboxes_out = (int)out.size();
  VDN_CATCH
}
