// split_colour_probe.hip -- EXPERIMENT (round 5), not part of the library: what would one red-black Gauss-Seidel colour pass of the finest MAC level cost if
// phi, rhs and rho were stored SPLIT BY COLOUR (red cells and black cells in separate arrays)?  The library's pass (kk_cc_gsrb_rho_pair, interleaved layout)
// moves 572 MB per pass at 256^3 -- whole lines of phi, rhs and rho although half of phi / rhs is the other colour -- in 0.1135 ms (5.0 TB/s).  Split storage
// moves 402 MB (phi_b 67 + phi_r 67 r + 67 w + rhs_r 67 + rho 134).  This probe times such a pass on synthetic data.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/split_colour_probe.hip -o gpurun_out/split_probe && gpurun_out/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static __device__ __forceinline__ double lane_prev(double v) { return __shfl_up(v, 1); }
static __device__ __forceinline__ double lane_next(double v) { return __shfl_down(v, 1); }
struct Lev { int n, PXH, PY; long sy, sz; double hi2; };
// index of half-row entry ih of row (j,k) in a split array with one ghost row / plane and 8 entries of padding in front
static __device__ __forceinline__ long idx(const Lev &L, int ih, int j, int k) { return (long)(ih + 8) + L.sy * (j + 1) + L.sz * (k + 1); }
// one pass over the cells of colour `col` (0: (i+j+k) even).  own / oth: the arrays of this colour / of the other one.
// a thread owns two consecutive entries (ih, ih+1) of a row; 64 lanes = 128 entries = a whole row of 256 cells
template <int ROWS, int XCD> __global__ void __launch_bounds__(64 * ROWS) k_pass(Lev L, double *__restrict__ phi_own, const double *__restrict__ phi_oth, const double *__restrict__ rhs_own,
                                               const double *__restrict__ rho_own, const double *__restrict__ rho_oth, int col) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {        // XCD-aware order: the workgroups an XCD receives (every 8th) form a contiguous piece of the tile sequence
    const int gx = gridDim.x, N = gx * (int)gridDim.y, id = (int)blockIdx.x + gx * (int)blockIdx.y;
    const int q = N >> 3, r = N & 7, x = id & 7, slot = id >> 3;
    const int Lq = (x < r) ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
    bx = Lq % gx; by = Lq / gx;
  }
  const int lane = threadIdx.x, j = bx * ROWS + threadIdx.y, k = by;
  const int ih = 2 * lane;
  const int p = (j + k + col) & 1;                 // parity of i of this row's cells of the colour: i = 2 ih + p
  const long c = idx(L, ih, j, k);
  const double2 po = *(const double2 *)(phi_own + c), rh = *(const double2 *)(rhs_own + c), ro = *(const double2 *)(rho_own + c);
  const double2 px = *(const double2 *)(phi_oth + c), rx = *(const double2 *)(rho_oth + c);
  const double2 pym = *(const double2 *)(phi_oth + c - L.sy), pyp = *(const double2 *)(phi_oth + c + L.sy), pzm = *(const double2 *)(phi_oth + c - L.sz), pzp = *(const double2 *)(phi_oth + c + L.sz);
  const double2 rym = *(const double2 *)(rho_oth + c - L.sy), ryp = *(const double2 *)(rho_oth + c + L.sy), rzm = *(const double2 *)(rho_oth + c - L.sz), rzp = *(const double2 *)(rho_oth + c + L.sz);
  // x neighbours of entry A (ih) and B (ih+1): p = 0: (oth[ih-1], oth[ih]) and (oth[ih], oth[ih+1]);  p = 1: (oth[ih], oth[ih+1]) and (oth[ih+1], oth[ih+2])
  const double pl = lane_prev(px.y), pr = lane_next(px.x), rl = lane_prev(rx.y), rr = lane_next(rx.x);
  const double pAm = p ? px.x : pl, pAp = p ? px.y : px.x, pBm = p ? px.y : px.x, pBp = p ? pr : px.y;
  const double rAm = p ? rx.x : rl, rAp = p ? rx.y : rx.x, rBm = p ? rx.y : rx.x, rBp = p ? rr : rx.y;
  double2 out;
  {
    const double p0 = po.x, r0 = ro.x;
    const double bxm = 2.0 / (r0 + rAm), bxp = 2.0 / (rAp + r0), bym = 2.0 / (r0 + rym.x), byp = 2.0 / (ryp.x + r0), bzm = 2.0 / (r0 + rzm.x), bzp = 2.0 / (rzp.x + r0);
    const double ax = (bxp * (p0 - pAp) + bxm * (p0 - pAm)) * L.hi2, ay = (byp * (p0 - pyp.x) + bym * (p0 - pym.x)) * L.hi2, az = (bzp * (p0 - pzp.x) + bzm * (p0 - pzm.x)) * L.hi2;
    const double Ap = ax + ay + az, dg = (bxp + bxm) * L.hi2 + (byp + bym) * L.hi2 + (bzp + bzm) * L.hi2;
    out.x = p0 + (rh.x - Ap) / dg;
  }
  {
    const double p0 = po.y, r0 = ro.y;
    const double bxm = 2.0 / (r0 + rBm), bxp = 2.0 / (rBp + r0), bym = 2.0 / (r0 + rym.y), byp = 2.0 / (ryp.y + r0), bzm = 2.0 / (r0 + rzm.y), bzp = 2.0 / (rzp.y + r0);
    const double ax = (bxp * (p0 - pBp) + bxm * (p0 - pBm)) * L.hi2, ay = (byp * (p0 - pyp.y) + bym * (p0 - pym.y)) * L.hi2, az = (bzp * (p0 - pzp.y) + bzm * (p0 - pzm.y)) * L.hi2;
    const double Ap = ax + ay + az, dg = (bxp + bxm) * L.hi2 + (byp + bym) * L.hi2 + (bzp + bzm) * L.hi2;
    out.y = p0 + (rh.y - Ap) / dg;
  }
  *(double2 *)(phi_own + c) = out;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 200;
  Lev L; L.n = n; L.PXH = n / 2 + 16; L.PY = n + 2; L.sy = L.PXH; L.sz = (long)L.PXH * L.PY; L.hi2 = (double)n * n;
  const size_t tot = (size_t)L.sz * (n + 2);
  std::vector<double> h(tot);
  for (size_t q = 0; q < tot; q++) h[q] = 1.0 + 0.001 * (double)((q * 2654435761u) % 1000);
  double *a[6];
  for (int q = 0; q < 6; q++) { CK(hipMalloc(&a[q], tot * sizeof(double))); CK(hipMemcpy(a[q], h.data(), tot * sizeof(double), hipMemcpyHostToDevice)); }
  // a[0] phi_r, a[1] phi_b, a[2] rhs_r, a[3] rhs_b, a[4] rho_r, a[5] rho_b
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (n != 256) { printf("this probe is written for n = 256 (64 lanes x 2 entries x 2 colours per row)\n"); return 1; }
  #define RUN(ROWS, XCD) { const dim3 g(n / ROWS, n), b(64, ROWS);                                                                                        \
    for (int w = 0; w < 5; w++) { hipLaunchKernelGGL((k_pass<ROWS, XCD>), g, b, 0, 0, L, a[0], a[1], a[2], a[4], a[5], 0); hipLaunchKernelGGL((k_pass<ROWS, XCD>), g, b, 0, 0, L, a[1], a[0], a[3], a[5], a[4], 1); } \
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));                                                                                                   \
    for (int r = 0; r < reps; r++) { hipLaunchKernelGGL((k_pass<ROWS, XCD>), g, b, 0, 0, L, a[0], a[1], a[2], a[4], a[5], 0); hipLaunchKernelGGL((k_pass<ROWS, XCD>), g, b, 0, 0, L, a[1], a[0], a[3], a[5], a[4], 1); } \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                  \
    const double per = ms / (2.0 * reps);                                                                                                                  \
    printf("rows %d xcd %d: %.4f ms per colour pass -> %.2f TB/s on 402 MB   (the library's interleaved pass: 0.1135 ms, 572 MB)\n", ROWS, XCD, per, 402.0e6 / (per * 1e-3) / 1e12); }
  RUN(4, 0) RUN(4, 1) RUN(8, 0) RUN(8, 1) RUN(2, 1) RUN(16, 1)
  return 0;
}
