// nd_fused_probe.hip -- EXPERIMENT (round 6), not part of the library: would the two pre-smoothing sweeps of a nodal V-cycle run faster as ONE k-march?
// "stage 2 trails stage 1 by one plane, phi' of stage 1 stays on chip, phi read once and phi'' written once".  The library's sweep
// (kk_nd_march_pair<0,4>, mg_nd.hip) is bound by its load instructions (texture addresser 62-80 % busy): 6 sixteen-byte loads per node pair and plane
// (phi rows j-1..j+1, sigma rows j-1..j, rhs), 154 VGPRs, 3 waves per SIMD, 0.133 ms at 257^3.  Here:
//   k_sweep   the library's march in essence (a thread owns a node pair of one row, 64 x 4 threads own 62 pairs x 4 rows, DPP for the x-neighbours)
//   k_fused   two damped-Jacobi sweeps in one march: stage 1 as above on plane k; its phi' pair goes to an LDS ring of four planes (64 x 8 threads, one barrier
//             per plane); stage 2 forms the stencil of plane k-1 from the ring (rows ty-1..ty+1 of planes k-2..k, x-neighbours by DPP), reuses stage 1's sigma
//             and rhs of that plane from registers, and stores phi''.  A workgroup owns 60 pairs x 6 rows (stage 2 needs stage 1 on one more lane / row
//             each side): 6 loads per pair and plane for TWO sweeps at 70 % tile efficiency = 8.6 per owned pair against 2 x 6 / 0.97 = 12.4.
// Same arithmetic (nd_stencil: the 21-point isotropic form), the fused result is compared bit for bit with two k_sweep launches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/nd_fused_probe.hip -o gpurun_out/nd_fused_probe && gpurun_out/nd_fused_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define DEVI __device__ __forceinline__
DEVI double lane_prev(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
DEVI double lane_next(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
DEVI double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
struct Lev { int nx, ny, nz, PX, PY; long sy, sz; double w0, w3, w5, w6, w7; };
// node (i, j, k), i in [-4, nx + 3], j, k in [-2, n + 1]
DEVI long nidx(const Lev &L, int i, int j, int k) { return (long)(i + 16) + L.sy * (long)(j + 2) + L.sz * (long)(k + 2); }
// the isotropic branch of nd_stencil (mg_nd.hip): differences first, grouped by neighbour type
DEVI void nd_stencil(const Lev &L, const double p[3][3][3], const double sg[2][2][2], double &Kp, double &diag) {
  double cz[2][2], cy[2][2], cx[2][2];
  #pragma unroll
  for (int b = 0; b < 2; b++)
    #pragma unroll
    for (int a = 0; a < 2; a++) { cz[b][a] = sg[0][b][a] + sg[1][b][a]; cy[b][a] = sg[b][0][a] + sg[b][1][a]; cx[b][a] = sg[b][a][0] + sg[b][a][1]; }
  const double S8 = (cz[0][0] + cz[0][1]) + (cz[1][0] + cz[1][1]);
  const double p0 = p[1][1][1];
  #define D(c, b, a) (p[c][b][a] - p0)
  double A7 = sg[0][0][0] * D(0, 0, 0);
  A7 = fma(sg[0][0][1], D(0, 0, 2), A7); A7 = fma(sg[0][1][0], D(0, 2, 0), A7); A7 = fma(sg[0][1][1], D(0, 2, 2), A7);
  A7 = fma(sg[1][0][0], D(2, 0, 0), A7); A7 = fma(sg[1][0][1], D(2, 0, 2), A7); A7 = fma(sg[1][1][0], D(2, 2, 0), A7); A7 = fma(sg[1][1][1], D(2, 2, 2), A7);
  double A3 = cz[0][0] * D(1, 0, 0); A3 = fma(cz[0][1], D(1, 0, 2), A3); A3 = fma(cz[1][0], D(1, 2, 0), A3); A3 = fma(cz[1][1], D(1, 2, 2), A3);
  double A5 = cy[0][0] * D(0, 1, 0); A5 = fma(cy[0][1], D(0, 1, 2), A5); A5 = fma(cy[1][0], D(2, 1, 0), A5); A5 = fma(cy[1][1], D(2, 1, 2), A5);
  double A6 = cx[0][0] * D(0, 0, 1); A6 = fma(cx[0][1], D(0, 2, 1), A6); A6 = fma(cx[1][0], D(2, 0, 1), A6); A6 = fma(cx[1][1], D(2, 2, 1), A6);
  #undef D
  double acc = L.w3 * A3;
  acc = fma(L.w5, A5, acc); acc = fma(L.w6, A6, acc); acc = fma(L.w7, A7, acc);
  Kp = acc; diag = L.w0 * S8;
}
__device__ double g_sink[256];
#define LOADP(pl, off) { _Pragma("unroll") for (int b = 0; b < 3; b++) { const double2 v = ld2(phi + (off) + (b - 1) * sy); q[pl][b][1] = v.x; q[pl][b][2] = v.y; } }
#define EXCHP(pl) { _Pragma("unroll") for (int b = 0; b < 3; b++) { q[pl][b][0] = lane_prev(q[pl][b][2]); q[pl][b][3] = lane_next(q[pl][b][1]); } }
#define LOADS(dk, off) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) { const double2 v = ld2(sig + (off) + (dj - 1) * sy); sg[dk][dj][1] = v.x; sg[dk][dj][2] = v.y; } }
#define EXCHS(dk) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) sg[dk][dj][0] = lane_prev(sg[dk][dj][2]); }
#define SPLIT(pa, pb, sa, sb, Q, S)                                                                                                           \
  double pa[3][3][3], pb[3][3][3], sa[2][2][2], sb[2][2][2];                                                                                   \
  _Pragma("unroll") for (int pl = 0; pl < 3; pl++) _Pragma("unroll") for (int b = 0; b < 3; b++) _Pragma("unroll") for (int a = 0; a < 3; a++) { pa[pl][b][a] = Q[pl][b][a]; pb[pl][b][a] = Q[pl][b][a + 1]; } \
  _Pragma("unroll") for (int dk = 0; dk < 2; dk++) _Pragma("unroll") for (int dj = 0; dj < 2; dj++) _Pragma("unroll") for (int a = 0; a < 2; a++) { sa[dk][dj][a] = S[dk][dj][a]; sb[dk][dj][a] = S[dk][dj][a + 1]; }

// ---- the library's sweep in essence: 64 x 4 threads, 62 pairs x 4 rows, k-chunks --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sweep(Lev L, const double *__restrict__ phi, double *__restrict__ out, const double *__restrict__ sig, const double *__restrict__ rhsv, double omega, int kchunk) {
  const int lane = threadIdx.x;
  const int ia = 2 * ((int)blockIdx.x * 62 + lane - 1), j = (int)blockIdx.y * 4 + (int)threadIdx.y;
  const int k0 = (int)blockIdx.z * kchunk, k1 = min(k0 + kchunk - 1, L.nz - 1);
  const bool act = lane >= 1 && lane <= 62 && ia < L.nx && j < L.ny;
  const int iac = min(ia, L.nx + 2), jc = min(j, L.ny);
  const long sy = L.sy, sz = L.sz;
  long c = nidx(L, iac, jc, k0);
  double *op = act ? out + c : g_sink + 2 * lane;
  const long ostep = act ? sz : 0;
  double q[3][3][4], sg[2][2][3];
  LOADP(0, c - sz) LOADP(1, c) LOADS(0, c - sz)
  EXCHP(0) EXCHP(1) EXCHS(0)
  for (int k = k0; k <= k1; k++, c += sz, op += ostep) {
    LOADP(2, c + sz) LOADS(1, c)
    const double2 rhs = ld2(rhsv + c);
    EXCHP(2) EXCHS(1)
    SPLIT(pa, pb, sa, sb, q, sg)
    double KpA, dgA, KpB, dgB;
    nd_stencil(L, pa, sa, KpA, dgA); nd_stencil(L, pb, sb, KpB, dgB);
    double2 o; o.x = q[1][1][1]; o.y = q[1][1][2];
    if (dgA != 0.0) o.x = o.x + omega * ((rhs.x - KpA) / dgA);
    if (dgB != 0.0) o.y = o.y + omega * ((rhs.y - KpB) / dgB);
    *reinterpret_cast<double2 *>(op) = o;
    #pragma unroll
    for (int b = 0; b < 3; b++)
      #pragma unroll
      for (int a = 0; a < 4; a++) { q[0][b][a] = q[1][b][a]; q[1][b][a] = q[2][b][a]; }
    #pragma unroll
    for (int dj = 0; dj < 2; dj++)
      #pragma unroll
      for (int a = 0; a < 3; a++) sg[0][dj][a] = sg[1][dj][a];
  }
}

// ---- two sweeps in one march ---------------------------------------------------------------------------------------------------------------------------------
constexpr int FR = 8;            // rows of threads per workgroup: FR - 2 owned
__global__ void __launch_bounds__(64 * FR) k_fused(Lev L, const double *__restrict__ phi, double *__restrict__ out, const double *__restrict__ sig, const double *__restrict__ rhsv, double om1, double om2, int kchunk) {
  __shared__ double2 ring[4][FR][64];
  const int lane = threadIdx.x, ty = threadIdx.y;
  const int ia = 2 * ((int)blockIdx.x * 60 + lane - 2), j = (int)blockIdx.y * (FR - 2) + ty - 1;
  const int k0 = (int)blockIdx.z * kchunk, k1 = min(k0 + kchunk - 1, L.nz - 1);
  const bool own = lane >= 2 && lane <= 61 && ty >= 1 && ty <= FR - 2 && ia < L.nx && j < L.ny;
  const int iac = min(max(ia, -4), L.nx + 2), jc = min(max(j, -1), L.ny);
  const int tym = max(ty - 1, 0), typ = min(ty + 1, FR - 1);
  const long sy = L.sy, sz = L.sz;
  // stage 1 starts one plane below the chunk: plane k0 - 1 is needed by stage 2 of plane k0
  long c = nidx(L, iac, jc, k0 - 1);
  double *op = own ? out + c - sz : g_sink + 2 * lane;          // stage 2 writes plane k - 1 in iteration k
  const long ostep = own ? sz : 0;
  double q[3][3][4], sg[2][2][3];
  LOADP(0, c - sz) LOADP(1, c) LOADS(0, c - sz)
  EXCHP(0) EXCHP(1) EXCHS(0)
  double sgo[2][3] = { { 0.0, 0.0, 0.0 }, { 0.0, 0.0, 0.0 } };     // sigma of the plane below stage 2's lower plane
  double2 rhs_prev = make_double2(0.0, 0.0), dg_prev = make_double2(1.0, 1.0);
  for (int k = k0 - 1; k <= k1 + 1; k++, c += sz, op += ostep) {
    // ---- stage 1: phi' on plane k ----
    LOADP(2, c + sz) LOADS(1, c)
    const double2 rhs = ld2(rhsv + c);
    EXCHP(2) EXCHS(1)
    double2 o1, dg1;
    {
      SPLIT(pa, pb, sa, sb, q, sg)
      double KpA, KpB;
      nd_stencil(L, pa, sa, KpA, dg1.x); nd_stencil(L, pb, sb, KpB, dg1.y);
      o1.x = q[1][1][1]; o1.y = q[1][1][2];
      if (dg1.x != 0.0) o1.x = o1.x + om1 * ((rhs.x - KpA) / dg1.x);
      if (dg1.y != 0.0) o1.y = o1.y + om1 * ((rhs.y - KpB) / dg1.y);
    }
    ring[k & 3][ty][lane] = o1;
    __syncthreads();
    // ---- stage 2: phi'' on plane k - 1 from phi' on planes k - 2 .. k (ring), sigma planes k - 2 (sgo) and k - 1 (sg[0]), rhs and diag of plane k - 1 ----
    if (k >= k0 + 1) {
      double r[3][3][4];
      #pragma unroll
      for (int pl = 0; pl < 3; pl++) {
        const int slot = (k - 2 + pl) & 3;
        const double2 vm = ring[slot][tym][lane], v0 = ring[slot][ty][lane], vp = ring[slot][typ][lane];
        r[pl][0][1] = vm.x; r[pl][0][2] = vm.y; r[pl][1][1] = v0.x; r[pl][1][2] = v0.y; r[pl][2][1] = vp.x; r[pl][2][2] = vp.y;
        #pragma unroll
        for (int b = 0; b < 3; b++) { r[pl][b][0] = lane_prev(r[pl][b][2]); r[pl][b][3] = lane_next(r[pl][b][1]); }
      }
      double s2[2][2][3];
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int a = 0; a < 3; a++) { s2[0][dj][a] = sgo[dj][a]; s2[1][dj][a] = sg[0][dj][a]; }
      SPLIT(pa, pb, sa, sb, r, s2)
      double KpA, dA, KpB, dB;
      nd_stencil(L, pa, sa, KpA, dA); nd_stencil(L, pb, sb, KpB, dB);
      double2 o2; o2.x = r[1][1][1]; o2.y = r[1][1][2];
      if (dg_prev.x != 0.0) o2.x = o2.x + om2 * ((rhs_prev.x - KpA) / dg_prev.x);
      if (dg_prev.y != 0.0) o2.y = o2.y + om2 * ((rhs_prev.y - KpB) / dg_prev.y);
      *reinterpret_cast<double2 *>(op) = o2;
    }
    rhs_prev = rhs; dg_prev = dg1;
    #pragma unroll
    for (int dj = 0; dj < 2; dj++)
      #pragma unroll
      for (int a = 0; a < 3; a++) { sgo[dj][a] = sg[0][dj][a]; sg[0][dj][a] = sg[1][dj][a]; }
    #pragma unroll
    for (int b = 0; b < 3; b++)
      #pragma unroll
      for (int a = 0; a < 4; a++) { q[0][b][a] = q[1][b][a]; q[1][b][a] = q[2][b][a]; }
  }
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  Lev L; L.nx = n; L.ny = n; L.nz = n;
  L.PX = ((n + 40 + 15) / 16) * 16; L.PY = n + 4; L.sy = L.PX; L.sz = (long)L.PX * L.PY;
  const long tot = L.sz * (n + 4);
  const double f = 1.0 / (36.0 * (1.0 / n) * (1.0 / n)), F = 3.0 * f;
  L.w0 = 4.0 * F; L.w3 = -2.0 * f - 2.0 * f + f; L.w5 = -2.0 * f + f - 2.0 * f; L.w6 = f - 2.0 * f - 2.0 * f; L.w7 = -F;
  std::vector<double> hphi(tot), hsig(tot), hrhs(tot);
  srand(7);
  for (long i = 0; i < tot; i++) { hphi[i] = rand() / (double)RAND_MAX - 0.5; hsig[i] = 0.1 + rand() / (double)RAND_MAX; hrhs[i] = (rand() / (double)RAND_MAX - 0.5) * 1e3; }
  double *phi, *t1, *t2, *fu, *sig, *rhs;
  for (double **p : { &phi, &t1, &t2, &fu, &sig, &rhs }) CK(hipMalloc((void **)p, tot * sizeof(double)));
  CK(hipMemcpy(phi, hphi.data(), tot * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sig, hsig.data(), tot * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rhs, hrhs.data(), tot * 8, hipMemcpyHostToDevice));
  // t1 carries phi' with the SAME ghost values as phi outside the region the sweep writes, so that two k_sweep launches and the fused march read the same inputs only inside
  CK(hipMemcpy(t1, phi, tot * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(t2, phi, tot * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(fu, phi, tot * 8, hipMemcpyDeviceToDevice));
  const int np = n / 2, kch = argc > 2 ? atoi(argv[2]) : 16;      // planes per k-chunk (the fused march runs two more per chunk)
  const dim3 gs((np + 61) / 62, (n + 3) / 4, (n + kch - 1) / kch), bs(64, 4);
  const dim3 gf((np + 59) / 60, (n + FR - 3) / (FR - 2), (n + kch - 1) / kch), bf(64, FR);
  const double om1 = 1.45, om2 = 0.7;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto body, int reps) { body(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) body(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  const float ms2 = time([&] { hipLaunchKernelGGL(k_sweep, gs, bs, 0, 0, L, phi, t1, sig, rhs, om1, kch); hipLaunchKernelGGL(k_sweep, gs, bs, 0, 0, L, t1, t2, sig, rhs, om2, kch); }, 50);
  const float msf = time([&] { hipLaunchKernelGGL(k_fused, gf, bf, 0, 0, L, phi, fu, sig, rhs, om1, om2, kch); }, 50);
  CK(hipGetLastError());
  printf("n = %d, k-chunks of %d: two sweeps %.4f ms (%.4f each), fused %.4f ms  -> %.2fx\n", n, kch, ms2, ms2 / 2, msf, ms2 / msf);
  // bit comparison well inside the region (two nodes from every side: the single sweeps see phi's ghost values where the fused march sees phi' it computed itself)
  std::vector<double> a(tot), b(tot);
  CK(hipMemcpy(a.data(), t2, tot * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), fu, tot * 8, hipMemcpyDeviceToHost));
  long bad = 0, cnt = 0; double worst = 0.0;
  for (int k = 2; k < n - 2; k++) for (int j = 2; j < n - 2; j++) for (int i = 2; i < n - 2; i++) {
    const long c = (long)(i + 16) + L.sy * (j + 2) + L.sz * (k + 2);
    cnt++;
    if (memcmp(&a[c], &b[c], 8)) { bad++; const double d = a[c] - b[c]; if (d * d > worst) worst = d * d; }
  }
  printf("fused vs two sweeps: %ld of %ld interior nodes differ (max |diff|^2 %.3e)\n", bad, cnt, worst);
  return bad ? 2 : 0;
}
