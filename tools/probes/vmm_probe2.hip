// variants of mapping several chunks into one reserved range (vmm_probe.hip: the second hipMemSetAccess failed with "invalid argument")
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define TRY(x) ([&]() { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("   FAILED %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); return false; } return true; })()
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double *p, size_t n, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main() {
  int dev = 0; (void)hipSetDevice(dev);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t C = (size_t)1 << 30;
  for (int variant = 0; variant < 4; variant++) {
    printf("variant %d: %s\n", variant, variant == 0 ? "equal 1 GB chunks, set access per chunk" : variant == 1 ? "equal chunks, set access on [base, mapped) each time"
           : variant == 2 ? "2 MB-aligned reserve (alignment argument 2 MB), per chunk" : "one reserve per chunk at a fixed address behind the previous one");
    void *base = nullptr; const size_t VA = (size_t)64 << 30;
    if (variant < 3) { if (!TRY(hipMemAddressReserve(&base, VA, variant == 2 ? (size_t)2 << 20 : 0, nullptr, 0))) continue; }
    else { if (!TRY(hipMemAddressReserve(&base, C, 0, nullptr, 0))) continue; }
    std::vector<hipMemGenericAllocationHandle_t> hs; size_t mapped = 0; bool ok = true;
    for (int c = 0; c < 4 && ok; c++) {
      hipMemGenericAllocationHandle_t h; double a = now();
      if (variant == 3 && c > 0) { void *p = nullptr; ok = TRY(hipMemAddressReserve(&p, C, 0, (char *)base + mapped, 0)); if (ok && p != (char *)base + mapped) { printf("   got %p, wanted %p\n", p, (char *)base + mapped); ok = false; } if (!ok) break; }
      ok = TRY(hipMemCreate(&h, C, &prop, 0)); if (!ok) break;
      double b = now();
      ok = TRY(hipMemMap((char *)base + mapped, C, 0, h, 0)); if (!ok) break;
      double cc = now();
      ok = variant == 1 ? TRY(hipMemSetAccess(base, mapped + C, &acc, 1)) : TRY(hipMemSetAccess((char *)base + mapped, C, &acc, 1));
      double d = now();
      printf("   chunk %d: create %.3f ms, map %.3f ms, set access %.3f ms %s\n", c, 1e3 * (b - a), 1e3 * (cc - b), 1e3 * (d - cc), ok ? "" : "(failed)");
      if (ok) { hs.push_back(h); mapped += C; }
    }
    if (mapped) {
      double t0 = now();
      hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double *)base, mapped / 8, 1.0);
      bool k = TRY(hipDeviceSynchronize());
      printf("   fill of %zu MB across the chunks: %s, %.2f ms\n", mapped >> 20, k ? "ok" : "FAILED", 1e3 * (now() - t0));
      TRY(hipMemUnmap(base, mapped));
    }
    for (auto h : hs) TRY(hipMemRelease(h));
    TRY(hipMemAddressFree(base, variant < 3 ? VA : C));
  }
  // the other growth path: what do hipMalloc / hipFree of large blocks cost here?
  for (size_t gb : { 1, 8, 32, 100 }) {
    void *p; double t0 = now(); if (!TRY(hipMalloc(&p, gb << 30))) break; double t1 = now(); TRY(hipFree(p)); double t2 = now();
    printf("hipMalloc of %3zu GB: %.1f ms, hipFree %.1f ms\n", gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
  }
  for (size_t gb : { 100 }) {      // a second time: is the cost in first use of the pages?
    void *p; double t0 = now(); if (!TRY(hipMalloc(&p, gb << 30))) break; double t1 = now(); TRY(hipFree(p)); double t2 = now();
    printf("hipMalloc of %3zu GB again: %.1f ms, hipFree %.1f ms\n", gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
  }
  printf("DONE\n");
  return 0;
}
