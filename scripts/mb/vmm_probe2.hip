// Which sequences of hipMemMap / hipMemSetAccess into one reserved range does this ROCm accept?   args: sizes in MiB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MB = 1u << 20;
  std::vector<size_t> seq;
  for (int i = 1; i < argc; ++i) seq.push_back((size_t)atoll(argv[i]));
  size_t tot = 0; for (size_t m : seq) tot += m;
  void* va = nullptr;
  if (hipMemAddressReserve(&va, (tot + 64) * MB, 2 * MB, nullptr, 0) != hipSuccess) { fprintf(stderr, "reserve failed\n"); return 1; }
  size_t off = 0; bool ok = true;
  fprintf(stderr, "seq:");
  for (size_t m : seq) {
    hipMemGenericAllocationHandle_t h;
    hipError_t e1 = hipMemCreate(&h, m * MB, &prop, 0);
    hipError_t e2 = hipMemMap((char*)va + off, m * MB, 0, h, 0);
    hipError_t e3 = hipMemSetAccess((char*)va + off, m * MB, &acc, 1);
    (void)hipGetLastError();
    if (e1 || e2 || e3 || seq.size() < 12) fprintf(stderr, " %zuMB@%zu[%d %d %d]", m, off / MB, (int)e1, (int)e2, (int)e3);
    off += m * MB;
    if (e1 || e2 || e3) { ok = false; break; }
  }
  if (ok) { hipError_t e = hipMemset(va, 3, off); hipError_t e2 = hipDeviceSynchronize(); fprintf(stderr, " all %zu mapped, memset %d %d", seq.size(), (int)e, (int)e2); }
  fprintf(stderr, "\n");
  return 0;
}
