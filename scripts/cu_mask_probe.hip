// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask name on this chip?  (dev tool)  One block per launch slot
// records the XCC / SE / SH / CU it ran on; for a set of mask patterns the program prints how many distinct CUs ran and on
// which XCDs.     hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe scripts/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // spin a little so that every CU of the mask gets a block
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000ull) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xfu) << 16 | (hw & 0xffffu);
}

static void probe(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    printf("%-28s stream create failed\n", name);
    return;
  }
  const int n = 4096;
  uint32_t* d;
  hipMalloc(&d, n * 4);
  hipMemsetAsync(d, 0xff, n * 4, s);
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, s, d);
  std::vector<uint32_t> h(n);
  hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, s);
  hipStreamSynchronize(s);
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  for (uint32_t v : h) {
    const uint32_t xcc = v >> 16, hw = v & 0xffffu;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
  }
  size_t total = 0;
  printf("%-28s", name);
  for (auto& kv : per_xcc) {
    printf(" xcc%u:%zu", kv.first, kv.second.size());
    total += kv.second.size();
  }
  printf("  = %zu CUs\n", total);
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount, words = (n_cu + 31) / 32;
  printf("%s, %d CUs\n", p.gcnArchName, n_cu);
  auto make = [&](auto pred) {
    std::vector<uint32_t> m(words, 0u);
    for (int i = 0; i < n_cu; i++)
      if (pred(i)) m[i >> 5] |= 1u << (i & 31);
    return m;
  };
  probe("all", make([](int) { return true; }));
  probe("i % 2 == 0", make([](int i) { return i % 2 == 0; }));
  probe("i % 8 == 0", make([](int i) { return i % 8 == 0; }));
  probe("i % 8 < 2", make([](int i) { return i % 8 < 2; }));
  probe("i % 8 < 4", make([](int i) { return i % 8 < 4; }));
  probe("i % 5 == 0", make([](int i) { return i % 5 == 0; }));
  probe("i < 32", make([](int i) { return i < 32; }));
  probe("i < 64", make([](int i) { return i < 64; }));
  probe("i < 128", make([](int i) { return i < 128; }));
  probe("32 <= i < 64", make([](int i) { return i >= 32 && i < 64; }));
  probe("i / 32 % 2 == 0", make([](int i) { return i / 32 % 2 == 0; }));
  probe("i == 0", make([](int i) { return i == 0; }));
  probe("i == 1", make([](int i) { return i == 1; }));
  probe("i == 8", make([](int i) { return i == 8; }));
  return 0;
}
