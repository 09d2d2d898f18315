// gather_calib.hip -- calibration of the HBM-side counters and the random-gather ceiling for THIS path's
// access pattern: every lane reads one W-byte entry (8 B = one F=4 fp16 table entry, 16 B = a paired load,
// 4 B = an F=2 entry) at a pseudo-random aligned index of a table of T bytes; 8 independent loads in flight
// per lane per iteration.  Known byte count: loads x W useful bytes, loads x line bytes when every load
// misses (T >> caches).  Dev tool; built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_calib scripts/gather_calib.hip && /tmp/gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// mode 2: L1-hit rate (distinct lines per lane inside an 8 KiB per-block window);
// mode 0: every lane its own random index; mode 1: the 64 lanes of a wave share a random 4 KiB window
// (coherent rays: neighbours land in the same few lines)
template <typename V>
__global__ __launch_bounds__(256) void gather_kernel(const V* __restrict__ tab, uint64_t mask, int iters, int mode,
                                                     uint32_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t wave = tid >> 6;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint64_t idx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (mode == 0) {
        idx[k] = mix64(tid * 0x9E3779B97F4A7C15ull + (uint64_t)(it * 8 + k)) & mask;
      } else if (mode == 2) { // L1-resident: every block re-reads its own 8 KiB window, one random line per lane
        const uint64_t in = mix64(tid * 0x9E3779B97F4A7C15ull + (uint64_t)(it * 8 + k)) & (8192 / sizeof(V) - 1);
        idx[k] = (((uint64_t)blockIdx.x * (8192 / sizeof(V))) & mask) | in;
      } else {
        const uint64_t win = mix64(wave * 0x9E3779B97F4A7C15ull + (uint64_t)(it * 8 + k));
        const uint64_t in = mix64(tid + 77 * (it * 8 + k)) & (4096 / sizeof(V) - 1);
        idx[k] = ((win & mask) & ~(uint64_t)(4096 / sizeof(V) - 1)) | in;
      }
    }
    V v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = tab[idx[k]];
#pragma unroll
    for (int k = 0; k < 8; k++) acc ^= ((const uint32_t*)&v[k])[0];
  }
  if (acc == 0x12345678u) out[tid & 1023] = acc; // keep the loads
}

template <typename V>
void run(const void* tab, size_t bytes, int mode, uint32_t* out, const char* name) {
  const uint64_t mask = bytes / sizeof(V) - 1;
  const int blocks = 256 * 16, iters = 64; // 1 Mi lanes x 64 x 8 = 512 Mi loads
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(gather_kernel<V>, dim3(blocks), dim3(256), 0, 0, (const V*)tab, mask, 4, mode, out); // warm
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(gather_kernel<V>, dim3(blocks), dim3(256), 0, 0, (const V*)tab, mask, iters, mode, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double loads = (double)blocks * 256 * iters * 8;
  printf("%-6s W=%2zu B  table %6.0f MiB  mode %d: %8.3f ms  %7.2f G loads/s  useful %7.1f GB/s  (x64 B lines: %7.1f GB/s)\n",
         name, sizeof(V), bytes / 1048576.0, mode, ms, loads / ms / 1e6, loads * sizeof(V) / ms / 1e6, loads * 64 / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t max_bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4096) << 20;
  void* tab;
  uint32_t* out;
  if (hipMalloc(&tab, max_bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 1, max_bytes);
  const size_t sizes_mib[] = {1, 2, 4, 8, 16, 64, 1024, 4096};
  for (size_t s : sizes_mib) {
    if ((s << 20) > max_bytes) continue;
    for (int mode = 0; mode < (s == 64 ? 3 : 2); mode++) {
      run<uint32_t>(tab, s << 20, mode, out, "u32");
      run<uint2>(tab, s << 20, mode, out, "u32x2");
      run<uint4>(tab, s << 20, mode, out, "u32x4");
    }
  }
  hipDeviceSynchronize();
  return 0;
}
