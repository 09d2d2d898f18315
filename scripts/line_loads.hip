// line_loads.hip -- does the vector L1 (TCP) merge the loads of ONE 64-byte line that a lane issues back to back?  (dev tool, round 6)
// Every lane issues 8 independent 16-byte loads per iteration:
//   mode 0: at 8 random lines (one 16-byte piece of each)                -> 8 lines per lane and iteration
//   mode 1: 2 random lines, all four 16-byte pieces of each, back to back -> 2 lines per lane and iteration
//   mode 2: as 1, but the four pieces of a line are issued with the other line's in between
// in two flavours of coherence: every lane its own random lines (c = 0) / the 64 lanes of a wave inside one random 4 KiB window (c = 1).
// If the TCP merges, mode 1 makes a quarter of mode 0's L2 requests (rocprofv3 --pmc TCC_REQ_sum) and runs faster per load.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/line_loads scripts/line_loads.hip && /tmp/line_loads [MiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int MODE, int COH>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ tab, uint64_t line_mask, int iters, uint32_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, wave = tid >> 6;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint64_t line[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t key = (uint64_t)(it * 8 + j);
      if (COH == 0) line[j] = mix64(tid * 0x9E3779B97F4A7C15ull + key) & line_mask;
      else line[j] = ((mix64(wave * 0x9E3779B97F4A7C15ull + key) & line_mask) & ~63ull) | (mix64(tid + 77 * key) & 63ull); // a 4 KiB window per wave
    }
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t l;
      int piece;
      if (MODE == 0) { l = line[j]; piece = j & 3; }
      else if (MODE == 1) { l = line[j >> 2]; piece = j & 3; }
      else { l = line[j & 1]; piece = j >> 1; }
      v[j] = tab[l * 4 + piece];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].w;
  }
  if (acc == 0x12345678u) out[tid & 1023] = acc;
}

template <int MODE, int COH>
void run(const void* tab, size_t bytes, uint32_t* out) {
  const uint64_t line_mask = bytes / 64 - 1;
  const int blocks = 256 * 16, iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, COH>), dim3(blocks), dim3(256), 0, 0, (const uint4*)tab, line_mask, 4, out);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((k<MODE, COH>), dim3(blocks), dim3(256), 0, 0, (const uint4*)tab, line_mask, iters, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double loads = (double)blocks * 256 * iters * 8;
  printf("table %5.0f MiB  mode %d coherent %d: %8.3f ms  %7.2f G loads/s  %7.2f G lines/s\n", bytes / 1048576.0, MODE, COH, ms, loads / ms / 1e6,
         loads / (MODE == 0 ? 1 : 4) / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 64;
  void* tab;
  uint32_t* out;
  if (hipMalloc(&tab, mib << 20) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
  hipMemset(tab, 1, mib << 20);
  for (size_t s : {(size_t)4, (size_t)16, mib}) {
    run<0, 0>(tab, s << 20, out); run<1, 0>(tab, s << 20, out); run<2, 0>(tab, s << 20, out);
    run<0, 1>(tab, s << 20, out); run<1, 1>(tab, s << 20, out); run<2, 1>(tab, s << 20, out);
  }
  hipDeviceSynchronize();
  return 0;
}
