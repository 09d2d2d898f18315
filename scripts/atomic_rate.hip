// atomic_rate.hip -- what the memory side serves of THIS path's table-gradient adds (dev tool, DESIGN.md section 3, trainer):
// no-return global_atomic_add_f32, one dword per lane, into a 36 MB f32 table (the canonical gradient of the 256^3
// field); the lanes of a wave-instruction form G groups of 64 / G consecutive dwords, every group at its own random
// aligned place.  G = 1: 256 contiguous bytes (the guide's full-rate shape) ... G = 16: sixteen 16-byte table entries
// (F = 4: what the backward tile kernel issues) ... G = 64: every lane its own line.  `active` < 64: only the first
// lanes add (run-merged coarse levels: a few lanes per instruction).  Prints wave-instructions/s, 64-byte requests/s
// (distinct lines per instruction, as issued) and added bytes/s, for 2, 4 and 8 waves per SIMD-quad (blocks per CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate scripts/atomic_rate.hip && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// group_dwords = 64 / G; entries: table size in groups
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ tab, uint32_t n_groups, int group_dwords, int active, int iters) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
  const uint32_t wave = tid >> 6, grp = lane / (uint32_t)group_dwords, in = lane % (uint32_t)group_dwords;
  if ((int)lane >= active) return;
  for (int it = 0; it < iters; it++) {
    const uint32_t g = mix32((wave * 64u + grp) * 0x9E3779B9u + (uint32_t)it * 0x85EBCA6Bu) % n_groups;
    atomicAdd(tab + (size_t)g * (size_t)group_dwords + in, 1.0f);
  }
}

int main() {
  const size_t bytes = 36u << 20;
  float* tab;
  hipMalloc(&tab, bytes);
  hipMemset(tab, 0, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 256;
  printf("%-34s %8s %12s %12s %10s\n", "shape", "blocks/CU", "G instr/s", "G req64/s", "TB/s added");
  for (int bpc : {2, 4, 8}) {
    for (int gd : {64, 32, 16, 8, 4, 1}) {
      for (int active : {64, 16}) {
        if (active < 64 && gd != 4) continue;
        const int blocks = 256 * bpc;
        const uint32_t n_groups = (uint32_t)(bytes / 4 / gd);
        hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, 0, tab, n_groups, gd, active, 8);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, 0, tab, n_groups, gd, active, iters);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double instr = (double)blocks * 4 * iters;
        const int groups = (active + gd - 1) / gd;                      // groups per instruction
        const double req = instr * groups * (gd >= 16 ? gd / 16 : 1);   // 64-byte lines per instruction (aligned groups)
        char name[64];
        snprintf(name, sizeof name, "%d x %d B%s", groups, gd * 4, active < 64 ? " (16 lanes active)" : "");
        printf("%-34s %8d %12.2f %12.2f %10.3f\n", name, bpc, instr / ms / 1e6, req / ms / 1e6, instr * active * 4 / ms / 1e9);
      }
    }
  }
  // one CU's own ceiling: the 16 x 16 B shape from 8 ... 256 blocks (blocks go round the XCDs first: one per CU up to 256)
  printf("%-34s %8s %12s %12s\n", "16 x 16 B, 4 waves per block", "blocks", "G req64/s", "M req64/s per block");
  for (int blocks : {8, 32, 64, 128, 256, 512}) {
    const uint32_t n_groups = (uint32_t)(bytes / 4 / 4);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, 0, tab, n_groups, 4, 64, 8);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, 0, tab, n_groups, 4, 64, 4 * iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double req = (double)blocks * 4 * 4 * iters * 16;
    printf("%-34s %8d %12.2f %12.1f\n", "", blocks, req / ms / 1e6, req / ms / 1e3 / blocks);
  }
  return 0;
}
