// lds_occupancy.hip -- how many 256-thread blocks of a given LDS footprint really share a CU (dev tool; round 6).  The backward tile kernel's
// block takes 80,640 bytes of dynamic LDS and 320 of static: two fill a CU's 160 KB.  A build with one more 4-byte __shared__ word ran at
// 0.49 ms per step for 0.34.  What the runtime's calculator says (hipOccupancyMaxActiveBlocksPerMultiprocessor) and what the hardware does
// (blocks count themselves per CU while they spin) for static = 320 / 324 / 336 bytes and a range of dynamic sizes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_occupancy scripts/lds_occupancy.hip && /tmp/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int EXTRA>
__global__ __launch_bounds__(256) void k(int* per_cu, int* max_seen, float* out) {
  extern __shared__ float lds[];
  __shared__ int st[80 + EXTRA];
  if (threadIdx.x < 80 + EXTRA) st[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x] = 1.0f;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned id = __builtin_amdgcn_s_getreg((16 << 11) | 4); // HW_ID, 16 bits: wave, simd, pipe, cu [11:8], sh [12], se [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned cu = xcc * 256u + ((id >> 8) & 0xffu); // (cu, sh, se) within the XCD
    const int now = atomicAdd(per_cu + cu, 1) + 1;
    atomicMax(max_seen, now);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20000ull) __builtin_amdgcn_s_sleep(16); // 200 us at 100 MHz
    atomicMax(max_seen, atomicAdd(per_cu + cu, 0));
    atomicSub(per_cu + cu, 1);
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255] + st[threadIdx.x % (80 + EXTRA)];
}

template <int EXTRA>
void q(int dyn, int* per_cu, int* max_seen, float* out) {
  int n = -1;
  hipFuncSetAttribute((const void*)k<EXTRA>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<EXTRA>, 256, dyn);
  hipMemset(per_cu, 0, 4 * 2048);
  hipMemset(max_seen, 0, 4);
  hipLaunchKernelGGL(k<EXTRA>, dim3(1024), dim3(256), dyn, 0, per_cu, max_seen, out);
  int seen = 0;
  hipMemcpy(&seen, max_seen, 4, hipMemcpyDeviceToHost);
  printf("static %3d B + dynamic %5d B = %5d: calculator %d blocks per CU, measured %d\n", (80 + EXTRA) * 4, dyn, (80 + EXTRA) * 4 + dyn, n, seen);
}

int main() {
  int *per_cu, *max_seen;
  float* out;
  hipMalloc(&per_cu, 4 * 2048);
  hipMalloc(&max_seen, 4);
  hipMalloc(&out, 4 * 1024 * 256);
  for (int dyn : {40000, 54000, 80000, 80640, 81000, 81280, 81600}) {
    q<0>(dyn, per_cu, max_seen, out);
    q<1>(dyn, per_cu, max_seen, out);
    q<4>(dyn, per_cu, max_seen, out);
  }
  return 0;
}
