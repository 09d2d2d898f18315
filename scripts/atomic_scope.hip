// atomic_scope.hip -- does the scope of a no-return f32 add change where it is performed (dev tool; round 6)?  The table-gradient adds of
// the backward tile kernel are agent-scope global_atomic_add_f32: the XCDs' L2s are not coherent with each other, so such an add is
// forwarded to the memory side (TCC_EA0_ATOMIC), which serves ~21 G 64-byte requests/s whatever they carry (atomic_rate.hip).  If a
// NARROWER scope lets an XCD's L2 perform the add itself, a gradient table per XCD (8 copies, summed by the Adam pass) would trade the
// request rate for sweep bandwidth.  Shape: 16 x 16 B per wave-instruction (what the kernel issues), random places in a 36 MB table;
// `own` = 1: every XCD adds into its own copy (index = XCC_ID), 0: all into one.  Checks the grand total afterwards.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_scope scripts/atomic_scope.hip && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int SCOPE>
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ tab, size_t copy_floats, uint32_t n_groups, int own, int iters) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
  const uint32_t wave = tid >> 6, grp = lane / 4u, in = lane % 4u;
  const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
  float* t = tab + (own ? (size_t)xcc * copy_floats : 0);
  for (int it = 0; it < iters; it++) {
    const uint32_t g = mix32((wave * 64u + grp) * 0x9E3779B9u + (uint32_t)it * 0x85EBCA6Bu) % n_groups;
    __hip_atomic_fetch_add(t + (size_t)g * 4u + in, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}

__global__ void sum_kernel(const float* __restrict__ tab, size_t n, double* out) {
  double a = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) a += (double)tab[i];
  atomicAdd(out, a);
}

template <int SCOPE>
void run(const char* name, float* tab, size_t copy_floats, double* dsum) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int own : {0, 1}) {
    for (int bpc : {2, 8}) {
      const int blocks = 256 * bpc, iters = 256;
      hipMemset(tab, 0, copy_floats * 4 * 8);
      hipMemset(dsum, 0, 8);
      const uint32_t n_groups = (uint32_t)(copy_floats / 4);
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(add_kernel<SCOPE>, dim3(blocks), dim3(256), 0, 0, tab, copy_floats, n_groups, own, iters);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      hipLaunchKernelGGL(sum_kernel, dim3(2048), dim3(256), 0, 0, tab, copy_floats * 8, dsum);
      double got = 0;
      hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost);
      const double want = (double)blocks * 256.0 * iters, req = (double)blocks * 4 * iters * 16;
      printf("%-12s %-16s %3d blocks/CU  %8.2f G req64/s   sum %.0f of %.0f%s\n", name, own ? "a copy per XCD" : "one table", bpc, req / ms / 1e6, got, want,
             got == want ? "" : "   <-- LOST ADDS");
    }
  }
}

int main() {
  const size_t copy_floats = (36u << 20) / 4;
  float* tab;
  double* dsum;
  hipMalloc(&tab, copy_floats * 4 * 8);
  hipMalloc(&dsum, 8);
  run<__HIP_MEMORY_SCOPE_AGENT>("agent", tab, copy_floats, dsum);
  run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", tab, copy_floats, dsum);
  run<__HIP_MEMORY_SCOPE_WAVEFRONT>("wavefront", tab, copy_floats, dsum);
  run<__HIP_MEMORY_SCOPE_SINGLETHREAD>("singlethread", tab, copy_floats, dsum);
  return 0;
}
