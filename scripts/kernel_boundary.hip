// What does a kernel boundary cost on this chip, and do the boundaries of different hardware queues overlap?  (dev tool)
// Every stream runs a chain of N dependent kernels; each kernel writes `bytes` of its own buffer (one block per 64 KB) and
// is nothing else.  Reported: microseconds per kernel of a chain, for 1 ... 8 streams side by side, streams from
// hipStreamCreateWithFlags (the runtime's pool of hardware queues) and streams that own a queue (hipExtStreamCreateWithCUMask
// with every CU), plain launches and one captured graph per stream.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/kernel_boundary scripts/kernel_boundary.hip && /tmp/kernel_boundary
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void write_kernel(uint4* p, size_t n16, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const bool with_own = argc > 1; // (creating and destroying streams that own a queue takes seconds once a few dozen have existed: off by default)
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, kChain = 200;
  printf("%s, %d CUs; chains of %d dependent kernels per stream; us per kernel of a chain\n", prop.gcnArchName, n_cu, kChain);
  const size_t sizes[] = {0, 1 << 20, 16 << 20};
  for (int own = 0; own < (with_own ? 2 : 1); own++)
    for (int graph = 0; graph < 4; graph++) {
      const int per_graph = graph == 1 ? kChain : graph == 2 ? 5 : graph == 3 ? 10 : 0; // kernels per captured graph
      printf("\n## %s streams, %s\n%-12s", own ? "own-queue" : "pooled", graph == 0 ? "plain launches" : graph == 1 ? "one captured graph of the whole chain per stream" : graph == 2 ? "captured graphs of 5 kernels, launched chain/5 times" : "captured graphs of 10 kernels, launched chain/10 times", "bytes/kernel");
      for (int ns : {1, 2, 3, 5, 8}) printf("  %d stream%s", ns, ns > 1 ? "s" : " ");
      printf("\n");
      for (size_t bytes : sizes) {
        printf("%-12zu", bytes);
        for (int ns : {1, 2, 3, 5, 8}) {
          std::vector<hipStream_t> st(ns);
          std::vector<uint4*> buf(ns);
          std::vector<hipGraphExec_t> ge(ns, nullptr);
          const std::vector<uint32_t> every((size_t)(n_cu + 31) / 32, 0xffffffffu);
          for (int s = 0; s < ns; s++) {
            if (own) hipExtStreamCreateWithCUMask(&st[s], (uint32_t)every.size(), every.data());
            else hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
            hipMalloc(&buf[s], bytes ? bytes : 16);
          }
          const size_t n16 = bytes / 16;
          const unsigned blocks = bytes ? (unsigned)((bytes + (64 << 10) - 1) / (64 << 10)) : 1u;
          auto chain = [&](int s, hipStream_t q, int n_k = kChain) {
            for (int k = 0; k < n_k; k++) hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, q, buf[s], n16, (uint32_t)k);
          };
          if (graph) {
            hipStream_t cap;
            hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
            for (int s = 0; s < ns; s++) {
              hipGraph_t g;
              hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
              chain(s, cap, per_graph);
              hipStreamEndCapture(cap, &g);
              hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
              hipGraphDestroy(g);
            }
            hipStreamDestroy(cap);
          }
          auto run = [&]() {
            if (!graph)
              for (int s = 0; s < ns; s++) chain(s, st[s]);
            else
              for (int r = 0; r < kChain / per_graph; r++) // round robin over the streams, as the trainer's enqueue loop
                for (int s = 0; s < ns; s++) hipGraphLaunch(ge[s], st[s]);
            for (int s = 0; s < ns; s++) hipStreamSynchronize(st[s]);
          };
          run(); // warm-up
          const double t0 = now();
          run();
          const double dt = now() - t0;
          printf("  %9.2f", dt / kChain * 1e6);
          fflush(stdout);
          for (int s = 0; s < ns; s++) {
            if (ge[s]) hipGraphExecDestroy(ge[s]);
            hipFree(buf[s]);
            hipStreamDestroy(st[s]);
          }
        }
        printf("\n");
      }
    }
  return 0;
}
