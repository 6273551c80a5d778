// Micro-benchmark: sustained rate of random 4-byte / 16-byte gathers out of tables of 2 MB .. 64 MB on gfx950
// (ceiling for the hash-grid lookups).  hipcc --offload-arch=gfx950 -O3 gather_bw.hip -o gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int F>
__global__ void k_gather(const float* __restrict__ table, uint32_t mask, int iters, float* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = tid * 2654435761u + 12345u;
  float acc = 0.0f;
  for (int i = 0; i < iters; i += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s = s * 1664525u + 1013904223u;
      const uint32_t idx = (s >> 8) & mask;
      if constexpr (F == 4) {
        const float4 q = reinterpret_cast<const float4*>(table)[idx];
        v[k] = q.x + q.y + q.z + q.w;
      } else {
        v[k] = table[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc == 123.456f) out[tid] = acc;
}

template <int F>
void run(size_t entries, int blocks, int iters) {
  float* table; float* out;
  hipMalloc(&table, entries * F * sizeof(float));
  hipMemset(table, 0, entries * F * sizeof(float));
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather<F>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(entries - 1), iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 256 * iters;
  printf("F=%d table %6.1f MB blocks %6d: %8.2f G gathers/s  = %7.1f GB/s algorithmic (%.3f ms)\n", F, entries * F * 4 / 1048576.0, blocks,
         n / ms / 1e6, n * F * 4 / ms / 1e6, ms);
  hipFree(table); hipFree(out);
}

int main() {
  for (size_t e : {(size_t)1 << 19, (size_t)1 << 22, (size_t)1 << 24}) {
    for (int blocks : {1024, 4096, 16384}) { run<1>(e, blocks, 256); run<4>(e, blocks, 256); }
  }
  return 0;
}
