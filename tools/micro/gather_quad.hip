// Does it matter WHICH lanes of a wave share a cache line in a random gather?  The texture-address path works through a
// wave-wide load a quad (4 adjacent lanes) at a time; the level-2 lookup of the fused kernels reads 32-byte
// [density | appearance] pairs with the two halves on lanes j and j + 32 (different quads).  Patterns, all out of one
// 8 MiB table of 2^18 32-byte pairs (the size of one hashed level pair is 16 MiB; L2-resident per XCD either way):
//   A  every lane its own random 16-byte entry                         (64 lines per wave-load)
//   B  lanes (2i, 2i+1) read the two halves of one random pair         (32 lines per wave-load, 2 per quad)
//   C  lanes (j, j+32) read the two halves of one random pair          (32 lines per wave-load, 4 per quad)
//   D  lanes (4i .. 4i+3) read the 4 16-byte quarters of one random 64-byte block  (16 lines per wave-load, 1 per quad)
// and for 4-byte entries (2 MiB table): E every lane its own, F lanes (2i, 2i+1) adjacent entries (the x-pair).
//   hipcc --offload-arch=gfx950 -O3 gather_quad.hip -o gather_quad && ./gather_quad
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ table, uint32_t mask, int iters, float* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t grp;       // lanes with the same grp draw the same random block
  uint32_t sub;       // which part of the block
  if (MODE == 0 || MODE == 4) { grp = tid; sub = 0; }
  else if (MODE == 1 || MODE == 5) { grp = tid >> 1; sub = lane & 1; }
  else if (MODE == 2) { grp = (tid & ~63u) | (lane & 31); sub = lane >> 5; }
  else { grp = tid >> 2; sub = lane & 3; }
  uint32_t s = grp * 2654435761u + 12345u;
  float acc = 0.0f;
  for (int i = 0; i < iters; i += 8) {
    float v[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      s = s * 1664525u + 1013904223u;
      const uint32_t r = s >> 8;
      if (MODE <= 3) {
        uint32_t idx;       // in 16-byte units
        if (MODE == 0) idx = r & mask;
        else if (MODE == 3) idx = ((r << 2) & mask) | sub;
        else idx = ((r << 1) & mask) | sub;
        const float4 q = reinterpret_cast<const float4*>(table)[idx];
        v[k2] = q.x + q.y + q.z + q.w;
      } else {
        const uint32_t idx = MODE == 4 ? (r & mask) : (((r << 1) & mask) | sub);
        v[k2] = table[idx];
      }
    }
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) acc += v[k2];
  }
  if (acc == 123.456f) out[tid] = acc;
}

template <int MODE>
void run(const char* name, size_t bytes, int blocks, int iters) {
  float* table; float* out;
  (void)hipMalloc(&table, bytes);
  (void)hipMemset(table, 0, bytes);
  (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  const uint32_t mask = (uint32_t)(bytes / (MODE <= 3 ? 16 : 4) - 1);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = (double)blocks * 256 * iters;
  printf("%-62s blocks %5d (%2d waves/CU): %8.2f G lane-loads/s = %5.2f per clock per CU at 2.4 GHz\n", name, blocks,
         blocks * 4 / 256 > 32 ? 32 : blocks * 4 / 256, n / ms / 1e6, n / ms / 1e6 / 256 / 2.4);
  (void)hipFree(table); (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 2048}) {
    run<0>("A 16 B, every lane its own entry", 8u << 20, blocks, 512);
    run<1>("B 16 B, lanes (2i, 2i+1) = halves of one 32-B pair", 8u << 20, blocks, 512);
    run<2>("C 16 B, lanes (j, j+32) = halves of one 32-B pair", 8u << 20, blocks, 512);
    run<3>("D 16 B, lanes (4i..4i+3) = quarters of one 64-B block", 8u << 20, blocks, 512);
    run<4>("E  4 B, every lane its own entry", 2u << 20, blocks, 512);
    run<5>("F  4 B, lanes (2i, 2i+1) = adjacent entries", 2u << 20, blocks, 512);
  }
  return 0;
}
