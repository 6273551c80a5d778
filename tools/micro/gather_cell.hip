// would ONE 32-byte cell record per (point, hashed F = 1 level) out of a big table beat today's FOUR
// x-pair sectors out of the level's 2 MiB hash table?  A cell-record table of a hashed level stores, for every cell
// origin of the N^3 lattice ((N + 1)^3 records), the 8 corner values side by side: 69 MB at N = 128, 550 MB at 256,
// 4.3 GB at 512, 34 GB at 1024 (the MI355X has 288 GB).  Patterns (a "lookup" = one point at one level, 8 corners):
//   P4  today (pair_fetch): 4 wave-loads of 4 bytes per lookup; lanes j and j + 32 read the adjacent entries x / x + 1
//       of one hashed index -> 32 distinct sectors per wave-load, 4 sectors per lookup, table 2 MiB (L2 resident)
//   C1  proposed: 1 wave-load of 16 bytes per lookup; lanes j and j + 32 read the two halves of one random 32-byte
//       record -> 32 distinct sectors per wave-load, 1 sector per lookup, table 64 MiB ... 4 GiB
//   C2  the same record read by ONE lane as two 16-byte loads (64 distinct sectors per wave-load, 2 loads per lookup)
// Indices are uniformly random (the bench's incoherent rays; at N >= 128 consecutive samples of a ray are 4-8 cells
// apart, so image rays do not share records either).
//   hipcc --offload-arch=gfx950 -O3 gather_cell.hip -o gather_cell && ./gather_cell
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ table, uint64_t mask, int iters, float* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  // P4 / C1: lanes j and j + 32 share the random stream (one point on two lanes); C2: every lane its own
  const uint32_t grp = MODE == 2 ? tid : ((tid & ~63u) | (lane & 31));
  const uint32_t sub = lane >> 5;
  uint64_t s = (uint64_t)grp * 0x9E3779B97F4A7C15ull + 12345u;
  float acc = 0.0f;
  for (int i = 0; i < iters; i += 4) {
    float v[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {             // 4 lookups in flight per lane (as 4 levels of a tile)
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t r = s >> 20;
      if (MODE == 0) {                            // 4 corner pairs: hashed index, x-pair on (j, j + 32)
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t h = (uint32_t)(r >> (c * 7)) * 2654435761u;
          const uint32_t idx = ((h << 1) & (uint32_t)mask) | sub;
          a += table[idx];
        }
        v[k2] = a;
      } else if (MODE == 1) {                     // half of a 32-byte record
        const uint64_t rec = r & mask;            // record index
        const f32x4 q = reinterpret_cast<const f32x4*>(table)[rec * 2 + sub];
        v[k2] = q.x + q.y + q.z + q.w;
      } else {                                    // whole record, one lane
        const uint64_t rec = r & mask;
        const f32x4 q0 = reinterpret_cast<const f32x4*>(table)[rec * 2], q1 = reinterpret_cast<const f32x4*>(table)[rec * 2 + 1];
        v[k2] = (q0.x + q0.y + q0.z + q0.w) + (q1.x + q1.y + q1.z + q1.w);
      }
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) acc += v[k2];
  }
  if (acc == 123.456f) out[tid] = acc;
}

template <int MODE>
double run(const char* name, size_t bytes, int blocks, int iters) {
  float* table; float* out;
  if (hipMalloc(&table, bytes) != hipSuccess) { printf("%-40s %8.0f MiB: allocation failed\n", name, bytes / 1048576.0); return 0; }
  (void)hipMemset(table, 0, bytes);
  (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  const uint64_t mask = MODE == 0 ? bytes / 4 - 1 : bytes / 32 - 1;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0, best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  // lookups: P4 / C1 put one point on two lanes
  const double lookups = (double)blocks * 256 * iters / (MODE == 2 ? 1 : 2);
  const double sectors = lookups * (MODE == 0 ? 4 : 1);
  const double g = lookups / best / 1e6;
  printf("%-44s %8.0f MiB  blocks %5d: %7.2f G lookups/s  %7.2f G sectors/s  (%.3f ms)\n", name, bytes / 1048576.0, blocks, g,
         sectors / best / 1e6, best);
  (void)hipFree(table); (void)hipFree(out);
  return g;
}

int main() {
  for (int blocks : {512, 2048}) {
    run<0>("P4 today: 4 x-pair sectors, hashed table", 2u << 20, blocks, 1024);
    run<0>("P4 (8 MiB table, for reference)", 8u << 20, blocks, 1024);
    for (size_t mib : {64, 128, 512, 2048, 4096}) run<1>("C1 one 32-B record on lanes (j, j+32)", mib << 20, blocks, 1024);
    for (size_t mib : {64, 512, 4096}) run<2>("C2 one 32-B record on one lane", mib << 20, blocks, 1024);
  }
  return 0;
}
