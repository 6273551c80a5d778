// Step 0 of the two-waves-per-SIMD restructuring of k_cache_fused: does the matrix-pipe time of one wave hide the
// memory / barrier waits of a co-resident wave of ANOTHER workgroup on gfx950, and what does a [gather -> MFMA] phase
// chain gain when its work is cut into two half-size waves per SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 two_wave_probe.hip -o two_wave_probe && ./two_wave_probe
//
// Every workgroup is 4 waves (one per SIMD).  Dynamic LDS decides how many workgroups share a CU (144 KiB -> 1, 72 KiB
// -> 2).  A "phase" = G rounds of 32 independent random 16-byte loads per lane out of a 64 MiB table (s_waitcnt after each
// round, a few dependent VALU ops per load) followed by M fp32 MFMAs whose A operand is read from LDS, with a workgroup
// barrier every 64 MFMAs (the weight ring's seam).  ROLE: 0 = both parts, 1 = MFMA part only, 2 = gather part only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args { const float4* table; uint32_t mask; float* out; int phases, G, M; int role_by_block; int offset_odd; };

template <bool M16>
__global__ __launch_bounds__(256) void probe(Args a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0f / (float)(i + 1);
  __syncthreads();
  // role of this workgroup: 0 both, 1 MFMA only, 2 gather only
  int role = 0;
  if (a.role_by_block == 1) role = (blockIdx.x & 1) ? 2 : 1;
  if (a.role_by_block == 2) role = 1;
  if (a.role_by_block == 3) role = 2;
  uint32_t s = blockIdx.x * 9781u + threadIdx.x * 6271u + 12345u;
  float vsum = 0.0f;
  f32x16 acc0, acc1;
  f32x4 c0, c1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
  for (int i = 0; i < 4; ++i) { c0[i] = 0.0f; c1[i] = 0.0f; }
  const float bop = 1.0f + lane;
  const bool mfma_first = a.offset_odd && (blockIdx.x & 1);
  for (int ph = 0; ph < a.phases; ++ph) {
    for (int part = 0; part < 2; ++part) {
      const bool do_gather = (part == 0) != mfma_first;
      if (do_gather) {
        if (role == 1) continue;
        for (int g = 0; g < a.G; ++g) {
          float4 v[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            s = s * 1664525u + 1013904223u;
            v[k] = a.table[(s >> 7) & a.mask];
          }
#pragma unroll
          for (int k = 0; k < 32; ++k) vsum = vsum * 0.999f + (v[k].x * v[k].y + v[k].z - v[k].w);
        }
      } else {
        if (role == 2) continue;
        for (int m0 = 0; m0 < a.M; m0 += 64) {
          const float* ap = lds + (m0 & 63) * 64 + lane;
          if (M16) {
#pragma unroll
            for (int m = 0; m < 64; m += 2) {       // 64 x 16x16x4 = half the cycles of 64 x 32x32x2
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m * 64], bop, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m * 64 + 64], bop, c1, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int m = 0; m < 64; m += 2) {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m * 64], bop, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m * 64 + 64], bop, acc1, 0, 0, 0);
            }
          }
          __syncthreads();
        }
      }
    }
  }
  float r = vsum;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i];
  a.out[blockIdx.x * 256 + threadIdx.x] = r + wave;
}

template <bool M16>
static float run(Args a, int blocks, int lds_bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<M16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(probe<M16>, dim3(blocks), dim3(256), lds_bytes, 0, a);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<M16>, dim3(blocks), dim3(256), lds_bytes, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const size_t entries = (64u << 20) / 16;
  float4* table;
  CK(hipMalloc(&table, entries * 16));
  std::vector<float> h(entries * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-5f;
  CK(hipMemcpy(table, h.data(), entries * 16, hipMemcpyHostToDevice));
  float* out;
  CK(hipMalloc(&out, 1024 * 256 * 4));
  const int ONE = 144 * 1024, TWO = 72 * 1024;
  Args a{table, (uint32_t)(entries - 1), out, 8, 2, 512, 0, 0};
  printf("two_wave_probe (gfx950): phase = G rounds x 32 random 16-B loads + M fp32 MFMAs (A from LDS, barrier per 64); us per launch, best of 5\n");
  // A: the present shape -- 256 workgroups, one wave per SIMD, full-size phases
  const float a_full = run<false>(a, 256, ONE);
  Args m = a; m.role_by_block = 2; const float a_mfma = run<false>(m, 256, ONE);
  Args g = a; g.role_by_block = 3; const float a_gath = run<false>(g, 256, ONE);
  printf("A  1 wave/SIMD, 8 x [2 x 32 loads, 512 MFMA 32x32x2]           : both %8.1f   MFMA part alone %8.1f   gather part alone %8.1f\n", a_full, a_mfma, a_gath);
  // B: the same total work as 512 workgroups of half-size phases, two per CU (two waves per SIMD, independent barriers)
  Args b = a; b.G = 1; b.M = 256;
  const float b_full = run<false>(b, 512, TWO);
  Args bo = b; bo.offset_odd = 1; const float b_off = run<false>(bo, 512, TWO);
  printf("B  2 waves/SIMD (2 WG/CU), 8 x [1 x 32 loads, 256 MFMA 32x32x2]: both %8.1f   odd workgroups MFMA-first %8.1f\n", b_full, b_off);
  Args b16 = a; b16.G = 1;
  const float b16_full = run<true>(b16, 512, TWO);
  printf("B' 2 waves/SIMD (2 WG/CU), 8 x [1 x 32 loads, 512 MFMA 16x16x4]: both %8.1f\n", b16_full);
  // C: roles -- even workgroups only the MFMA part, odd workgroups only the gather part, co-resident on every CU
  Args c = a; c.role_by_block = 1;
  const float c_both = run<false>(c, 512, TWO);
  printf("C  2 WG/CU, even = MFMA part only (512 x 8), odd = gather part only (2 x 32 x 8): co-resident %8.1f   (alone: %8.1f / %8.1f, sum %8.1f)\n",
         c_both, a_mfma, a_gath, a_mfma + a_gath);
  const float c16 = run<true>(c, 512, TWO);
  Args m16 = a; m16.role_by_block = 2; const float a_mfma16 = run<true>(m16, 256, ONE);
  printf("C' the same with v_mfma_f32_16x16x4_f32 (512 x 8 of them = half the matrix cycles): co-resident %8.1f   (MFMA alone %8.1f)\n", c16, a_mfma16);
  // D: gather part at two workgroups per CU, full-size each (is the memory system the limit, or the issue / latency of one wave?)
  const float d_g2 = run<false>(g, 512, TWO);
  printf("D  gather part only, 512 WGs x full size (2x the loads of A): %8.1f  (A's gather part alone %8.1f)\n", d_g2, a_gath);
  return 0;
}
