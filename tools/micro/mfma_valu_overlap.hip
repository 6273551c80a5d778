// Does one wave's VALU work run in the shadow of its own MFMAs on gfx950?  Three kernels, one wave per SIMD:
//   M: chain of v_mfma_f32_32x32x2_f32 (two accumulators alternating)       V: 12 fp32 FMAs per iteration
//   B: both, 12 FMAs after every MFMA (sched_barrier pins the order)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.0f; a1[i] = 0.0f; }
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
  float x = seed + threadIdx.x, y = seed * 0.5f;
  const float c1 = 1.0001f * seed, c2 = 0.5f * seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) {
        // volatile asm statements keep their program order (intrinsics without side effects do not: SelectionDAG is
        // free to emit them on either side of a sched_barrier)
        if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      }
      __builtin_amdgcn_sched_barrier(0);       // pin the order: MFMA, 12 FMAs, MFMA, ...
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 3: 512 threads = two waves per SIMD; waves 0-3 run the MFMA loop, waves 4-7 the FMA loop
template <int SPLIT>
__global__ __launch_bounds__(512) void k2(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool mfma_role = SPLIT == 0 ? wave < 4 : (SPLIT == 1 ? (wave & 1) == 0 : (wave & 2) == 0);
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.0f; a1[i] = 0.0f; }
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
  float x = seed + threadIdx.x, y = seed * 0.5f;
  const float c1 = 1.0001f * seed, c2 = 0.5f * seed;
  if (mfma_role) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 48; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 12]) : "v"(c1), "v"(c2));
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SPLIT>
float run2(float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k2<SPLIT>, dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k2<SPLIT>, dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int MODE>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 512 * sizeof(float));
  const int iters = 20000;
  const float m = run<0>(d, iters), v = run<1>(d, iters), b = run<2>(d, iters);
  printf("4 MFMA / iteration: %.3f ms   48 FMA / iteration: %.3f ms   both interleaved: %.3f ms   (sum %.3f, max %.3f)\n", m, v, b, m + v,
         m > v ? m : v);
  printf("two waves per SIMD, four MFMA waves + four FMA waves; MFMA role = wave < 4: %.3f ms, even waves: %.3f ms, (wave & 2) == 0: %.3f ms\n",
         run2<0>(d, iters), run2<1>(d, iters), run2<2>(d, iters));
  return 0;
}
