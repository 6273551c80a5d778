// Micro-benchmark: a chain of 64 -> 64 ReLU layers on 32-sample tiles, operands out of LDS as in rc_dev_mlp.h's mlp_layer,
//   (a) v_mfma_f32_32x32x2_f32 (the product's arithmetic: one MFMA of 64 cycles per k-pair and 32-row tile), against
//   (b) "bf16x3": every fp32 operand split exactly into three bf16 pieces (8 + 8 + 8 significand bits, hi + mid + lo == x),
//       the six products whose weight is >= 2^-16 of the full product on v_mfma_f32_32x32x16_bf16 (32 cycles per 16 k),
//       fp32 accumulation: 6 x 32 = 192 cycles for what (a) spends 8 x 64 = 512 on.
// Prints the time per layer-tile of both and their error against an fp64 evaluation of the same chain.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLayers = 2;          // distinct weight sets, cycled
constexpr int kWaves = 8;           // waves per workgroup, one workgroup per CU: two waves per SIMD
constexpr int kK = 64, kM = 64;     // layer shape
constexpr int kSteps = kK / 2;      // fp32 k-steps
constexpr int kBlocks = kK / 16;    // bf16 k-blocks
constexpr int kNT = kM / 32;

__device__ __forceinline__ float relu0(float x) { return fmaxf(x, 0.0f); }   // NOT the inline-asm form of rc_dev_mlp.h: see the note in main()
__host__ __device__ inline int acc_feat(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// (a) fragments: [layer][step][tile][64 lanes] floats.  (b): [layer][block][tile][piece][64 lanes][4 dwords]
__global__ __launch_bounds__(kWaves * 64) void k_f32(const float* __restrict__ wf, const float* __restrict__ x, float* __restrict__ y, int reps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w = lds;                                   // kLayers * kSteps * kNT * 64
  float* act_all = lds + kLayers * kSteps * kNT * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < kLayers * kSteps * kNT * 64; i += kWaves * 64) w[i] = wf[i];
  float* act = act_all + wave * (kSteps * 64) + lane;
  const int tile = blockIdx.x * kWaves + wave;
  for (int s = 0; s < kSteps; ++s) act[s * 64] = x[((size_t)tile * kSteps + s) * 64 + lane];
  __syncthreads();
  f32x16 acc[kNT];
  for (int it = 0; it < reps; ++it) {
    const float* wl = w + (it % kLayers) * (kSteps * kNT * 64) + lane;
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const float b = act[s * 64];
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(s * kNT + t) * 64], b, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(t * 16 + r) * 64] = relu0(acc[t][r]);
  }
  for (int s = 0; s < kSteps; ++s) y[((size_t)tile * kSteps + s) * 64 + lane] = act[s * 64];
}

__device__ __forceinline__ uint32_t pack_hi(float a1, float a0) {   // bf16 (truncated) of a0 in the low half, of a1 in the high half
  return __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
}
__device__ __forceinline__ float top16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    hi[p] = pack_hi(a1, a0);
    const float r0 = a0 - top16(a0), r1 = a1 - top16(a1);
    mid[p] = pack_hi(r1, r0);
    const float l0 = r0 - top16(r0), l1 = r1 - top16(r1);
    lo[p] = pack_hi(l1, l0);
  }
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NPROD>
__global__ __launch_bounds__(kWaves * 64) void k_split(const uint32_t* __restrict__ wf, const float* __restrict__ x, float* __restrict__ y, int reps, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int kWl = kBlocks * kNT * 3 * 64 * 4;   // dwords per layer
  uint32_t* w = reinterpret_cast<uint32_t*>(lds);
  float* act_all = lds + kLayers * kWl;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < kLayers * kWl; i += kWaves * 64) w[i] = wf[i];
  float* act = act_all + wave * (kSteps * 64) + lane;
  const int tile = blockIdx.x * kWaves + wave;
  for (int s = 0; s < kSteps; ++s) act[s * 64] = x[((size_t)tile * kSteps + s) * 64 + lane];
  __syncthreads();
  f32x16 acc[kNT];
  for (int it = 0; it < reps; ++it) {
    if (stagger) {
      // uneven phases: every wave idles a different, changing number of cycles in front of every layer
      const unsigned hsh = (unsigned)(tile * 2654435761u + it * 40503u + stagger * 97u);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (hsh >> 7) % 1500u) __builtin_amdgcn_s_sleep(1);
    }
    const u32x4* wl = reinterpret_cast<const u32x4*>(w + (it % kLayers) * kWl) + lane;
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < kBlocks; ++q) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act[(8 * q + j) * 64];
      u32x4 b1, b2, b3;
      split8(v, b1, b2, b3);
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        const u32x4 a1 = wl[((q * kNT + t) * 3 + 0) * 64], a2 = wl[((q * kNT + t) * 3 + 1) * 64], a3 = wl[((q * kNT + t) * 3 + 2) * 64];
        if (NPROD >= 6) { acc[t] = mfma_bf16(a3, b1, acc[t]); acc[t] = mfma_bf16(a1, b3, acc[t]); acc[t] = mfma_bf16(a2, b2, acc[t]); }
        if (NPROD >= 3) { acc[t] = mfma_bf16(a2, b1, acc[t]); acc[t] = mfma_bf16(a1, b2, acc[t]); }
        acc[t] = mfma_bf16(a1, b1, acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(t * 16 + r) * 64] = relu0(acc[t][r]);
  }
  for (int s = 0; s < kSteps; ++s) y[((size_t)tile * kSteps + s) * 64 + lane] = act[s * 64];
}


// layout probe: D = A x B with A[m][k] = (m == k), B[k][n] = 100 k + n, operands placed by the assumed rule
// (lane = row / column + 32 * (k / 8), element = k % 8); also the exactness of split8
__global__ void k_probe(float* out, float* sp) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  u32x4 a, b;
  for (int p = 0; p < 4; ++p) {
    float a0 = (i == 8 * h + 2 * p) ? 1.0f : 0.0f, a1 = (i == 8 * h + 2 * p + 1) ? 1.0f : 0.0f;
    a[p] = pack_hi(a1, a0);
    float b0 = 100.0f * (8 * h + 2 * p) + i, b1 = 100.0f * (8 * h + 2 * p + 1) + i;
    b[p] = pack_hi(b1, b0);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.0f;
  c = mfma_bf16(a, b, c);
  for (int r = 0; r < 16; ++r) out[r * 64 + lane] = c[r];
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.2345678f * (lane + 1) / (j + 3) * ((j & 1) ? -1.0f : 1.0f);
  u32x4 p1, p2, p3;
  split8(v, p1, p2, p3);
  for (int p = 0; p < 4; ++p) { sp[(lane * 3 + 0) * 4 + p] = __uint_as_float(p1[p]); sp[(lane * 3 + 1) * 4 + p] = __uint_as_float(p2[p]); sp[(lane * 3 + 2) * 4 + p] = __uint_as_float(p3[p]); }
}

static uint16_t top_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float top_val(float f) { uint32_t u; memcpy(&u, &f, 4); u &= 0xffff0000u; float g; memcpy(&g, &u, 4); return g; }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 64;          // layers per tile (activations stay O(1): weights are scaled for it)
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;
  const int tiles = wgs * kWaves;
  int stagger = getenv("STAGGER") ? 1 : 0;

  if (argc > 3) {
    float *po, *ps; hipMalloc(&po, 16 * 64 * 4); hipMalloc(&ps, 64 * 12 * 4);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, po, ps);
    std::vector<float> o(16 * 64), sp(64 * 12);
    hipMemcpy(o.data(), po, o.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(sp.data(), ps, sp.size() * 4, hipMemcpyDeviceToHost);
    printf("probe D (expect D[m][n] = 100 m + n for m < 16, 0 above): lane 0 regs:");
    for (int r = 0; r < 16; ++r) printf(" %g", o[r * 64]);
    printf("\nlane 33 regs:");
    for (int r = 0; r < 16; ++r) printf(" %g", o[r * 64 + 33]);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = lane & 31;
        const float want = m < 16 ? 100.0f * m + n : 0.0f;
        // bf16 truncation of 100 k + n: compare against the truncated value
        float tv = top_val(want);
        if (o[r * 64 + lane] != tv) ++bad;
      }
    printf("\nlayout mismatches: %d of 1024\n", bad);
    int sbad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const float x = 1.2345678f * (lane + 1) / (j + 3) * ((j & 1) ? -1.0f : 1.0f);
        float pc[3];
        for (int k = 0; k < 3; ++k) { uint32_t d; memcpy(&d, &sp[(lane * 3 + k) * 4 + j / 2], 4); uint32_t u = (j & 1) ? (d & 0xffff0000u) : (d << 16); memcpy(&pc[k], &u, 4); }
        if ((pc[0] + pc[1]) + pc[2] != x) { if (sbad < 4) printf("split lane %d j %d: %.9g vs %.9g %.9g %.9g\n", lane, j, x, pc[0], pc[1], pc[2]); ++sbad; }
      }
    printf("split mismatches: %d of 512\n", sbad);
    return 0;
  }
  srand(7);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  // weights W[layer][k][m], step order of the activations: step s, half h <-> feature acc_feat(s / 16, s % 16, h)
  std::vector<float> W((size_t)kLayers * kK * kM);
  for (auto& v : W) v = rnd() * 0.3f;
  if (getenv("IDENT")) for (int l = 0; l < kLayers; ++l) for (int k = 0; k < kK; ++k) for (int m = 0; m < kM; ++m) W[((size_t)l * kK + k) * kM + m] = (atoi(getenv("IDENT")) == 2 ? (m == (k + 1) % kM) : (k == m)) ? 1.0f : 0.0f;
  std::vector<float> X((size_t)tiles * kSteps * 64);
  for (auto& v : X) v = fabsf(rnd());
  std::vector<float> wa((size_t)kLayers * kSteps * kNT * 64);
  std::vector<uint32_t> wb((size_t)kLayers * kBlocks * kNT * 3 * 64 * 4);
  for (int l = 0; l < kLayers; ++l)
    for (int s = 0; s < kSteps; ++s)
      for (int t = 0; t < kNT; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int h = lane >> 5, i = lane & 31;
          const int k = acc_feat(s / 16, s % 16, h);
          const float w = W[((size_t)l * kK + k) * kM + 32 * t + i];
          wa[(((size_t)l * kSteps + s) * kNT + t) * 64 + lane] = w;
          const int q = s / 8, j = s % 8;
          const float p1 = top_val(w), r1 = w - p1, p2 = top_val(r1), r2 = r1 - p2;
          const float pc[3] = {p1, p2, r2};
          if (top_val(r2) != r2) { printf("split not exact\n"); return 1; }
          for (int pcs = 0; pcs < 3; ++pcs) {
            uint32_t& d = wb[(((((size_t)l * kBlocks + q) * kNT + t) * 3 + pcs) * 64 + lane) * 4 + j / 2];
            const uint32_t b = top_bits(pc[pcs]);
            d = (j & 1) ? ((d & 0x0000ffffu) | (b << 16)) : ((d & 0xffff0000u) | b);
          }
        }
  float *dwa, *dx, *dy; uint32_t* dwb;
  hipMalloc(&dwa, wa.size() * 4); hipMalloc(&dwb, wb.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&dy, X.size() * 4);
  hipMemcpy(dwa, wa.data(), wa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  const int lds_a = (kLayers * kSteps * kNT * 64 + kWaves * kSteps * 64) * 4, lds_b = (kLayers * kBlocks * kNT * 3 * 64 * 4 + kWaves * kSteps * 64) * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_f32), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split<6>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  printf("LDS per workgroup: f32 %d B, split %d B; %d workgroups of 8 waves, %d layers per tile\n", lds_a, lds_b, wgs, reps);
  // fp64 reference of the first 8 tiles
  const int chk = getenv("CHK_ALL") ? tiles : 8;
  std::vector<double> ref((size_t)chk * kSteps * 64);
  for (int tl = 0; tl < chk; ++tl)
    for (int n = 0; n < 32; ++n) {
      double a[kK], b[kM];
      for (int s = 0; s < kSteps; ++s) for (int h = 0; h < 2; ++h) a[acc_feat(s / 16, s % 16, h)] = X[((size_t)tl * kSteps + s) * 64 + n + 32 * h];
      for (int it = 0; it < reps; ++it) {
        const float* w = &W[(size_t)(it % kLayers) * kK * kM];
        for (int m = 0; m < kM; ++m) { double sum = 0; for (int k = 0; k < kK; ++k) sum += a[k] * (double)w[(size_t)k * kM + m]; b[m] = sum > 0 ? sum : 0; }
        memcpy(a, b, sizeof(a));
      }
      for (int s = 0; s < kSteps; ++s) for (int h = 0; h < 2; ++h) ref[((size_t)tl * kSteps + s) * 64 + n + 32 * h] = a[acc_feat(s / 16, s % 16, h)];
    }
  std::vector<float> Y(X.size());
  auto report = [&](const char* name, float ms) {
    hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
    double e = 0, mag = 0;
    size_t glitch = 0;
    for (size_t i = 0; i < ref.size(); ++i) { const double d = fabs((double)Y[i] - ref[i]); if (d > 1e-3) ++glitch; e = fmax(e, d); mag = fmax(mag, fabs(ref[i])); }
    if (getenv("CHK_ALL")) printf("  values off by more than 1e-3: %zu of %zu\n", glitch, ref.size());
    if (getenv("IDENT")) { int shown = 0; for (size_t i = 0; i < ref.size() && shown < 12; ++i) if (fabs((double)Y[i] - ref[i]) > 1e-3) { printf("  tile %zu step %zu lane %zu: got %g want %g\n", i / (kSteps * 64), (i / 64) % kSteps, i % 64, Y[i], ref[i]); ++shown; } }
    printf("%-28s %8.3f us per launch, %7.2f ns per layer-tile-wave-slot, max |err| vs fp64 %.3e (|y| <= %.3g)\n", name, ms * 1e3, ms * 1e6 / reps, e, mag);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
  };
  float ms;
  ms = time([&]() { hipLaunchKernelGGL(k_f32, dim3(wgs), dim3(kWaves * 64), lds_a, 0, dwa, dx, dy, reps); });
  report("fp32 32x32x2", ms);
  ms = time([&]() { hipLaunchKernelGGL(k_split<6>, dim3(wgs), dim3(kWaves * 64), lds_b, 0, dwb, dx, dy, reps, stagger ? stagger++ : 0); });
  report("bf16x3, 6 products", ms);
  ms = time([&]() { hipLaunchKernelGGL(k_split<3>, dim3(wgs), dim3(kWaves * 64), lds_b, 0, dwb, dx, dy, reps, 0); });
  report("bf16x3, 3 products", ms);
  ms = time([&]() { hipLaunchKernelGGL(k_split<1>, dim3(wgs), dim3(kWaves * 64), lds_b, 0, dwb, dx, dy, reps, 0); });
  report("plain bf16 (1 product)", ms);
  if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
