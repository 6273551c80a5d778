// When does a v_mfma_f32_32x32x16_bf16 fetch its operands and deliver its result, with a second wave on the SIMD?
// (The question behind the split form's operand hazard, rc_dev_mlp.h HAZARD.)
//
// One workgroup of 8 waves per CU = two waves per SIMD.  Waves 0-3 ("probe") run, in hand-written asm with fixed registers,
//     [CHAIN dependent MFMAs acc += A x B]   (A x B adds exactly 1.0 to every accumulator element)
//     s_nop <gap>
//     either  v_mov A[0], <bf16 2.0>   (WAR: an MFMA that fetches A after this adds 2.0)  + repair of A afterwards
//     or      v_mov out, acc[0]        (RAW: a result that is not there yet is one short)
// ITER times.  Waves 4-7 ("hog") are the partners on the same SIMDs: idle, or back-to-back chains of the same MFMA, or
// the same chains with a few vector instructions between them.  The host counts the probe lanes whose sum is off, per gap.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_pending.hip -o mfma_pending ; run: ./mfma_pending
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// fixed registers of the probe: A = v[100:103], B = v[104:107], acc = v[108:123], scratch v124 (out), v125 (junk), v126 (A0 good)
#define ACC_CLOBBERS "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123"
#define MFMA_CHAIN ".rept %c1\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\t.endr\n\t"
#define DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
template <int MODE, int GAP, int CHAIN>
__device__ __forceinline__ void probe_iter(float& out_sum) {
  float out;
  constexpr int NOP = GAP > 0 ? GAP - 1 : 0;
  if constexpr (MODE == 0 && GAP == 0) {   // WAR: overwrite A[0] right behind the last MFMA of the chain, repair it later
    asm volatile(MFMA_CHAIN "v_mov_b32 v100, v125\n\t" DRAIN "v_mov_b32 v100, v126\n\ts_nop 1\n\tv_mov_b32 %0, v108\n\t"
                 : "=v"(out) : "n"(CHAIN) : "v100", ACC_CLOBBERS, "memory");
  } else if constexpr (MODE == 0) {
    asm volatile(MFMA_CHAIN "s_nop %c2\n\tv_mov_b32 v100, v125\n\t" DRAIN "v_mov_b32 v100, v126\n\ts_nop 1\n\tv_mov_b32 %0, v108\n\t"
                 : "=v"(out) : "n"(CHAIN), "n"(NOP) : "v100", ACC_CLOBBERS, "memory");
  } else if constexpr (MODE == 2) {        // chain of dependent MFMAs with GAP vector instructions between two of them; read after a full drain
    asm volatile(".rept %c1\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\t"
                 ".rept %c2\n\tv_add_f32 v124, v124, v124\n\t.endr\n\t.endr\n\t" DRAIN "v_mov_b32 %0, v108\n\t"
                 : "=v"(out) : "n"(CHAIN), "n"(GAP) : "v124", ACC_CLOBBERS, "memory");
  } else if constexpr (GAP == 0) {         // RAW: read acc[0] right behind the last MFMA of the chain
    asm volatile(MFMA_CHAIN "v_mov_b32 %0, v108\n\t" DRAIN : "=v"(out) : "n"(CHAIN) : ACC_CLOBBERS, "memory");
  } else {
    asm volatile(MFMA_CHAIN "s_nop %c2\n\tv_mov_b32 %0, v108\n\t" DRAIN : "=v"(out) : "n"(CHAIN), "n"(NOP) : ACC_CLOBBERS, "memory");
  }
  out_sum = out;
}

template <int MODE, int GAP, int CHAIN>
__global__ __launch_bounds__(512) void k_probe(int iters, int hog, float* __restrict__ result, int slot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4) {
    // partner waves: hog = 0 idle, 1 back-to-back dependent MFMAs, 2 the same with vector work in between
    if (hog == 0) return;
    asm volatile("v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
                 "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\ts_nop 1\n\t" ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    for (int it = 0; it < iters * 3; ++it) {
      if (hog == 1)
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\t.endr\n\t" ::: "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
      else
        asm volatile(".rept 4\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\tv_add_f32 v124, v124, v124\n\tv_add_f32 v124, v124, v124\n\t.endr\n\t" ::: "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124");
    }
    return;
  }
  // probe: A[k-slot 0 of the first half-wave] = 1.0 on every row, B likewise on every column -> every MFMA adds 1.0 everywhere
  const uint32_t one = lane < 32 ? 0x00003f80u : 0u, two = lane < 32 ? 0x00004000u : 0u;
  asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
               "v_mov_b32 v104, %0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
               "v_mov_b32 v125, %1\n\tv_mov_b32 v126, %0\n\t"
               "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
               "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\ts_nop 1\n\t"
               :: "v"(one), "v"(two)
               : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v125", "v126");
  int bad = 0;
  float last = 0.0f;
  for (int it = 0; it < iters; ++it) {
    float v;
    probe_iter<MODE, GAP, CHAIN>(v);
    const float want = (float)((it + 1) * CHAIN);
    if (MODE == 0) { if (it == iters - 1) last = v; }         // WAR: only the final sum matters (read after everything drained)
    else if (MODE == 2) { if (v != want) ++bad; }             // interleaved chain: every drained read must be complete
    else if (v != want) ++bad;                                 // RAW: every read must see all MFMAs issued so far
  }
  if (MODE == 0) bad = last != (float)(iters * CHAIN) ? 1 : 0;
  // one number per launch: lanes (of all probe waves of all workgroups) that saw a wrong value
  if (bad) atomicAdd(&result[slot], 1.0f);
  if (MODE == 0 && lane == 0 && wave == 0 && blockIdx.x == 0) result[64 + slot] = last;
}

// The same with the partner in ANOTHER workgroup: 4 waves per workgroup, two workgroups per CU (512 workgroups), the
// role of a workgroup by its arrival order on its CU (hardware CU id + an atomic counter: first = probe, second = partner).
template <int MODE, int GAP, int CHAIN>
__global__ __launch_bounds__(256) void k_probe_wg(int iters, int hog, float* __restrict__ result, int slot, int* __restrict__ cu_slots) {
  __shared__ int s_role;
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: CU_ID 11:8, SH_ID 12, SE_ID 15:13
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // HW_REG_XCC_ID 3:0
    s_role = atomicAdd(&cu_slots[slot * 4096 + (((xcc & 15u) << 8) | ((hw >> 8) & 255u))], 1) & 1;
  }
  __syncthreads();
  const int role = s_role;
  const int lane = threadIdx.x & 63;
  if (role == 1) {
    if (hog == 0) return;
    asm volatile("v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
                 "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\ts_nop 1\n\t" ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    for (int it = 0; it < iters * 3; ++it) {
      if (hog == 1)
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\t.endr\n\t" ::: ACC_CLOBBERS);
      else
        asm volatile(".rept 4\n\tv_mfma_f32_32x32x16_bf16 v[108:123], v[100:103], v[104:107], v[108:123]\n\tv_add_f32 v124, v124, v124\n\tv_add_f32 v124, v124, v124\n\t.endr\n\t" ::: ACC_CLOBBERS, "v124");
    }
    return;
  }
  const uint32_t one = lane < 32 ? 0x00003f80u : 0u, two = lane < 32 ? 0x00004000u : 0u;
  asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
               "v_mov_b32 v104, %0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
               "v_mov_b32 v125, %1\n\tv_mov_b32 v126, %0\n\t"
               "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
               "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\ts_nop 1\n\t"
               :: "v"(one), "v"(two)
               : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", ACC_CLOBBERS, "v125", "v126");
  int bad = 0;
  float last = 0.0f;
  for (int it = 0; it < iters; ++it) {
    float v;
    probe_iter<MODE, GAP, CHAIN>(v);
    const float want = (float)((it + 1) * CHAIN);
    if (MODE == 0) { if (it == iters - 1) last = v; }
    else if (v != want) ++bad;
  }
  if (MODE == 0) bad = last != (float)(iters * CHAIN) ? 1 : 0;
  if (bad) atomicAdd(&result[slot], 1.0f);
}

template <int MODE, int CHAIN, int... GAPS>
void sweep_wg(const char* what, int iters, float* d_res, int* d_slots) {
  constexpr int gaps[] = {GAPS...};
  for (int hog = 0; hog < 3; ++hog) {
    hipMemset(d_res, 0, 128 * sizeof(float));
    hipMemset(d_slots, 0, 16 * 4096 * sizeof(int));
    int slot = 0;
    auto launch = [&](auto G) { k_probe_wg<MODE, decltype(G)::value, CHAIN><<<dim3(512), dim3(256), 0, 0>>>(iters, hog, d_res, slot, d_slots); ++slot; };
    (launch(std::integral_constant<int, GAPS>{}), ...);
    hipDeviceSynchronize();
    std::vector<float> r(128);
    hipMemcpy(r.data(), d_res, 128 * sizeof(float), hipMemcpyDeviceToHost);
    printf("%s, chain of %d, partner WORKGROUP %s:", what, CHAIN, hog == 0 ? "idle          " : (hog == 1 ? "MFMA chains   " : "MFMA + vector "));
    for (int i = 0; i < (int)sizeof...(GAPS); ++i) printf("  gap %2d: %6.0f", gaps[i], r[i]);
    printf("   (probe lanes with a wrong value)\n");
  }
}

template <int MODE, int CHAIN, int... GAPS>
void sweep(const char* what, int iters, float* d_res) {
  constexpr int gaps[] = {GAPS...};
  for (int hog = 0; hog < 3; ++hog) {
    hipMemset(d_res, 0, 128 * sizeof(float));
    int slot = 0;
    auto launch = [&](auto G) { k_probe<MODE, decltype(G)::value, CHAIN><<<dim3(256), dim3(512), 0, 0>>>(iters, hog, d_res, slot++); };
    (launch(std::integral_constant<int, GAPS>{}), ...);
    hipDeviceSynchronize();
    std::vector<float> r(128);
    hipMemcpy(r.data(), d_res, 128 * sizeof(float), hipMemcpyDeviceToHost);
    printf("%s, chain of %d, partner %s:", what, CHAIN, hog == 0 ? "idle          " : (hog == 1 ? "MFMA chains   " : "MFMA + vector "));
    for (int i = 0; i < (int)sizeof...(GAPS); ++i) printf("  gap %2d: %6.0f", gaps[i], r[i]);
    printf("   (lanes with a wrong value, of %d)\n", 256 * 4 * 64);
  }
}

int main() {
  float* d_res;
  hipMalloc(&d_res, 128 * sizeof(float));
  const int iters = 2000;
  printf("v_mfma_f32_32x32x16_bf16, two waves per SIMD, %d iterations per probe wave; gap = wait states (s_nop) behind the last MFMA\n", iters);
  sweep<0, 1, 0, 1, 2, 4, 8, 12, 16, 24, 32, 48, 64>("operand overwritten (WAR)", iters, d_res);
  sweep<0, 6, 0, 1, 2, 4, 8, 12, 16, 24, 32, 48, 64>("operand overwritten (WAR)", iters, d_res);
  sweep<1, 1, 0, 4, 8, 10, 12, 14, 16, 24, 32, 48, 64>("accumulator read (RAW)   ", iters, d_res);
  sweep<1, 6, 0, 4, 8, 10, 12, 14, 16, 24, 32, 48, 64>("accumulator read (RAW)   ", iters, d_res);
  printf("dependent chain with `gap` vector instructions between two MFMAs, read after a drain:\n");
  sweep<2, 6, 1, 2, 3, 4, 5, 6, 8, 12>("interleaved chain        ", iters, d_res);
  sweep<2, 42, 1, 2, 3, 4, 5, 6, 8, 12>("interleaved chain        ", iters / 4, d_res);
  int* d_slots;
  hipMalloc(&d_slots, 16 * 4096 * sizeof(int));
  printf("partner in ANOTHER workgroup (two 4-wave workgroups per CU, roles by arrival on the CU):\n");
  sweep_wg<0, 6, 0, 1, 2, 4, 8, 16, 32>("operand overwritten (WAR)", iters, d_res, d_slots);
  sweep_wg<1, 6, 0, 4, 8, 10, 12, 16, 32>("accumulator read (RAW)   ", iters, d_res, d_slots);
  sweep_wg<2, 6, 1, 2, 3, 4, 6, 8, 12>("interleaved chain        ", iters, d_res, d_slots);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
