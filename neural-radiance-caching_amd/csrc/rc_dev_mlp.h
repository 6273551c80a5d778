// Device-side building blocks of the MFMA MLP kernels (shared by rc_mlp.hip and rc_fused.hip).
// See rc_mlp.hip for the formulation (transposed fp32 MFMA, weight stream through an LDS ring).
#pragma once
#include "rc_internal.h"

namespace rcdev {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;   // waves per workgroup; each wave owns 32 points

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// ---------------------------------------------------------------------------------------------
// Split arithmetic (rc_pack_host.h, RC_SPLIT_MFMA): fp32 operands as three bf16 pieces each, six products per 16 k on
// v_mfma_f32_32x32x16_bf16.  The weights are split by the host; the activations here, once per block of 8 k-steps (the 8
// values a lane holds: its own column, its own half-wave -- the k-slots of the MFMA's B operand, which pair with the
// A operand's slots positionally, so the step order of the fp32 form stays as it is).
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 (truncated) of a0 in the low half, of a1 in the high half
__device__ __forceinline__ uint32_t pack_top16(float a1, float a0) {
  return __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
}
__device__ __forceinline__ float top16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
// v[j] == hi[j] + mid[j] + lo[j] exactly (each residual is exact: the top shares its leading bits)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&b)[3]) {
  u32x4 &hi = b[0], &mid = b[1], &lo = b[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    hi[p] = pack_top16(a1, a0);
    const float r0 = a0 - top16(a0), r1 = a1 - top16(a1);
    mid[p] = pack_top16(r1, r0);
    const float l0 = r0 - top16(r0), l1 = r1 - top16(r1);
    lo[p] = pack_top16(l1, l0);
  }
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// INSTABILITY and the rules below (tools/stress_repeat.py, profiles/r04_split_mfma_hazard.txt).  What is measured:
//  * With the layers written plainly (pieces prefetched one cell ahead, the compiler free to schedule), results differ
//    from launch to launch in a few rays of a few hundred by ~1e-3 wherever waves of more than one workgroup or kernel
//    share a SIMD with a split layer: the two-wave kernel of rc_fused2.hip (two workgroups per CU), and even the kernels
//    with one 4-wave workgroup per CU when another stream's or the material stage's concurrent kernel lands on their CU
//    (5 of the 129 GPU tests: the multi-stream and the material launch-plan tests).  Never with a kernel alone on its CUs,
//    never with the fp32 MFMA.
//  * The obvious cause -- the bf16 MFMA fetching operands or delivering results late beside another wave's MFMAs, with
//    the compiler reusing an operand register one instruction behind it -- does NOT show in isolation:
//    tools/micro/mfma_pending.hip overwrites an operand 0 wait states behind the MFMA and reads accumulators 8 wait states
//    behind it, beside idle and MFMA-hogging partner waves, and every sum is right (profiles/r04_mfma_pending_probe.txt).
//  * With the three rules below the full GPU suite and the repeat screen pass (129 tests, every plan and size bitwise
//    stable), except for the two-wave kernel with EVERY layer split at 4097+ rays -- which is why the density MLPs stay
//    fp32 (mlp_layer_d) and a split build does not run that kernel (rc_api.hip).
//  * The common factor of every unstable launch is a wave of ANOTHER workgroup or kernel on the SIMD.  With the SIMD kept
//    exclusive (split_exclusive_simd below: the wave allocates the whole register file) even the plain form passes the
//    robustness tests (20 of 20, three runs; 5 of 20 fail without it).  That is the containment the product relies on;
//    the rules stay as a second line.
// So the rules are empirical: they are what separates the stable from the unstable form here, the mechanism is not
// established.  They cost nothing measurable.
//  1. the MFMAs of a cell issue back to back (scheduling barriers), nothing of the wave in between;
//  2. the operand registers of a cell stay live (split_keep: an empty asm that reads them) until the MFMAs of the NEXT
//     cell have issued; weight pieces rotate through three register sets, activation pieces through two, loads and
//     splits into a set follow the marker that retires it;
//  3. behind a layer's last cell two flush MFMAs (16x16x32 into a 4-register sink that lives as long as the stream
//     object and is never read) issue before the accumulators are read or the last operands are released.
// (diagnostic switches: -DRC_RULE1=0 / -DRC_RULE2=0 / -DRC_RULE3=0 take a rule out; profiles/r04_split_mfma_hazard.txt, 4.)
#ifndef RC_RULE1
#define RC_RULE1 1
#endif
#ifndef RC_RULE2
#define RC_RULE2 1
#endif
#ifndef RC_RULE3
#define RC_RULE3 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_fence() { if constexpr (RC_RULE1 != 0) __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void split_keep(const u32x4& x) { if constexpr (RC_RULE2 != 0) asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void split_keep3(const u32x4 (&x)[3]) { if constexpr (RC_RULE2 != 0) asm volatile("" ::"v"(x[0]), "v"(x[1]), "v"(x[2])); }
template <class WS>
__device__ __forceinline__ void split_flush(const WS& w, const u32x4& any) {
  if constexpr (RC_RULE3 == 0) return;
  __builtin_amdgcn_sched_barrier(0);
  w.sink = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, any), __builtin_bit_cast(bf16x8, any), w.sink, 0, 0, 0);
  w.sink = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, any), __builtin_bit_cast(bf16x8, any), w.sink, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1" : "+v"(w.sink));
  __builtin_amdgcn_sched_barrier(0);
}
// A kernel that runs split-form layers keeps its SIMDs to itself: naming the last architectural and the last accumulation
// register makes the wave allocate the whole 512-entry register file of its SIMD, so no wave of ANY other kernel or stream
// can be placed beside it -- "one wave per SIMD" then holds under concurrent streams too, and a foreign wave on the SIMD
// is the common factor of every unstable launch seen (INSTABILITY above).  Costs nothing when the kernel is alone.
__device__ __forceinline__ void split_exclusive_simd() {
  if constexpr (kRcSplit) asm volatile("" ::: "v255", "a255");
}
// the six products of one (block, tile) cell, smallest first
__device__ __forceinline__ void mfma_split6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16& acc) {
  split_fence();
  acc = mfma_bf16(a[2], b[0], acc);
  acc = mfma_bf16(a[0], b[2], acc);
  acc = mfma_bf16(a[1], b[1], acc);
  acc = mfma_bf16(a[1], b[0], acc);
  acc = mfma_bf16(a[0], b[1], acc);
  acc = mfma_bf16(a[0], b[0], acc);
  split_fence();
}

// ---------------------------------------------------------------------------------------------
// Weight stream.  All MLP layers of a kernel are packed by the host into ONE linear stream of
// 256-byte MFMA A-fragments in exactly the order the kernel consumes them.  The workgroup pulls
// the stream through a two-chunk LDS ring with LDS-DMA (global_load_lds_dwordx4, no VGPRs): while
// the waves run the MFMAs of chunk c out of LDS, chunk c+1 is in flight.  One barrier per chunk
// (kChunk fragments = kChunk MFMAs per wave) is the only synchronisation; every wave of the
// workgroup executes the identical, fully unrolled fragment sequence.
// ---------------------------------------------------------------------------------------------
constexpr int kChunk = 64;                       // fragments per chunk (16 KiB)
constexpr int kRingFloats = 2 * kChunk * 64;     // two chunks

typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct WStream {
  const float* g;    // packed fragment stream (padded to a whole number of chunks)
  float* ring;       // LDS ring [2 * kChunk][64]
  int lane, wave;
  mutable f32x4 sink = {0.0f, 0.0f, 0.0f, 0.0f};     // split form: destination of the flush MFMAs (see INSTABILITY above), never read
  __device__ WStream() = default;
  __device__ WStream(const float* g_, float* ring_, int lane_, int wave_) : g(g_), ring(ring_), lane(lane_), wave(wave_) {}
};

// Issue the LDS-DMA of chunk c (this wave's quarter: 4 x 1 KiB).
// CH: fragments per chunk of this kernel's ring (kChunk by default; the EnvMap kernel runs a 2 x 8 KiB ring).
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ void ws_issue(const WStream& w, int c) {
  static_assert(CH % (4 * W) == 0, "whole 1-KiB pieces per wave");
#pragma unroll
  for (int k = 0; k < CH / 4 / W; ++k) {
    const int i = w.wave + W * k;                  // 1-KiB piece inside the chunk
    const int frag0 = c * CH + 4 * i;
    if (frag0 < NF) {
      const float* src = w.g + (size_t)frag0 * 64 + w.lane * 4;
      float* dst = w.ring + ((c & 1) * CH + 4 * i) * 64;
      __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_ptr)dst, 16, 0, 0);
    }
  }
}

template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ void ws_begin(const WStream& w) {
  ws_issue<NF, W, CH>(w, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (CH < NF) ws_issue<NF, W, CH>(w, 1);
}

// Entering chunk c: it has landed (issued one chunk ago), everybody is done with chunk c-1.
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ void ws_advance(const WStream& w, int c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((c + 1) * CH < NF) ws_issue<NF, W, CH>(w, c + 1);
}

// Read fragment f of the stream as a per-lane value (used for lane-layout constant vectors).
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ float ws_read(const WStream& w, int f) {
  if (f > 0 && f % CH == 0) ws_advance<NF, W, CH>(w, f / CH);
  return w.ring[(f % (2 * CH)) * 64 + w.lane];
}

// One 1-KiB piece of the split form (fragments [f, f + 4), f a multiple of 4): this lane's 8 bf16.
__device__ __forceinline__ u32x4 lds_piece(const float* p) { return *reinterpret_cast<const u32x4*>(p); }
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ u32x4 ws_read4(const WStream& w, int f) {
  static_assert(CH % 4 == 0, "pieces never straddle a chunk");
  if (f > 0 && f % CH == 0) ws_advance<NF, W, CH>(w, f / CH);
  return lds_piece(w.ring + (f % (2 * CH)) * 64 + w.lane * 4);
}

// The same fragment stream read straight from global memory (L2-resident: one kernel's stream is <= 0.7 MB), one
// coalesced 256-byte load per fragment and wave, no LDS ring and therefore NO workgroup barriers: the waves of a
// workgroup are free to drift apart.  Which stream type a kernel uses is a property of the kernel; the arithmetic
// (fragment order, MFMA order) is identical.
struct WDirect {
  const float* g;
  int lane;
  mutable f32x4 sink = {0.0f, 0.0f, 0.0f, 0.0f};
};
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ float ws_read(const WDirect& w, int f) { return w.g[(size_t)f * 64 + w.lane]; }
template <int NF, int W = kWaves, int CH = kChunk>
__device__ __forceinline__ u32x4 ws_read4(const WDirect& w, int f) { return *reinterpret_cast<const u32x4*>(w.g + (size_t)f * 64 + w.lane * 4); }

// One pass over KS k-steps for NT output tiles; the layer's fragments are [FBASE, FBASE + KS*NT)
// of the stream.  act: this lane's activation column (act[s * 64] is step s).
// Software pipelined in groups of SG k-steps: the LDS reads (A fragments + B activations) of group
// g+1 are issued before the MFMAs of group g, with scheduling fences so they stay there; the MFMA
// pipe then runs back to back while the next operands are in flight.
template <int NT, int KS, int FBASE, int NF, int SG, int W, int CH, class WS>
__device__ __forceinline__ void mlp_layer_f32(const WS& w, const float* act, f32x16 (&acc)[NT]) {
  constexpr int NG = (KS + SG - 1) / SG;
  float a[3][SG][NT], b[3][SG];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int d = 0; d < SG; ++d) {
      const int s = g * SG + d;
      if (s < KS) {
        b[buf][d] = act[s * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int f = FBASE + s * NT + t;            // compile-time after unrolling
          a[buf][d][t] = ws_read<NF, W, CH>(w, f);
        }
      }
    }
  };
  auto comp = [&](int g, int buf) {
#pragma unroll
    for (int d = 0; d < SG; ++d) {
      if (g * SG + d < KS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][d][t], b[buf][d], acc[t], 0, 0, 0);
      }
    }
  };
  // operands are fetched TWO groups ahead (three register sets): a chunk seam (barrier + LDS-DMA issue) inside
  // load(g + 2) then sits between groups whose operands are already in registers, and the LDS latency of the
  // new reads has a whole group of MFMAs to hide behind
  load(0, 0);
  if (NG > 1) load(1, 1);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 2 < NG) load(g + 2, (g + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    comp(g, g % 3);
    __builtin_amdgcn_sched_barrier(0);
  }
}


// The split form of the same pass: blocks of 8 k-steps, cells (block, tile) of three 1-KiB pieces in stream order.  The
// activations of the next block and the pieces of the next cell are read before the six MFMAs of the current one.
template <int NT, int KS, int FBASE, int NF, int W, int CH, class WS>
__device__ __forceinline__ void mlp_layer_split(const WS& w, const float* act, f32x16 (&acc)[NT]) {
  static_assert(FBASE % 4 == 0, "split layers start on a 1-KiB piece");
  constexpr int NB = (KS + 7) / 8, NC = NB * NT;
  float bv[8];
  u32x4 a[3][3], b[2][3];
  auto load_b = [&](int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (8 * q + j < KS) ? act[(8 * q + j) * 64] : 0.0f;
  };
  auto load_a = [&](int cell) {
#pragma unroll
    for (int p = 0; p < 3; ++p) a[cell % 3][p] = ws_read4<NF, W, CH>(w, FBASE + (cell * 3 + p) * 4);
  };
  load_b(0);
  load_a(0);
  if (NC > 1) load_a(1);
  split8(bv, b[0]);
  if (NB > 1) load_b(1);
#pragma unroll
  for (int q = 0; q < NB; ++q) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int cell = q * NT + t;
      mfma_split6(a[cell % 3], b[q & 1], acc[t]);
      // the MFMAs of this cell have issued: those of the cell before have started, its registers may go
      if (cell >= 1) split_keep3(a[(cell - 1) % 3]);
      if (cell + 2 < NC) load_a(cell + 2);
      if (t == 0) {
        if (q >= 1) split_keep3(b[(q - 1) & 1]);
        if (q + 1 < NB) {
          split8(bv, b[(q + 1) & 1]);
          if (q + 2 < NB) load_b(q + 2);
        }
      }
    }
  }
  split_flush(w, b[(NB - 1) & 1][0]);
  split_keep3(a[(NC - 1) % 3]);
  split_keep3(b[(NB - 1) & 1]);
}

template <int NT, int KS, int FBASE, int NF, int SG = (NT >= 8 ? 1 : (NT >= 4 ? 2 : (NT >= 2 ? 4 : 8))), int W = kWaves, int CH = kChunk, class WS = WStream>
__device__ __forceinline__ void mlp_layer(const WS& w, const float* act, f32x16 (&acc)[NT]) {
  if constexpr (kRcSplit) mlp_layer_split<NT, KS, FBASE, NF, W, CH, WS>(w, act, acc);
  else mlp_layer_f32<NT, KS, FBASE, NF, SG, W, CH, WS>(w, act, acc);
}

// A layer of a proposal level's density MLP: the exact fp32 MFMA chain in every build (rc_pack_host.h rc_lfr32).
template <int NT, int KS, int FBASE, int NF, int SG = (NT >= 8 ? 1 : (NT >= 4 ? 2 : (NT >= 2 ? 4 : 8))), int W = kWaves, int CH = kChunk, class WS = WStream>
__device__ __forceinline__ void mlp_layer_d(const WS& w, const float* act, f32x16 (&acc)[NT]) {
  mlp_layer_f32<NT, KS, FBASE, NF, SG, W, CH, WS>(w, act, acc);
}

// The bias k-step of a layer (fragments [FBASE, FBASE + NT): the bias row of each tile) with its B operand -- 1 on the
// first half-wave, 0 on the second -- taken from a register instead of an activation slot: a 256-wide layer then needs
// 128 activation steps in LDS, not 129.  Same MFMAs in the same order as a 129th step of mlp_layer.
template <int NT, int FBASE, int NF, int W = kWaves, int CH = kChunk, class WS = WStream>
__device__ __forceinline__ void mlp_bias_step(const WS& w, f32x16 (&acc)[NT]) {
  if constexpr (kRcSplit) {
    // the layer's last block, whose only live step is the bias step (slot 0 of the first half-wave): B = bf16(1) there,
    // so the three pieces of the bias row add up exactly (hi + mid + lo) in three MFMAs per tile
    u32x4 b1;
    b1[0] = w.lane < 32 ? 0x00003f80u : 0u; b1[1] = 0u; b1[2] = 0u; b1[3] = 0u;
    u32x4 a[NT][3];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[t][p] = ws_read4<NF, W, CH>(w, FBASE + (t * 3 + p) * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      split_fence();
      acc[t] = mfma_bf16(a[t][2], b1, acc[t]);
      acc[t] = mfma_bf16(a[t][1], b1, acc[t]);
      acc[t] = mfma_bf16(a[t][0], b1, acc[t]);
      split_fence();
    }
    split_flush(w, b1);
#pragma unroll
    for (int t = 0; t < NT; ++t) split_keep3(a[t]);
    split_keep(b1);
    return;
  }
  const float one = w.lane < 32 ? 1.0f : 0.0f;
  float a[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) a[t] = ws_read<NF, W, CH>(w, FBASE + t);
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], one, acc[t], 0, 0, 0);
}

// Same as mlp_layer_d (density MLPs: fp32 MFMA in every build) for PT point-tiles per wave (64 points): every A fragment
// read from the ring feeds PT MFMAs.  Tile p's activation column starts at act + p * tile_stride.
template <int PT, int NT, int KS, int FBASE, int NF, int SG = (NT * PT >= 8 ? 1 : (NT * PT >= 4 ? 2 : 4)), int W = kWaves, class WS = WStream>
__device__ __forceinline__ void mlp_layer_pt(const WS& w, const float* act, int tile_stride, f32x16 (&acc)[PT][NT]) {
  constexpr int NG = (KS + SG - 1) / SG;
  float a[3][SG][NT], b[3][SG][PT];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int d = 0; d < SG; ++d) {
      const int s = g * SG + d;
      if (s < KS) {
#pragma unroll
        for (int p = 0; p < PT; ++p) b[buf][d][p] = act[p * tile_stride + s * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int f = FBASE + s * NT + t;
          a[buf][d][t] = ws_read<NF, W, kChunk>(w, f);
        }
      }
    }
  };
  auto comp = [&](int g, int buf) {
#pragma unroll
    for (int d = 0; d < SG; ++d) {
      if (g * SG + d < KS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int p = 0; p < PT; ++p)
            acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][d][t], b[buf][d][p], acc[p][t], 0, 0, 0);
      }
    }
  };
  load(0, 0);
  if (NG > 1) load(1, 1);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 2 < NG) load(g + 2, (g + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    comp(g, g % 3);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// max(x, 0) as ONE instruction.  fmaxf() on a value the compiler cannot prove canonical -- an MFMA result -- is preceded
// by a canonicalising `v_max_f32 x, x, x`: two instructions per activation, 64 per 32-point tile of a proposal MLP.
// Split form: the inline-asm read of an accumulator right behind a v_mfma_f32_32x32x16_bf16 was NOT covered by the
// compiler's hazard handling (tools/micro/split_mfma.hip returned the accumulator of one MFMA earlier), so the ReLU is the
// integer maximum of the bit pattern against 0 there -- also one instruction, no canonicalisation, visible to the
// compiler: negative floats (and -0) are negative integers.
__device__ __forceinline__ float relu0(float x) {
  if constexpr (kRcSplit) {
    const int i = __float_as_int(x);
    return __int_as_float(i > 0 ? i : 0);
  }
  float y;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

// Park NT accumulator tiles as the next layer's activation steps [base, base + 16 NT).
template <int NT, bool RELU>
__device__ __forceinline__ void park(const f32x16 (&acc)[NT], float* act, int base) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[t][r];
      act[(base + t * 16 + r) * 64] = RELU ? relu0(v) : v;
    }
}

// Output layers with a handful of rows (density + predicted normals, integrated BRDF, ambient rgb): in-register dot
// products over the hidden activations as they sit in the accumulators (ReLU applied here), instead of a mostly
// empty MFMA tile.  Fragments [FBASE, FBASE + NO * NT * 16 + NO): for output o, tile t, register r the weights
// w_o[feature(t, r, half-wave)], then one fragment per output holding the bias on every lane.  PT point-tiles share
// each weight read.  `keep` (optional, KEEP) returns the weights of output 0 (the backward pass of the analytic
// normals starts from them).  Two partial sums per output (even / odd registers), the two half-waves added last.
// NOB: outputs the layer was packed with (the bias fragments sit behind NOB * NT * 16 weight fragments); a caller
// that needs only the first NO < NOB outputs skips the rest.  CH: fragments per ring chunk (see ws_issue).
template <int PT, int NT, int NO, int FBASE, int NF, bool KEEP = false, int W = kWaves, int NOB = NO, int CH = kChunk, class WS = WStream>
__device__ __forceinline__ void dot_out(const WS& w, const f32x16 (&hid)[PT][NT], float (&out)[PT][NO],
                                        float (&keep)[KEEP ? NT * 16 : 1]) {
  float part[PT][NO][2];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int o = 0; o < NO; ++o) { part[p][o][0] = 0.0f; part[p][o][1] = 0.0f; }
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float wv = ws_read<NF, W, CH>(w, FBASE + (o * NT + t) * 16 + r);
        if constexpr (KEEP) { if (o == 0) keep[t * 16 + r] = wv; }
#pragma unroll
        for (int p = 0; p < PT; ++p) part[p][o][r & 1] = __builtin_fmaf(relu0(hid[p][t][r]), wv, part[p][o][r & 1]);
      }
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float bias = ws_read<NF, W, CH>(w, FBASE + NOB * NT * 16 + o);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      float sum = part[p][o][0] + part[p][o][1];
      sum = sum + __shfl_xor(sum, 32, 64);
      out[p][o] = sum + bias;
    }
  }
}
template <int NT, int NO, int FBASE, int NF, bool KEEP = false, int W = kWaves, int NOB = NO, int CH = kChunk, class WS = WStream>
__device__ __forceinline__ void dot_out1(const WS& w, const f32x16 (&hid)[NT], float (&out)[NO],
                                         float (&keep)[KEEP ? NT * 16 : 1]) {
  const f32x16 (&h1)[1][NT] = reinterpret_cast<const f32x16 (&)[1][NT]>(hid);
  float (&o1)[1][NO] = reinterpret_cast<float (&)[1][NO]>(out);
  dot_out<1, NT, NO, FBASE, NF, KEEP, W, NOB, CH, WS>(w, h1, o1, keep);
}

__device__ __forceinline__ float softplus(float x) {
  // jax.nn.softplus = logaddexp(x, 0) = max(x, 0) + log1p(exp(-|x|))
  return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

#ifndef RC_DEV_CONTRACT3
#define RC_DEV_CONTRACT3
__device__ __forceinline__ void contract3(float& x, float& y, float& z, float radius) {
  x = rc_div(x, radius); y = rc_div(y, radius); z = rc_div(z, radius);
  float mag = x * x + y * y + z * z;
  mag = fmaxf(1.0f, mag);
  const float scale = (2.0f * sqrtf(mag) - 1.0f) / mag;
  x = scale * x; y = scale * y; z = scale * z;
}
#endif

// nan_to_num(-l2_normalize(g)) (ref_utils.py:45-72, geometry.py:460,471)
__device__ __forceinline__ void neg_normalize(float& x, float& y, float& z) {
  const float dsq = x * x + y * y + z * z;
  const float inv = sqrtf(fmaxf(RC_TINY, dsq));
  float nx = -(x / inv), ny = -(y / inv), nz = -(z / inv);
  if (dsq < RC_TINY) { nx = 0.0f; ny = 0.0f; nz = 0.0f; }
  auto fix = [](float v) {
    if (v != v) return 0.0f;
    return fminf(fmaxf(v, -RC_FMAX), RC_FMAX);
  };
  x = fix(nx); y = fix(ny); z = fix(nz);
}


// (l, m) of IDE term i for deg_view = 5: l in {1,2,4,8,16}, m = 0..l (ref_utils.py:105-115)
__host__ __device__ constexpr int ide_l(int i) { return i < 2 ? 1 : (i < 5 ? 2 : (i < 10 ? 4 : (i < 19 ? 8 : 16))); }
__host__ __device__ constexpr int ide_m(int i) { return i < 2 ? i : (i < 5 ? i - 2 : (i < 10 ? i - 5 : (i < 19 ? i - 10 : i - 19))); }


// activation slice of the shader (steps): [0,48) feature (hidden 32 | appearance 16) | [48,84) IDE | 84 bias(1|0) |
// (the (n.v | 1) step of the integrated BRDF reuses step 48 once the IDE is dead).  The 128-wide shader bottleneck
// (Dense(96 -> 128) WITHOUT activation, nerf.py:394-396) only feeds
// linear layers -- SLF layer_0, the input part of SLF layer_bottleneck, integrated_brdf_layers_0 -- so the host
// folds it into those (W' = W_b W[:128], b' = b + b_b W[:128], products in fp64): the layer itself and 16 k-steps of
// each consumer disappear (1767 instead of 2123 MFMAs per tile), results equal up to fp32 rounding of the folded weights.
constexpr int kShActSteps = 102;
constexpr int kStepIde = 48;
constexpr int kStepBias = 84;

// fragment offsets of the shader's layers inside its weight stream (host: rc_api.hip, same order)
struct ShaderFrags {
  static constexpr int F_H = 0, F_S0 = F_H + rc_lfr(49, 1), F_I0 = F_S0 + rc_lfr(85, 8), F_I1 = F_I0 + rc_lfr(49, 2), F_IO = F_I1 + rc_lfr(33, 2),
                       F_S1 = F_IO + rc_dfr(1, 2), F_S2 = F_S1 + rc_lfr(65, 4), F_SB = F_S2 + rc_lfr(65, 4), F_SO = F_SB + rc_lfr(64, 4),
                       COUNT = F_SO + rc_dfr(3, 4);
};

struct ShaderConsts { float roughness_bias, irradiance_bias, ambient_bias, rgb_max, slf_ambient_bias; };
struct ShadeOut { float rgb[3], ad[3], idf[3], is[3], tint[3]; };

// The cache shader on one 32-point tile.  Expects act steps [0,49) = [hidden density feature (32,
// accumulator order) | appearance features (16 natural pairs) | bias] (the feature stays there until the IBRDF
// chain has read it); (nx,ny,nz) the shading normal
// and (vx,vy,vz) the view direction of this lane's point.  F0 = offset of the shader's fragments in
// the kernel's weight stream of NF fragments.
#ifdef RC_STAMPS
#define RC_TSTAMP(i) do { if (st) { __builtin_amdgcn_sched_barrier(0); st[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RC_TSTAMP(i) do { } while (0)
#endif
template <int F0, int NF, class WS = WStream>
__device__ __forceinline__ ShadeOut shader_tile(const WS& ws, float* act, int lane, int h, float nx, float ny, float nz,
                                                float vx, float vy, float vz, const RcIdeTable* tb, const ShaderConsts& k,
                                                unsigned long long* st = nullptr) {
  RC_TSTAMP(0);
  // ---- small heads tile on the feature (the bottleneck is folded into its consumers, see kShActSteps)
  float rough, tint[3], ad[3], idf[3];
  {
    f32x16 acc[1];
    acc[0] = zero16();
    mlp_layer<1, 49, F0 + ShaderFrags::F_H, NF>(ws, act, acc);
    // by accumulator register (same on both half-waves): 0 roughness, 1-3 tint, 4-6 ambient irradiance, 7-9 irradiance
    rough = softplus(acc[0][0] + k.roughness_bias);                       // nerf.py:633-634
    tint[0] = sigmoidf(acc[0][1]); tint[1] = sigmoidf(acc[0][2]); tint[2] = sigmoidf(acc[0][3]);   // :976
    const float ar[3] = {acc[0][4], acc[0][5], acc[0][6]};
    const float ir[3] = {acc[0][7], acc[0][8], acc[0][9]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ad[c] = fminf(fmaxf(softplus(ar[c] + k.ambient_bias), 0.0f), k.rgb_max);      // nerf.py:965-969
      idf[c] = fminf(fmaxf(softplus(ir[c] + k.irradiance_bias), 0.0f), k.rgb_max);  // nerf.py:1008-1012
    }
  }
  RC_TSTAMP(1);
  // ---- normals, n.(-v), reflection direction, IDE
  float dot_nv;
  {
    const float dotp = nx * (-vx) + ny * (-vy) + nz * (-vz);        // nerf.py:474
    // reflect(-v, n) = 2 (n . -v) n - (-v)  (ref_utils.py:25-42)
    const float rx = 2.0f * dotp * nx - (-vx), ry = 2.0f * dotp * ny - (-vy), rz = 2.0f * dotp * nz - (-vz);
    act[kStepBias * 64] = h == 0 ? 1.0f : 0.0f;
    dot_nv = dotp;
    // IDE (ref_utils.py:155-190): low half-wave keeps real parts, high half-wave imaginary parts.
    float zp[RC_IDE_ZPOW];
    zp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k < RC_IDE_ZPOW; ++k) zp[k] = zp[k - 1] * rz;
    float cpw[RC_IDE_ZPOW];   // Re or Im of (x + i y)^m for this half-wave
    {
      float cre = 1.0f, cim = 0.0f;
      cpw[0] = h == 0 ? cre : cim;
#pragma unroll
      for (int m = 1; m < RC_IDE_ZPOW; ++m) {
        const float nre = cre * rx - cim * ry;
        const float nim = cre * ry + cim * rx;
        cre = nre; cim = nim;
        cpw[m] = h == 0 ? cre : cim;
      }
    }
#pragma unroll
    for (int i = 0; i < RC_IDE_TERMS; ++i) {
      const int l = ide_l(i), m = ide_m(i);
      float poly = 0.0f;
#pragma unroll
      for (int k = 0; k < RC_IDE_ZPOW; ++k) {
        // structurally non-zero coefficients only: k <= l - m and (l - m - k) even
        if (k <= l - m && ((l - m - k) & 1) == 0) poly = poly + zp[k] * tb->coef[i][k];
      }
      const float att = expf(-(0.5f * (float)(l * (l + 1))) * rough);
      act[(kStepIde + i) * 64] = (cpw[m] * poly) * att;
    }
  }
  RC_TSTAMP(2);
  // ---- SLF layer_0 (tiles 0-3) + input part of layer_bottleneck (tiles 4-7), bottleneck folded in: one pass
  //      over [feature | IDE | bias]; results stay in registers while the IBRDF chain runs.
  f32x16 s0[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) s0[t] = zero16();
  mlp_layer<8, 85, F0 + ShaderFrags::F_S0, NF>(ws, act, s0);
  RC_TSTAMP(3);
  // ---- integrated BRDF: (bottleneck, n.v) 129 -> 64 -> 64 -> 1 (nerf.py:461-482), first layer on the feature
  float ibrdf;
  {
    f32x16 ib[2];
    ib[0] = zero16(); ib[1] = zero16();
    // IDE is dead after s0': its first step becomes the (n.v | bias) step right behind the 48 feature steps
    act[48 * 64] = h == 0 ? dot_nv : 1.0f;
    mlp_layer<2, 49, F0 + ShaderFrags::F_I0, NF>(ws, act, ib);
    // steps [48, 81) are scratch for the IBRDF tail
    park<2, true>(ib, act, 48);
    act[(48 + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    ib[0] = zero16(); ib[1] = zero16();
    mlp_layer<2, 33, F0 + ShaderFrags::F_I1, NF>(ws, act + 48 * 64, ib);
    float o[1], nokeep[1];
    dot_out1<2, 1, F0 + ShaderFrags::F_IO, NF>(ws, ib, o, nokeep);     // output_integrated_brdf_layer on relu(ib)
    ibrdf = sigmoidf(o[0] + 1.0986123f);        // + log(3), nerf.py:481
  }
  RC_TSTAMP(4);
  // ---- SLF trunk: layer_1, layer_2, layer_bottleneck (x part accumulates onto the input part)
  float amb[3];
  {
    f32x16 acc[4] = {s0[0], s0[1], s0[2], s0[3]};
    f32x16 skip[4] = {s0[4], s0[5], s0[6], s0[7]};
    park<4, true>(acc, act, 0);
    act[64 * 64] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    mlp_layer<4, 65, F0 + ShaderFrags::F_S1, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    RC_TSTAMP(5);
    mlp_layer<4, 65, F0 + ShaderFrags::F_S2, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
    RC_TSTAMP(6);
    mlp_layer<4, 64, F0 + ShaderFrags::F_SB, NF>(ws, act, skip);
    RC_TSTAMP(7);
    float o[3], nokeep[1];
    dot_out1<4, 3, F0 + ShaderFrags::F_SO, NF>(ws, skip, o, nokeep);   // output_ambient_rgb_layer on relu(layer_bottleneck)
#pragma unroll
    for (int c = 0; c < 3; ++c) amb[c] = fmaxf(softplus(o[c] + k.slf_ambient_bias), 0.0f);      // slf.py:1053-1059
  }
  RC_TSTAMP(8);
  ShadeOut o;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // nerf.py:1034-1053; ambient_specular is an exact 0 (ref_acc == 1)
    const float is = fminf(fmaxf(tint[c] * ibrdf * (amb[c] * 1.0f), 0.0f), k.rgb_max);
    const float ambient = ad[c] + 0.0f;
    const float indirect = idf[c] + is;
    o.rgb[c] = ambient + indirect; o.ad[c] = ad[c]; o.idf[c] = idf[c]; o.is[c] = is; o.tint[c] = tint[c];
  }
  return o;
}

}  // namespace rcdev
