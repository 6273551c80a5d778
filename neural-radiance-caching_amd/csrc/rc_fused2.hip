// Fused cache forward for primary rays, TWO wavefronts per ray and TWO workgroups per CU: the plain cache pass of
// BASELINE configs 1, 2, 4 in builds with the fp32-MFMA shader (RC_SPLIT_MFMA=0).  The default build runs the shader on
// the bf16 pipe (rc_dev_mlp.h, split form) and the one-wavefront-per-ray kernel of rc_fused.hip for this pass too: the
// kernel was unstable with the split form in every layer (INSTABILITY there; RC_TEAM_SPLIT=1 puts it back for
// diagnosis).  rc_fused.hip is also the transient front end and the material stage's export.
//
// Why: with one wavefront per ray a 1024-ray batch is one wave per SIMD, and everything a wave waits for -- the
// texture-address unit working through 10 752 divergent lane requests per ray (the gather phases run at the CU's ~1
// lane-address per clock, profiles/r03_two_wave_probe.txt), LDS operand latency, the weight ring's barriers, MFMA
// result latency in front of every park -- is time in which its SIMD issues nothing: 86 us of issue in a 135 us kernel
// (profiles/pmc_k_cache_fused.json).  Here a workgroup is 4 waves = 2 rays x 2 waves, needs <= 80 KiB of LDS and <= 256
// registers, so two workgroups share a CU: every SIMD hosts two waves of DIFFERENT workgroups -- different rings,
// different barriers, free to drift -- and one's waits are the other's issue slots.
//
// How a ray is split between its two waves (q = 0, 1) -- "feature split": same weight stream, same fragment order, same
// MFMA order per accumulator as rc_fused.hip / the stand-alone kernels, hence the same bits:
//   proposal levels 0 / 1 (64 samples): wave q owns samples [32 q, 32 q + 32) end to end -- lookup (the two half-waves
//     split the grid levels by parity, as rc_level.hip does), density MLP on one point tile;
//   level 2 + shader (32 samples): the lookup is split by grid level (wave q: levels 4 q .. 4 q + 3 of both grids), every
//     MLP layer by OUTPUT TILE (wave q computes half of the layer's 32-row tiles for all 32 samples; the activations live
//     in the ray's LDS slice both waves read); one-tile layers (heads, the last backward layer) run on wave 0 while
//     wave 1 evaluates the directional-encoding polynomials;
//   the per-ray scans (alpha weights, step-function resampling, compositing) run on wave 0.
// Hand-offs between the two waves go through LDS and a workgroup barrier (4 waves); every wave of the workgroup executes
// the same sequence of barriers (ring seams included: a wave that does not read a fragment range still walks its seams).
#include "rc_fused_common.h"
#include <type_traits>
#include <utility>

using namespace rcdev;
using namespace rcfused;

namespace {

// Rays per workgroup.  2 (the product): 4 waves, 79.8 KiB of LDS, TWO workgroups per CU, each with its own 24 KiB ring.
// 4 (experiment, VERDICT r3 item 4: `make diag DIAG_EXTRA="-DRC_TEAM_RAYS=4 -DRC_TCH=96"`): ONE workgroup of 8 waves per
// CU on ONE ring of 48 KiB -- the weight stream crosses L2 -> LDS once per CU instead of twice and has half the seams, but
// every hand-off and seam is then a barrier of eight waves (four rays in lockstep).  Measured: profiles/r04_team_rays.txt.
#ifndef RC_TEAM_RAYS
#define RC_TEAM_RAYS 2
#endif
constexpr int kRays = RC_TEAM_RAYS;
constexpr int kTW = 2 * kRays;                // waves per workgroup: kRays rays x 2 waves
#ifndef RC_TCH
#define RC_TCH 48
#endif
constexpr int kTCH = RC_TCH;                  // fragments per ring chunk (48: 12 KiB, the ring is 24 KiB)
constexpr int kTRing = 2 * kTCH * 64;         // floats
constexpr int kTScratch = 9 * 68;             // per ray: the 7 step-function arrays + density exchange + small hand-offs
constexpr int kNF = NF_FUSED;

// Hand-off barrier between the two waves of a ray (all four waves of the workgroup take it).  Its release fence also
// drains vmcnt, i.e. waits for the ring's LDS-DMA in flight; a barrier that only waits for this wave's LDS writes
// ("s_waitcnt lgkmcnt(0); s_barrier") was measured 1.2 % SLOWER (127.6 vs 126.1 us per launch).
#ifdef RC_STAMPS
// diagnostic build: cycles this wave spends in hand-off barriers (arrival skew + the fence's drain), summed per wave
#define TB() do { const unsigned long long tb0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tb_wait += __builtin_amdgcn_s_memtime() - tb0_; ++tb_count; } while (0)
#else
#define TB() __syncthreads()
#endif

// ---- ring helpers with this kernel's geometry
__device__ __forceinline__ float tw_read(const WStream& w, int f) { return ws_read<kNF, kTW, kTCH>(w, f); }
// walk the chunk seams of fragments [F0, F0 + COUNT) without reading (the waves that do read hit the same barriers)
template <int F0, int COUNT>
__device__ __forceinline__ void tw_skip(const WStream& w) {
#pragma unroll
  for (int f = F0; f < F0 + COUNT; ++f)
    if (f > 0 && f % kTCH == 0) ws_advance<kNF, kTW, kTCH>(w, f / kTCH);
}

// which wave (q) owns global tile t of a layer of NT tiles, and in which of its accumulators
template <int NT> struct TileMap;
template <> struct TileMap<2> { static constexpr int NTL = 1, QS = 1; static constexpr int owner(int t) { return t; } static constexpr int slot(int) { return 0; } };
template <> struct TileMap<4> { static constexpr int NTL = 2, QS = 2; static constexpr int owner(int t) { return t >> 1; } static constexpr int slot(int t) { return t & 1; } };
// 8 tiles = [SLF layer_0: 0-3 | input part of layer_bottleneck: 4-7]: wave q takes {2 q, 2 q + 1} of each
template <> struct TileMap<8> { static constexpr int NTL = 4, QS = 2; static constexpr int owner(int t) { return (t >> 1) & 1; } static constexpr int slot(int t) { return (t & 1) + 2 * (t >> 2); } };

// mlp_layer (rc_dev_mlp.h) with the layer's NT output tiles split between the two waves of a ray: wave q accumulates
// the TileMap<NT>::NTL tiles it owns.  Fragment order in the stream is unchanged ([step][tile]); a slot's two candidate
// fragments (wave 0's tile t, wave 1's tile t + QS) are read with ONE runtime-offset read when they sit in the same ring
// chunk, and under a wave-uniform branch each when a seam separates them (a fragment of chunk c may only be read
// between the barriers that open chunks c and c + 1).
// Split form (rc_pack_host.h RC_SPLIT_MFMA): cells (block of 8 k-steps, tile) of three 1-KiB pieces in stream order
// [block][tile][piece]; a slot's two candidate cells are QS * 12 fragments apart.  Every wave walks every seam.  Operand
// registers follow the rules of rc_dev_mlp.h (INSTABILITY): the pieces of a block are retired -- and the next loads into their
// registers issued -- behind the first MFMAs of the block after it; NBUF register sets of pieces (three when a wave owns
// one tile of the layer: the loads then run two blocks ahead), two of activation pieces; two flush MFMAs at the end.
template <int NT, int KS, int FBASE>
__device__ __forceinline__ void mlp_layer_team_split(const WStream& w, int q, const float* act, f32x16 (&acc)[TileMap<NT>::NTL]) {
  using TM = TileMap<NT>;
  constexpr int NTL = TM::NTL;
  constexpr int NB = (KS + 7) / 8;
  constexpr int NBUF = NTL == 1 ? 3 : 2, D = NBUF - 1;
  static_assert(FBASE % 4 == 0 && kTCH % 4 == 0, "split layers start on a 1-KiB piece");
  u32x4 a[NBUF][NTL][3], b[2][3];
  float bv[8];
  auto piece = [&](int fr, int off) { return lds_piece(w.ring + ((fr % (2 * kTCH)) + off) * 64 + w.lane * 4); };
  auto load_b = [&](auto BLK) {
    constexpr int blk = decltype(BLK)::value;
    static_for<8>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (8 * blk + j < KS) bv[j] = act[(8 * blk + j) * 64];
      else bv[j] = 0.0f;
    });
  };
  auto load_a = [&](auto BLK) {
    constexpr int blk = decltype(BLK)::value, buf = blk % NBUF;
    static_for<NT>([&](auto T) {
      constexpr int t = decltype(T)::value, sl = TM::slot(t);
      static_for<3>([&](auto P) {
        constexpr int p = decltype(P)::value, f = FBASE + ((blk * NT + t) * 3 + p) * 4;
        if constexpr (f > 0 && f % kTCH == 0) ws_advance<kNF, kTW, kTCH>(w, f / kTCH);
        if constexpr (TM::owner(t) == 0) {
          constexpr int f1 = f + TM::QS * 12;          // wave 1's piece of the same slot
          if constexpr (f / kTCH == f1 / kTCH) a[buf][sl][p] = piece(f, q * TM::QS * 12);
          else { if (q == 0) a[buf][sl][p] = piece(f, 0); }
        } else {
          constexpr int f0 = f - TM::QS * 12;
          if constexpr (f / kTCH != f0 / kTCH) { if (q == 1) a[buf][sl][p] = piece(f, 0); }
        }
      });
    });
  };
  load_b(std::integral_constant<int, 0>{});
  static_for<D>([&](auto I) { if constexpr (decltype(I)::value < NB) load_a(I); });
  split8(bv, b[0]);
  if constexpr (NB > 1) load_b(std::integral_constant<int, 1>{});
  static_for<NB>([&](auto BLK) {
    constexpr int blk = decltype(BLK)::value;
    mfma_split6(a[blk % NBUF][0], b[blk & 1], acc[0]);
    // this block's first MFMAs have issued: every MFMA of the block before has started, its registers may go
    if constexpr (blk >= 1) {
#pragma unroll
      for (int t = 0; t < NTL; ++t) split_keep3(a[(blk - 1) % NBUF][t]);
      split_keep3(b[(blk - 1) & 1]);
    }
    if constexpr (blk + D < NB) load_a(std::integral_constant<int, blk + D>{});
    if constexpr (blk + 1 < NB) {
      split8(bv, b[(blk + 1) & 1]);
      if constexpr (blk + 2 < NB) load_b(std::integral_constant<int, blk + 2>{});
    }
#pragma unroll
    for (int t = 1; t < NTL; ++t) mfma_split6(a[blk % NBUF][t], b[blk & 1], acc[t]);
  });
  split_flush(w, b[(NB - 1) & 1][0]);
#pragma unroll
  for (int t = 0; t < NTL; ++t) split_keep3(a[(NB - 1) % NBUF][t]);
  split_keep3(b[(NB - 1) & 1]);
}

template <int NT, int KS, int FBASE, int SG = (TileMap<NT>::NTL >= 4 ? 2 : (TileMap<NT>::NTL >= 2 ? 4 : 8))>
__device__ __forceinline__ void mlp_layer_team(const WStream& w, int q, const float* act, f32x16 (&acc)[TileMap<NT>::NTL]) {
  if constexpr (kRcSplit && FBASE >= F_SH) { mlp_layer_team_split<NT, KS, FBASE>(w, q, act, acc); return; }      // density MLPs: fp32 in every build
  using TM = TileMap<NT>;
  constexpr int NTL = TM::NTL;
  constexpr int NG = (KS + SG - 1) / SG;
  float a[3][SG][NTL], b[3][SG];
  auto load = [&](auto G, auto BUF) {
    constexpr int g = decltype(G)::value, buf = decltype(BUF)::value;
    static_for<SG>([&](auto D) {
      constexpr int d = decltype(D)::value, s = g * SG + d;
      if constexpr (s < KS) {
        b[buf][d] = act[s * 64];
        static_for<NT>([&](auto T) {
          constexpr int t = decltype(T)::value, f = FBASE + s * NT + t, sl = TM::slot(t);
          if constexpr (f > 0 && f % kTCH == 0) ws_advance<kNF, kTW, kTCH>(w, f / kTCH);
          if constexpr (TM::owner(t) == 0) {
            constexpr int f1 = f + TM::QS;             // wave 1's fragment of the same slot
            if constexpr (f / kTCH == f1 / kTCH) a[buf][d][sl] = w.ring[((f % (2 * kTCH)) + q * TM::QS) * 64 + w.lane];
            else { if (q == 0) a[buf][d][sl] = w.ring[(f % (2 * kTCH)) * 64 + w.lane]; }
          } else {
            constexpr int f0 = f - TM::QS;
            if constexpr (f / kTCH != f0 / kTCH) { if (q == 1) a[buf][d][sl] = w.ring[(f % (2 * kTCH)) * 64 + w.lane]; }
          }
        });
      }
    });
  };
  auto comp = [&](auto G, auto BUF) {
    constexpr int g = decltype(G)::value, buf = decltype(BUF)::value;
    static_for<SG>([&](auto D) {
      constexpr int d = decltype(D)::value;
      if constexpr (g * SG + d < KS) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][d][t], b[buf][d], acc[t], 0, 0, 0);
      }
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  load(I0{}, I0{});
  if constexpr (NG > 1) load(I1{}, I1{});
  static_for<NG>([&](auto G) {
    constexpr int g = decltype(G)::value;
    if constexpr (g + 2 < NG) load(std::integral_constant<int, g + 2>{}, std::integral_constant<int, (g + 2) % 3>{});
    __builtin_amdgcn_sched_barrier(0);
    comp(G, std::integral_constant<int, g % 3>{});
    __builtin_amdgcn_sched_barrier(0);
  });
}

// park N accumulator tiles as activation steps [base, base + 16 N) of this lane's column (base may be a runtime value)
template <int N, bool RELU>
__device__ __forceinline__ void park_n(const f32x16* acc, float* act, int base) {
#pragma unroll
  for (int t = 0; t < N; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[t][r];
      act[(base + t * 16 + r) * 64] = RELU ? relu0(v) : v;
    }
}

// dot_out (rc_dev_mlp.h) on hidden activations that are PARKED (ReLU applied) in the ray's LDS slice: `hid` is this
// lane's column at the first hidden step.  Same products in the same order as the in-register form.  KEEP: the weights of
// output 0 for the tile this wave owns (keep[r] = w_0[feature(q, r, half-wave)]).
template <int NT, int NO, int FBASE, bool KEEP = false>
__device__ __forceinline__ void dot_out_lds(const WStream& w, int q, const float* hid, float (&out)[NO], float (&keep)[KEEP ? 16 : 1]) {
  float hv[NT * 16];
#pragma unroll
  for (int s = 0; s < NT * 16; ++s) hv[s] = hid[s * 64];
  float part[NO][2];
#pragma unroll
  for (int o = 0; o < NO; ++o) { part[o][0] = 0.0f; part[o][1] = 0.0f; }
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float wv = tw_read(w, FBASE + (o * NT + t) * 16 + r);
        if constexpr (KEEP) { if (o == 0) keep[r] = (t == q) ? wv : keep[r]; }
        part[o][r & 1] = __builtin_fmaf(hv[t * 16 + r], wv, part[o][r & 1]);
      }
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float bias = tw_read(w, FBASE + NO * NT * 16 + o);
    float sum = part[o][0] + part[o][1];
    sum = sum + __shfl_xor(sum, 32, 64);
    out[o] = sum + bias;
  }
}

// One proposal level's lookup + density MLP on this wave's 32 samples (rc_level.hip's LevelK::tile with the weights
// coming through the ring): the two half-waves split the grid levels by parity.  Returns the raw density of sample j.
template <int NL, int FB, int G>
__device__ __forceinline__ float level_tile(const RcFusedArgs& a, const WStream& ws, float* act_tile, int lane, float cx, float cy, float cz,
                                            unsigned long long* stamp = nullptr) {
  constexpr int KS0 = (NL + 1) / 2 + 1;
  using FR = DensFrags<KS0>;
  const int hh = lane >> 5;
  float* act = act_tile + lane;
  const RcGridDev& grid = a.grid[G];
  const float ux = unit_box(grid.bbox, cx), uy = unit_box(grid.bbox, cy), uz = unit_box(grid.bbox, cz);
  // pair i = grid levels (2 i, 2 i + 1): loads split by corner between the half-waves (rc_dev_grid.h pair_fetch), kinds
  // compile-time (the fused plan is compiled for kFusedDense leading dense levels, fused_geometry_ok), level records in
  // scalar registers -- straight-line code, every load of the tile in flight before the first combine
  constexpr int NH = (NL + 1) / 2;
  PairCorners P[NH];
  static_for<NH>([&](auto I) {
    constexpr int i = decltype(I)::value, la = 2 * i, lb = 2 * i + 1;
    constexpr int ka = ref_level_kind(la, kFusedDense);
    constexpr int kb = lb >= NL ? kLevelNone : ref_level_kind(lb, kFusedDense);
    const RcGridLevel &LA = grid.lvl[la], &LB = grid.lvl[lb < NL ? lb : la];
    // a.cell_table: the cell table of a dense level, the cell records of a kLevelHRec one
    pair_fetch<ka, kb>(ka != kLevelHashed ? a.cell_table[G][la] : LA.table, LA.size, LA.mask,
                       (kb == kLevelCell || kb == kLevelHRec) ? a.cell_table[G][lb < NL ? lb : la] : LB.table, LB.size, LB.mask, hh, ux, uy, uz, P[i]);
  });
  __builtin_amdgcn_sched_barrier(0);
  // feature l of point j -> step l / 2, half l & 1 = hh: this lane's own column
  static_for<KS0 - 1>([&](auto I) {
    constexpr int i = decltype(I)::value;
    float v = 0.0f;
    if constexpr (i < NH) {
      Corners<1> C;
      pair_finish<(2 * i + 1 < NL)>(P[i], C);
      float f[1], jd[1];
      grid_combine<1, false>(C, f, jd);
      v = (2 * i + 1 < NL || hh == 0) ? f[0] * grid.precondition : 0.0f;
    }
    act[i * 64] = v;
  });
  act[(KS0 - 1) * 64] = hh == 0 ? 1.0f : 0.0f;
  lds_sync<false>();
#ifdef RC_STAMPS
  if (stamp) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); *stamp = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
  f32x16 acc[2];
  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, KS0, FB + FR::D0, kNF, 4, kTW, kTCH>(ws, act, acc);
  park<2, true>(acc, act, 0);
  act[32 * 64] = hh == 0 ? 1.0f : 0.0f;
  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, 33, FB + FR::D1, kNF, 4, kTW, kTCH>(ws, act, acc);
  float out[1], nokeep[1];
  dot_out1<2, 1, FB + FR::DO, kNF, false, kTW, 1, kTCH>(ws, acc, out, nokeep);
  return out[0];
}

template <bool GRAD>
__global__ __launch_bounds__(kTW * 64, kRays == 2 ? 2 : 1) void k_cache_fused_team(RcFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // ray slot of the workgroup, role half of the ray.  Wave 0 of a ray carries the single-wave phases (scans, heads,
  // compositing): prio_mode bit 3 (experiment) swaps the roles in the second half of the grid, so that a SIMD shared by
  // two workgroups does not host two such waves (waves go to SIMDs by their index)
  // kRays == 4: waves w and w + 4 share a SIMD -- (ray, role) = ((w + w / 4) % 4, w / 4) puts two DIFFERENT rays in
  // different roles there
  const int rs = kRays == 2 ? (wave & 1) : ((wave + (wave >> 2)) & 3);
  const int q = kRays == 2 ? ((wave >> 1) ^ (((a.prio_mode & 8) && blockIdx.x >= (gridDim.x >> 1)) ? 1 : 0)) : (wave >> 2);
  int64_t ray = (int64_t)blockIdx.x * kRays + rs;
  const bool ray_ok = ray < a.n;
  if (!ray_ok) ray = a.n - 1;                        // keep the wave in the workgroup's lockstep
  float* ring = lds_dyn;
  float* act_ray = lds_dyn + kTRing + rs * (kShActSteps * 64);
  float* scr = lds_dyn + kTRing + kRays * (kShActSteps * 64) + rs * kTScratch;
  float* s_sd[2] = {scr, scr + 68};
  float* s_td = scr + 2 * 68; float* s_cw = scr + 3 * 68; float* s_c = scr + 4 * 68; float* s_v = scr + 5 * 68;
  float* s_out = scr + 6 * 68;
  float* x_dens = scr + 7 * 68;                      // [64] densities of a proposal level, from both waves to wave 0
  float* x_misc = scr + 8 * 68;                      // [32] roughness of the shaded samples, from wave 0 to wave 1
  WStream ws;
  ws.g = a.wstream; ws.ring = ring; ws.lane = lane; ws.wave = wave;
#ifdef RC_STAMPS
  unsigned long long stamps[16];
  unsigned long long tb_wait = 0, tb_count = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // Stagger (see rc_launch_fused_team): the workgroups that share a CU run the same phases at the same time -- all of
  // them in the lookups (bound by the memory system's random-sector rate, SIMDs idle), then all of them in the matrix
  // phases.  Every second workgroup to ARRIVE on a physical CU starts `stagger_cycles` late, so one half of the batch
  // gathers while the other half multiplies.  Which workgroups share a CU is the dispatcher's business: the arrival
  // counter is indexed by the hardware's own CU identity.
  if (a.stagger_cycles > 0) {
    int late = 0;
    if (threadIdx.x == 0) {
      const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: CU_ID 11:8, SH_ID 12, SE_ID 15:13
      const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // HW_REG_XCC_ID 3:0
      const uint32_t key = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
      late = atomicAdd(&a.cu_slots[key], 1) & 1;
      lds_dyn[0] = __int_as_float(late);
    }
    __syncthreads();
    late = __float_as_int(lds_dyn[0]);
    __syncthreads();
    if (late) {
      const unsigned long long t_start = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t_start < (unsigned long long)a.stagger_cycles) __builtin_amdgcn_s_sleep(16);
    }
  }
  // Priorities.  The workgroup dispatched second onto a CU loses every issue arbitration to the older one (age): it
  // falls 25 us behind in the proposal levels and then runs the tail of its shader alone, its stalls exposed.  Mode 1
  // (default): the younger workgroup -- the second half of the grid, the dispatcher deals the first half one per CU --
  // runs at s_setprio 1 up to its shader; the two then reach their matrix phases 20 us apart the other way round and
  // end within 2 us of each other (129.9 -> 128.4 us per 1024-ray launch; 2 / 4: always the younger / the older, and
  // 3: the younger in the shader only, measured 129.9 / 129.9 / 128.5).  Wrong guesses about who shares a CU cost nothing.
  const bool young = blockIdx.x >= (gridDim.x >> 1);
  const int pm = a.prio_mode & 7;
  if ((young && (pm == 1 || pm == 2)) || pm == 5) __builtin_amdgcn_s_setprio(1);
  if (!young && pm == 4) __builtin_amdgcn_s_setprio(1);
  RC_FSTAMP(0);
  ws_begin<kNF, kTW, kTCH>(ws);

  const float ox = a.origins[3 * ray], oy = a.origins[3 * ray + 1], oz = a.origins[3 * ray + 2];
  const float dx = a.directions[3 * ray], dy = a.directions[3 * ray + 1], dz = a.directions[3 * ray + 2];
  const float near = a.near[ray], far = a.far[ray];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);

  // (wave 0) resample S intervals from (prev sdist in s_prev [P+1], logit per bin): sdist -> s_out, tdist -> s_td
  auto resample = [&](int level, int P, int S, float logit, const float* s_prev) {
    const bool hasj = a.jitter[level] != nullptr;
    const float jit = hasj ? a.jitter[level][ray] : 0.0f;
    sample_intervals_wave<false>(logit, P, S, a.us[level], hasj, jit, s_prev, s_cw, s_c, s_v, s_out, lane);
    for (int e2 = lane; e2 <= S; e2 += 64) s_td[e2] = s_out[e2] * far + (1.0f - s_out[e2]) * near;   // coord.py:259-260
    lds_sync<false>();
  };
  auto mean_of = [&](int idx, float& mx, float& my, float& mz, float& t0, float& t1) {
    t0 = s_td[idx]; t1 = s_td[idx + 1];
    const float sm = t0 + t1, d = t1 - t0;
    const float ratio = (d * d) / fmaxf(RC_EPS * RC_EPS, 3.0f * (sm * sm) + d * d);
    const float tm = sm * (0.5f + ratio);
    mx = dx * tm + ox; my = dy * tm + oy; mz = dz * tm + oz;
  };
  auto density_of = [&](float raw, float cx, float cy, float cz, float bbox) {
    const bool inside = (cx > -bbox) & (cx < bbox) & (cy > -bbox) & (cy < bbox) & (cz > -bbox) & (cz < bbox);
    const float d = rc_safe_exp(raw + a.density_bias);
    return inside ? d : 0.0f;
  };
  const int j = lane & 31, h = lane >> 5;

  // ------------------------------------------------------------------ level 0 (P = 1, S = 64)
  if (q == 0) {
    if (lane == 0) { s_sd[0][0] = 0.0f; s_sd[0][1] = 1.0f; }
    lds_sync<false>();
    resample(0, 1, 64, a.anneal * safe_log(1.0f + a.padding), s_sd[0]);
  }
  TB();
  RC_FSTAMP(1);
  {
    float mx, my, mz, t0, t1;
    mean_of(32 * q + j, mx, my, mz, t0, t1);
    float cx = mx, cy = my, cz = mz;
    contract3(cx, cy, cz, a.contract_radius);
    const float raw = level_tile<6, F_L0, 0>(a, ws, act_ray + q * kTileStride, lane, cx, cy, cz
#ifdef RC_STAMPS
                                            , &stamps[2]
#endif
    );
    if (h == 0) x_dens[32 * q + j] = density_of(raw, cx, cy, cz, a.grid[0].bbox);
  }
  TB();
  RC_FSTAMP(3);
  // ------------------------------------------------------------------ level 1 (P = 64, S = 64)
  if (q == 0) {
    const float w = alpha_weight(x_dens[lane], s_td[lane], s_td[lane + 1], dnorm, true, lane);
    for (int e2 = lane; e2 <= 64; e2 += 64) s_sd[1][e2] = s_out[e2];
    lds_sync<false>();
    resample(1, 64, 64, a.anneal * safe_log(w + a.padding), s_sd[1]);
  }
  TB();
  RC_FSTAMP(4);
  {
    float mx, my, mz, t0, t1;
    mean_of(32 * q + j, mx, my, mz, t0, t1);
    float cx = mx, cy = my, cz = mz;
    contract3(cx, cy, cz, a.contract_radius);
    const float raw = level_tile<7, F_L1, 1>(a, ws, act_ray + q * kTileStride, lane, cx, cy, cz
#ifdef RC_STAMPS
                                            , &stamps[5]
#endif
    );
    if (h == 0) x_dens[32 * q + j] = density_of(raw, cx, cy, cz, a.grid[1].bbox);
  }
  TB();
  RC_FSTAMP(6);
  // ------------------------------------------------------------------ level 2 (P = 64, S = 32)
  if (q == 0) {
    const float w = alpha_weight(x_dens[lane], s_td[lane], s_td[lane + 1], dnorm, true, lane);
    for (int e2 = lane; e2 <= 64; e2 += 64) s_sd[0][e2] = s_out[e2];
    lds_sync<false>();
    resample(2, 64, 32, a.anneal * safe_log(w + a.padding), s_sd[0]);
  }
  TB();
  RC_FSTAMP(7);
  float mx, my, mz, t0, t1;
  mean_of(j, mx, my, mz, t0, t1);
  float cx = mx, cy = my, cz = mz;
  const float zx = rc_div(mx, a.contract_radius), zy = rc_div(my, a.contract_radius), zz = rc_div(mz, a.contract_radius);
  contract3(cx, cy, cz, a.contract_radius);
  float* act = act_ray + lane;
  auto jac_at = [&](int e) -> float& { return act_ray[(kJac + (e >> 1)) * 64 + j + 32 * (e & 1)]; };
  {
    // half-wave 0 looks up the level-2 density grid, half-wave 1 the appearance grid (interleaved pair tables);
    // wave q takes grid levels 4 q .. 4 q + 3: 32 corner loads of 16 bytes in flight per lane.  The body is expanded
    // once per role (QQ is a constant inside): level records, table pointers and the LDS slots of the features and of
    // the Jacobian are then constants / scalar loads instead of runtime address arithmetic on every store.
    const RcGridDev& g = a.grid[2];
    const float ux = unit_box(g.bbox, cx), uy = unit_box(g.bbox, cy), uz = unit_box(g.bbox, cz);
    auto lookup = [&](auto QQ) {
      constexpr int qq = decltype(QQ)::value;
      Corners<4> C[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int l = 4 * qq + k;
        const RcGridLevel& L = a.grid[2].lvl[l];
        grid_fetch<4, true, 2, true>(a.pair_table[l] + 4 * h, L.size, L.mask, 0u, L.dense != 0, ux, uy, uz, C[k]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int base = h == 0 ? 0 : kAppTmp;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int l = 4 * qq + k;
        const int size = a.grid[2].lvl[l].size;
        const bool dense = a.grid[2].lvl[l].dense != 0;
        float v[4], jd[GRAD ? 12 : 1];
        grid_combine<4, GRAD>(C[k], v, jd);
        // feature kf = 4 l + c of point j -> step base + kf / 2, lane j + 32 (kf & 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) act_ray[(base + 2 * l + (c >> 1)) * 64 + j + 32 * (c & 1)] = v[c] * g.precondition;
        if constexpr (GRAD) {
          const float s = g.precondition * (float)size / (2.0f * g.bbox);
          if (h == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              jac_at(0 * 32 + 4 * l + c) = (dense ? jd[2 * 4 + c] : jd[0 * 4 + c]) * s;
              jac_at(1 * 32 + 4 * l + c) = jd[1 * 4 + c] * s;
              jac_at(2 * 32 + 4 * l + c) = (dense ? jd[0 * 4 + c] : jd[2 * 4 + c]) * s;
            }
          }
        }
      }
    };
    if (q == 0) {
      lookup(std::integral_constant<int, 0>{});
      act[16 * 64] = h == 0 ? 1.0f : 0.0f;
    } else {
      lookup(std::integral_constant<int, 1>{});
    }
  }
  TB();
  RC_FSTAMP(8);
  // density MLP of the last level: each wave one of the two 32-row tiles of every layer
  using FR = DensFrags<17>;
  float density, npx, npy, npz, ngx = 0.0f, ngy = 0.0f, ngz = 0.0f;
  {
    f32x16 acc[1];
    acc[0] = zero16();
    mlp_layer_team<2, 17, F_L2 + FR::D0>(ws, q, act, acc);
    uint32_t m0 = 0, m1 = 0;
    if constexpr (GRAD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 |= (acc[0][r] > 0.0f ? 1u : 0u) << r;
    }
    TB();                                            // both waves are through with the input steps [0, 17)
    park_n<1, true>(acc, act, 16 * q);
    if (q == 0) act[32 * 64] = h == 0 ? 1.0f : 0.0f;
    TB();
    acc[0] = zero16();
    mlp_layer_team<2, 33, F_L2 + FR::D1>(ws, q, act, acc);
    if constexpr (GRAD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m1 |= (acc[0][r] > 0.0f ? 1u : 0u) << r;
    }
    TB();
    park_n<1, true>(acc, act, 16 * q);               // hidden feature: stays at [0, 32) for the shader
    TB();
    float out[4], wout[GRAD ? 16 : 1];
    dot_out_lds<2, 4, F_L2 + FR::DO, GRAD>(ws, q, act, out, wout);       // density + predicted normals, on both waves
    density = density_of(out[0], cx, cy, cz, a.grid[2].bbox);
    npx = out[1]; npy = out[2]; npz = out[3];
    neg_normalize(npx, npy, npz);
    // export (the material stage's primary pass picks its shading point from these): fence posts, density, sample means
    // and predicted normals of the last level into the workspace, as rc_fused.hip's EXPORT instantiation leaves them
    if (a.f_density && q == 0 && ray_ok) {
      const int64_t np = a.n * 32, p = ray * 32 + j;
      for (int e2 = lane; e2 <= 32; e2 += 64) a.f_tdist[ray * 33 + e2] = s_td[e2];
      if (h == 0) {
        a.f_density[p] = density;
        a.f_means[p] = mx; a.f_means[np + p] = my; a.f_means[2 * np + p] = mz;
        a.f_normals_pred[p] = npx; a.f_normals_pred[np + p] = npy; a.f_normals_pred[2 * np + p] = npz;
      }
    }
    if constexpr (GRAD) {
      // the backward pass borrows [0, 32): every wave keeps its half of the hidden feature in registers meanwhile
      float hid[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[r] = act[(16 * q + r) * 64];
      TB();                                          // dot_out_lds of both waves has read [0, 32)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(16 * q + r) * 64] = ((m1 >> r) & 1u) ? wout[r] : 0.0f;
      TB();
      f32x16 g1[1];
      g1[0] = zero16();
      mlp_layer_team<2, 32, F_L2 + FR::B1>(ws, q, act, g1);
      TB();
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(16 * q + r) * 64] = ((m0 >> r) & 1u) ? g1[0][r] : 0.0f;
      TB();
      if (q == 0) {
        f32x16 gf[1];
        gf[0] = zero16();
        mlp_layer_d<1, 32, F_L2 + FR::B0, kNF, 8, kTW, kTCH>(ws, act, gf);
        float gp[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) gp[ax] += gf[0][r] * jac_at(ax * 32 + i);
        }
        float gw[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float oth = __shfl_xor(gp[ax], 32, 64);
          gw[ax] = h == 0 ? gp[ax] + oth : oth + gp[ax];        // half 0's partial first on both halves
        }
        const float msq = zx * zx + zy * zy + zz * zz;
        float gzx = gw[0], gzy = gw[1], gzz = gw[2];
        if (msq > 1.0f) {
          const float rt = sqrtf(msq);
          const float s = (2.0f * rt - 1.0f) / msq;
          const float ds = (1.0f - rt) / (msq * msq);
          const float gz_dot = gw[0] * zx + gw[1] * zy + gw[2] * zz;
          gzx = s * gw[0] + 2.0f * ds * gz_dot * zx;
          gzy = s * gw[1] + 2.0f * ds * gz_dot * zy;
          gzz = s * gw[2] + 2.0f * ds * gz_dot * zz;
        }
        ngx = rc_div(gzx, a.contract_radius); ngy = rc_div(gzy, a.contract_radius); ngz = rc_div(gzz, a.contract_radius);
        neg_normalize(ngx, ngy, ngz);
      } else {
        tw_skip<F_L2 + FR::B0, rc_lfr32(32, 1)>(ws);
      }
      TB();                                          // wave 0 has read the backward activations
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(16 * q + r) * 64] = hid[r];
    } else {
      tw_skip<F_L2 + FR::B1, F_SH - (F_L2 + FR::B1)>(ws);      // the stream is consumed strictly in order
    }
  }
  RC_FSTAMP(9);
  if ((young && pm == 1) || pm == 5) __builtin_amdgcn_s_setprio(0);
  if (young && pm == 3) __builtin_amdgcn_s_setprio(1);
  // ------------------------------------------------------------------ shader on the 32 samples (rc_dev_mlp.h shader_tile)
  constexpr int F0 = F_SH;
  const float vx = a.viewdirs[3 * ray], vy = a.viewdirs[3 * ray + 1], vz = a.viewdirs[3 * ray + 2];
  const RcIdeTable* tb = reinterpret_cast<const RcIdeTable*>(a.ide_coef);
  const ShaderConsts& k = a.sh;
  // appearance features to their place behind the hidden feature: [0, 32) hidden | [32, 48) appearance | 48 bias
  if (q == 0) {
#pragma unroll
    for (int s = 0; s < 16; ++s) act[(32 + s) * 64] = act[(kAppTmp + s) * 64];
    act[48 * 64] = h == 0 ? 1.0f : 0.0f;
  }
  TB();
  float rough = 0.0f, tint[3] = {0.0f, 0.0f, 0.0f}, ad[3] = {0.0f, 0.0f, 0.0f}, idf[3] = {0.0f, 0.0f, 0.0f};
  const float dot_nv = npx * (-vx) + npy * (-vy) + npz * (-vz);        // nerf.py:474
  float ide_v[RC_IDE_TERMS];                         // (wave 1) Re / Im of the terms before the roughness attenuation
  if (q == 0) {
    // ---- small heads tile on the feature (the bottleneck is folded into its consumers, see kShActSteps)
    f32x16 acc[1];
    acc[0] = zero16();
    mlp_layer<1, 49, F0 + ShaderFrags::F_H, kNF, 8, kTW, kTCH>(ws, act, acc);
    rough = softplus(acc[0][0] + k.roughness_bias);                       // nerf.py:633-634
    tint[0] = sigmoidf(acc[0][1]); tint[1] = sigmoidf(acc[0][2]); tint[2] = sigmoidf(acc[0][3]);   // :976
    const float ar[3] = {acc[0][4], acc[0][5], acc[0][6]};
    const float ir[3] = {acc[0][7], acc[0][8], acc[0][9]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ad[c] = fminf(fmaxf(softplus(ar[c] + k.ambient_bias), 0.0f), k.rgb_max);      // nerf.py:965-969
      idf[c] = fminf(fmaxf(softplus(ir[c] + k.irradiance_bias), 0.0f), k.rgb_max);  // nerf.py:1008-1012
    }
    if (h == 0) x_misc[j] = rough;
  } else {
    // ---- reflection direction, IDE polynomials (ref_utils.py:155-190): low half-wave real parts, high half-wave
    //      imaginary parts; the attenuation by the roughness follows once wave 0 has it
    const float rx = 2.0f * dot_nv * npx - (-vx), ry = 2.0f * dot_nv * npy - (-vy), rz = 2.0f * dot_nv * npz - (-vz);
    float zp[RC_IDE_ZPOW];
    zp[0] = 1.0f;
#pragma unroll
    for (int p = 1; p < RC_IDE_ZPOW; ++p) zp[p] = zp[p - 1] * rz;
    float cpw[RC_IDE_ZPOW];
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
      for (int p = 0; p < RC_IDE_ZPOW; ++p)
        if (p <= l - m && ((l - m - p) & 1) == 0) poly = poly + zp[p] * tb->coef[i][p];
      ide_v[i] = cpw[m] * poly;
    }
    tw_skip<F0 + ShaderFrags::F_H, rc_lfr(49, 1)>(ws);          // the seams wave 0 crosses inside the heads layer
  }
  TB();
  if (q == 1) {
    const float rg = x_misc[j];
#pragma unroll
    for (int i = 0; i < RC_IDE_TERMS; ++i) {
      const int l = ide_l(i);
      const float att = expf(-(0.5f * (float)(l * (l + 1))) * rg);
      act[(kStepIde + i) * 64] = ide_v[i] * att;
    }
  } else {
    act[kStepBias * 64] = h == 0 ? 1.0f : 0.0f;
  }
  TB();
  // ---- SLF layer_0 (tiles 0-3) + input part of layer_bottleneck (tiles 4-7): wave q owns {2 q, 2 q + 1} of each;
  //      results stay in registers while the IBRDF chain runs
  f32x16 s0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) s0[t] = zero16();
  mlp_layer_team<8, 85, F0 + ShaderFrags::F_S0>(ws, q, act, s0);
  TB();                                              // the IDE steps are dead: step 48 becomes (n.v | bias)
  if (q == 0) act[48 * 64] = h == 0 ? dot_nv : 1.0f;
  TB();
  // ---- integrated BRDF: (bottleneck, n.v) 129 -> 64 -> 64 -> 1 (nerf.py:461-482), one tile per wave
  float ibrdf;
  {
    f32x16 ib[1];
    ib[0] = zero16();
    mlp_layer_team<2, 49, F0 + ShaderFrags::F_I0>(ws, q, act, ib);
    TB();
    park_n<1, true>(ib, act, 48 + 16 * q);           // steps [48, 81) are scratch for the IBRDF tail
    if (q == 0) act[(48 + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    TB();
    ib[0] = zero16();
    mlp_layer_team<2, 33, F0 + ShaderFrags::F_I1>(ws, q, act + 48 * 64, ib);
    TB();
    park_n<1, true>(ib, act, 48 + 16 * q);
    TB();
    float o[1], nokeep[1];
    dot_out_lds<2, 1, F0 + ShaderFrags::F_IO>(ws, q, act + 48 * 64, o, nokeep);      // output_integrated_brdf_layer
    ibrdf = sigmoidf(o[0] + 1.0986123f);        // + log(3), nerf.py:481
  }
  // ---- SLF trunk: layer_1, layer_2, layer_bottleneck (x part accumulates onto the input part), two tiles per wave
  float amb[3];
  {
    f32x16 acc[2] = {s0[0], s0[1]};
    f32x16 skip[2] = {s0[2], s0[3]};
    TB();                                            // the IBRDF tail has been read: [0, 64) takes layer_0's output
    park_n<2, true>(acc, act, 32 * q);
    if (q == 0) act[64 * 64] = h == 0 ? 1.0f : 0.0f;
    TB();
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_team<4, 65, F0 + ShaderFrags::F_S1>(ws, q, act, acc);
    TB();
    park_n<2, true>(acc, act, 32 * q);
    TB();
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_team<4, 65, F0 + ShaderFrags::F_S2>(ws, q, act, acc);
    TB();
    park_n<2, true>(acc, act, 32 * q);
    TB();
    mlp_layer_team<4, 64, F0 + ShaderFrags::F_SB>(ws, q, act, skip);
    TB();
    park_n<2, true>(skip, act, 32 * q);
    TB();
    float o[3], nokeep[1];
    dot_out_lds<4, 3, F0 + ShaderFrags::F_SO>(ws, q, act, o, nokeep);   // output_ambient_rgb_layer on relu(layer_bottleneck)
#pragma unroll
    for (int c = 0; c < 3; ++c) amb[c] = fmaxf(softplus(o[c] + k.slf_ambient_bias), 0.0f);      // slf.py:1053-1059
  }
  RC_FSTAMP(10);
  // ------------------------------------------------------------------ volume compositing (k_composite): no barrier
  // behind this point.  Both waves hold the density, the fence posts and the predicted normals of all 32 samples, so the
  // weighted sums are split: wave 0 the colours (it has the heads) and the analytic normals, wave 1 the geometry
  // (means, distances, predicted normals) and the distance percentiles.  Same sums in the same order as one wave.
  const bool act_s = lane < 32;
  const float wnf = alpha_weight(density, t0, t1, dnorm, act_s, lane);
  auto store3 = [&](int id, float x, float y, float z) {
    if (lane == 0 && ray_ok && a.out.ptr[id]) { a.out.ptr[id][3 * ray] = x; a.out.ptr[id][3 * ray + 1] = y; a.out.ptr[id][3 * ray + 2] = z; }
  };
  auto store1 = [&](int id, float x) { if (lane == 0 && ray_ok && a.out.ptr[id]) a.out.ptr[id][ray] = x; };
  if (q == 0) {
    ShadeOut so;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // nerf.py:1034-1053; ambient_specular is an exact 0 (ref_acc == 1)
      const float is = fminf(fmaxf(tint[c] * ibrdf * (amb[c] * 1.0f), 0.0f), k.rgb_max);
      const float ambient = ad[c] + 0.0f;
      const float indirect = idf[c] + is;
      so.rgb[c] = ambient + indirect; so.ad[c] = ad[c]; so.idf[c] = idf[c]; so.is[c] = is; so.tint[c] = tint[c];
    }
    enum { V_ACC = 0, V_RGB = 1, V_AD = 4, V_IDF = 7, V_IS = 10, V_TINT = 13, V_DIF = 16, V_IND = 19, V_NG = 22, V_COUNT = 25 };
    float v[V_COUNT];
    v[V_ACC] = wnf;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v_rgb = act_s ? so.rgb[c] : 0.0f, v_ad = act_s ? so.ad[c] : 0.0f, v_id = act_s ? so.idf[c] : 0.0f;
      const float v_is = act_s ? so.is[c] : 0.0f, v_t = act_s ? so.tint[c] : 0.0f;
      v[V_RGB + c] = wnf * v_rgb;
      v[V_AD + c] = wnf * v_ad;
      v[V_IDF + c] = wnf * v_id;
      v[V_IS + c] = wnf * v_is;
      v[V_TINT + c] = wnf * v_t;
      v[V_DIF + c] = wnf * (v_ad + v_id);
      v[V_IND + c] = wnf * (v_id + v_is);
    }
    v[V_NG] = wnf * ngx; v[V_NG + 1] = wnf * ngy; v[V_NG + 2] = wnf * ngz;
    wave_sum_n<V_COUNT>(v);
    const float accw = v[V_ACC];
    const float bgw = fmaxf(0.0f, 1.0f - accw) * a.bg;
    store3(RC_OUT_RGB, v[V_RGB] + bgw, v[V_RGB + 1] + bgw, v[V_RGB + 2] + bgw);
    store3(RC_OUT_DIRECT_RGB, v[V_AD], v[V_AD + 1], v[V_AD + 2]);
    store3(RC_OUT_INDIRECT_DIFFUSE_RGB, v[V_IDF], v[V_IDF + 1], v[V_IDF + 2]);
    store3(RC_OUT_INDIRECT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);
    store3(RC_OUT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);
    store3(RC_OUT_ALBEDO_RGB, v[V_TINT], v[V_TINT + 1], v[V_TINT + 2]);
    store3(RC_OUT_DIFFUSE_RGB, v[V_DIF], v[V_DIF + 1], v[V_DIF + 2]);
    store3(RC_OUT_INDIRECT_RGB, v[V_IND], v[V_IND + 1], v[V_IND + 2]);
    store3(RC_OUT_INDIRECT_OCC, accw, accw, accw);
    store1(RC_OUT_ACC, accw);
    if constexpr (GRAD) store3(RC_OUT_NORMALS, v[V_NG], v[V_NG + 1], v[V_NG + 2]);
  } else {
    enum { V_ACC = 0, V_MEAN = 1, V_RD = 4, V_LD = 5, V_NP = 6, V_LOGT = 9, V_COUNT = 10 };
    float v[V_COUNT];
    v[V_ACC] = wnf;
    v[V_MEAN] = wnf * mx; v[V_MEAN + 1] = wnf * my; v[V_MEAN + 2] = wnf * mz;
    v[V_RD] = wnf * sqrtf((ox - mx) * (ox - mx) + (oy - my) * (oy - my) + (oz - mz) * (oz - mz));
    v[V_LD] = 0.0f;
    if (a.lights) {
      const float lx = a.lights[3 * ray], ly = a.lights[3 * ray + 1], lz = a.lights[3 * ray + 2];
      v[V_LD] = wnf * sqrtf((lx - mx) * (lx - mx) + (ly - my) * (ly - my) + (lz - mz) * (lz - mz));
    }
    v[V_NP] = wnf * npx; v[V_NP + 1] = wnf * npy; v[V_NP + 2] = wnf * npz;
    v[V_LOGT] = act_s ? wnf * logf(0.5f * (t0 + t1)) : 0.0f;
    wave_sum_n<V_COUNT>(v);
    const float accw = v[V_ACC];
    store3(RC_OUT_MEANS, v[V_MEAN], v[V_MEAN + 1], v[V_MEAN + 2]);
    store1(RC_OUT_RAY_DISTS, v[V_RD]);
    if (a.lights) store1(RC_OUT_LIGHT_DISTS, v[V_LD]);
    store3(RC_OUT_NORMALS_PRED, v[V_NP], v[V_NP + 1], v[V_NP + 2]);
    const float e = v[V_LOGT] / fmaxf(RC_EPS, accw);
    float dm = expf(e);
    if (dm != dm) dm = 0.0f;                     // nan_to_num(x, copy=inf): nan -> 0.0 (rc_sample.hip, k_composite)
    dm = fminf(dm, RC_FMAX);
    dm = fminf(fmaxf(dm, s_td[0]), s_td[32]);
    store1(RC_OUT_DISTANCE_MEAN, dm);
    const float wn = wnf / fmaxf(RC_EPS, accw);
    const float incl = wave_scan_incl(wn, lane);
    if (lane == 0) s_cw[0] = 0.0f;
    if (lane < 31) s_cw[lane + 1] = fminf(1.0f, incl);
    if (lane == 0) s_cw[32] = 1.0f;
    lds_sync<false>();
    if (lane < 3 && ray_ok) {
      const float ps = a.pct[lane] / 100.0f;
      const float pv = interp1(ps, s_cw, s_td, 33);
      const int id = lane == 0 ? RC_OUT_DISTANCE_PERCENTILE_5 : (lane == 1 ? RC_OUT_DISTANCE_MEDIAN : RC_OUT_DISTANCE_PERCENTILE_95);
      if (a.out.ptr[id]) a.out.ptr[id][ray] = pv;
    }
#ifdef RC_STAMPS
    RC_FSTAMP(11);
    if (lane == 0 && a.stamps && ray_ok) {          // wave 1 of a ray: second half of the stamp buffer
      unsigned long long* d = a.stamps + (ray + a.n) * 16;
      for (int i = 0; i < 12; ++i) d[i] = stamps[i];
      d[14] = rt0; d[15] = __builtin_amdgcn_s_memrealtime();
      d[12] = tb_wait; d[13] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | (tb_count << 32);
    }
#endif
    return;
  }
  RC_FSTAMP(11);
#ifdef RC_STAMPS
  if (lane == 0 && a.stamps && ray_ok) {
    unsigned long long* d = a.stamps + (ray + (q ? a.n : 0)) * 16;      // wave 1 of a ray: second half of the buffer
    for (int i = 0; i < 12; ++i) d[i] = stamps[i];
    d[14] = rt0; d[15] = __builtin_amdgcn_s_memrealtime();
    d[12] = tb_wait; d[13] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | (tb_count << 32);
  }
#endif
}

}  // namespace

void rc_launch_fused_team(const RcFusedArgs& a, bool grad, hipStream_t stream) {
  static std::atomic<uint64_t> prepared{0};
  const int lds = (kTRing + kRays * (kShActSteps * 64 + kTScratch)) * (int)sizeof(float);
  static_assert((kTRing + kRays * (kShActSteps * 64 + kTScratch)) * sizeof(float) <= (kRays == 2 ? 80 : 160) * 1024, "two workgroups (one of eight waves) per CU");
  if (rc_first_use_on_device(prepared)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused_team<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused_team<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  dim3 grid((unsigned)((a.n + kRays - 1) / kRays)), block(kTW * 64);
  RcFusedArgs b = a;
  // the priority scheme reads "second half of the grid" as "dispatched second onto its CU": true while the whole grid is
  // resident at once (two workgroups per CU); beyond that workgroups start as others finish and the scheme costs
  // 2.4 % (16 384 rays: 1790 -> 1833 us), so it is switched off
  if ((int64_t)grid.x > 2 * (int64_t)rc_device_cus() || kRays != 2) b.prio_mode = 0;
  if (b.stagger_cycles > 0) {
    static std::atomic<int32_t*> slots{nullptr};       // one process drives one GPU (one handle per device)
    int32_t* p = slots.load();
    if (!p) {
      if (hipMalloc((void**)&p, 4096 * sizeof(int32_t)) != hipSuccess || hipMemset(p, 0, 4096 * sizeof(int32_t)) != hipSuccess) p = nullptr;
      slots.store(p);
    }
    b.cu_slots = p;
    if (!p) b.stagger_cycles = 0;
  }
  if (grad) hipLaunchKernelGGL(k_cache_fused_team<true>, grid, block, lds, stream, b);
  else hipLaunchKernelGGL(k_cache_fused_team<false>, grid, block, lds, stream, b);
}
