// One proposal level of the launch-per-stage plan as ONE launch: contraction + multiresolution grid lookup + density
// MLP (+ convert_raw_density) per sample, i.e. k_hashgrid_fwd and k_density_mlp fused for the case where only the
// density of a sample is consumed behind it (proposal levels 0 / 1 always; the last level on the lean resampling pass of
// secondary rays, whose hidden feature is recomputed for the one picked sample per ray).
//
// Replaces, with the arithmetic and the operation order of rc_hashgrid.hip / rc_mlp.hip (bitwise the same densities):
//   coord.contract, HashEncoding.__call__                           internal/coord.py:37-69, internal/grid_utils.py:808-905
//   DensityMLP.run_network, convert_raw_density                      internal/geometry.py:155-168, 318-341
//
// Two forms: k_level walks 32-sample tiles of a level whose sample means k_sample_level left in the workspace;
// k_level_ray takes a ray per wave, draws the level's samples itself (rc_dev_sample.h sample_level_ray, the body of
// k_sample_level) and walks the ray's tiles -- one launch per proposal level instead of two, the means never leave the
// registers on their way into the lookup (rc_api.hip picks it from 24 576 rays on: (55 + 290) -> 316, (55 + 330) -> 350,
// (50 + 390) -> 411 us on the trace below; at 1-4 k rays the per-ray form costs 27 us per pass).
//
// Why: on the material stage's batched secondary trace (32 768 rays, 2 M + 2 M + 1 M samples) the two kernels per level
// ran back to back -- with the grid features going through HBM in between, and the MLP at 36-54 % of the MFMA rate
// because every 128-point workgroup re-streamed the level's weights and paid a start-up.  Here the level's whole weight
// stream (27-59 KB) is loaded into LDS ONCE per workgroup (one per CU), whose waves then walk independently -- no
// barrier after the load -- over 32-sample tiles: gather (all corner loads of a lane's levels in flight), features
// straight into the MFMA B-operand slots, MLP, density.  The two half-waves of a tile split the grid levels by parity,
// so a lane issues half of a point's loads and, for the F = 1 grids, writes its features into its own B column.
// Measured on that trace (us per level, before -> after): 462 -> 302, 446 -> 344, 522 -> 389.  The gather half is
// largely vector-ALU work (contraction, hashing, 64-bit addresses, trilinear weights) and fp32 MFMA shares the vector
// ALUs, so the two halves mostly ADD inside a SIMD: gather alone 206 / 296 / 372 us, MLP alone 201 / 205 / 135 us.
// A producer / consumer split of the workgroup (one gather wave per grid level feeding four MLP waves through a
// double-buffered B-operand ring) was built as well, bitwise equal, and measured slower (343 / 467 / 405 us): it repeats
// the contraction per level and gains nothing from running the two halves side by side.
// Counters of the three launches on that trace (profiles/r02_material_pmc_counters.txt): the F = 4 level misses the L2 on
// 78 % of its 28 M requests -- 21.8 M 64-byte fabric reads = 1.4 GB in 390 us = 3.6 TB/s of random sectors, about one
// per hashed (y, z) corner pair, which is what the lookup needs -- and its texture-address unit is stalled by the cache
// for 73 % of the launch: that level runs at the memory side's random-sector rate.  The F = 1 levels read 7.8 / 14.4 M
// sectors (TA stalled 23 / 33 %) next to 132 us of MFMA time and the gather arithmetic: a mix of the three.
// Copies of the hashed F = 1 tables with the two x-corners of a cell adjacent (one 8-byte load for 15 of 16 lanes with
// four copies) were built, bitwise equal, and measured slower (+26 / +100 us): they multiply the footprint the L2 sees,
// and the number of L1 lookups was not the bound.
// Round 3: the lookup as the generated code really ran it had ONE level in flight per lane -- `grid.lvl[2 i + h]` with the
// lane-dependent h made every field of the level record a per-lane load from the argument segment, each behind an
// s_waitcnt vmcnt(0).  Rebuilt (tile() below, rc_dev_grid.h pair_fetch): level records as scalars, compile-time level
// kinds for the reference's layout (ND), the loads of a level pair split by corner between the half-waves.  The F = 1
// levels of the trace: 314 -> 284, 458 -> 415 us (NOT bitwise-affecting: same features, same order).
#include <stdlib.h>
#include "rc_dev_grid.h"
#include "rc_dev_mlp.h"
#include "rc_dev_sample.h"

using namespace rcdev;

namespace {

constexpr int kLvActSteps = 33;
#ifndef RC_LV1_WAVES
#define RC_LV1_WAVES 12
#endif
#ifndef RC_LV4_GROUP
#define RC_LV4_GROUP 4
#endif

template <int F> struct LevelCfg;
template <> struct LevelCfg<1> { static constexpr int W = RC_LV1_WAVES; };
template <> struct LevelCfg<4> { static constexpr int W = 8; };

struct RcLevelKArgs {
  RcGridDev grid;
  const float* means;          // SoA [3][n]
  int64_t n;
  const float* wstream;        // [d0 | d1 | out (+ ...)] fragments of rc_api.hip's "dens_<l>" pack
  float density_bias, contract_radius;
  float* density;              // [n]
};

// NL grid levels of F features: K = NL * F grid features, KS0 = K / 2 (rounded up) + 1 k-steps in layer 0.
// ND >= 0 (F = 1): the level layout is known at compile time -- levels [0, ND) dense with cell tables, the others hashed
// (the reference's layout has ND = 3) -- and the lookup is rc_dev_grid.h's pair_fetch: loads split by corner between the
// half-waves, no divergence.  ND = -1: any layout, the kind of a level read from its record.
template <int F, int NL, int ND = -1>
struct LevelK {
  static constexpr int W = LevelCfg<F>::W;
  static constexpr int K = F * NL, KS0 = (K + 1) / 2 + 1;
  static constexpr int NOB = KS0 == 17 ? 4 : 1;                            // rows the output layer was packed with
  static constexpr int F_D0 = 0, F_D1 = rc_lfr32(KS0, 2), F_DO = F_D1 + rc_lfr32(33, 2), NF = F_DO + rc_dfr32(NOB, 2);
  static constexpr int CH = (NF / (4 * W) + 1) * 4 * W;   // > NF and a multiple of 4 W (ws_issue is instantiated, never run): the whole stream is one resident chunk
  static_assert(NF <= CH && CH % (4 * W) == 0, "stream must fit the resident chunk");
  static constexpr int kResFloats = ((NF + 3) / 4) * 4 * 64;               // [NF padded to 4][64]

  // the level's weights: once per workgroup, LDS-DMA in 1-KiB pieces (the packed stream is padded to whole 16-KiB chunks)
  static __device__ __forceinline__ void load_weights(const float* wstream, float* wres, int wave, int lane) {
    for (int piece = wave; piece < (NF + 3) / 4; piece += W)
      __builtin_amdgcn_global_load_lds((const void*)(wstream + (size_t)piece * 256 + lane * 4), (lds_void_ptr)(wres + piece * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // One 32-point tile on this wave's activation slice: contraction, grid lookup, density MLP, convert_raw_density.
  // (cx, cy, cz): the position of point j = lane & 31, the same on both half-waves.  Returns the density (all lanes).
  static __device__ __forceinline__ float tile(const RcGridDev& grid, float density_bias, float contract_radius, const WStream& ws,
                                               float* act_wave, int lane, float cx, float cy, float cz) {
    const int j = lane & 31, h = lane >> 5;
    float* act = act_wave + lane;
    const float bbox = grid.bbox;
    contract3(cx, cy, cz, contract_radius);
    const float ux = unit_box(bbox, cx), uy = unit_box(bbox, cy), uz = unit_box(bbox, cz);
    if constexpr (F == 1 && ND >= 0) {
      constexpr int NH = (NL + 1) / 2;
      PairCorners P[NH];
      static_for<NH>([&](auto I) {
        constexpr int i = decltype(I)::value, la = 2 * i, lb = 2 * i + 1 < NL ? 2 * i + 1 : 2 * i;
        constexpr int ka = ref_level_kind(la, ND);
        constexpr int kb = 2 * i + 1 >= NL ? kLevelNone : ref_level_kind(lb, ND);
        const RcGridLevel &LA = grid.lvl[la], &LB = grid.lvl[lb];
        pair_fetch<ka, kb>(ka == kLevelCell ? LA.cell : (ka == kLevelHRec ? LA.rec : LA.table), LA.size, LA.mask,
                           kb == kLevelCell ? LB.cell : (kb == kLevelHRec ? LB.rec : LB.table), LB.size, LB.mask, h, ux, uy, uz, P[i]);
      });
      __builtin_amdgcn_sched_barrier(0);
      // feature l of point j -> step l / 2, half l & 1 = h: this lane's own column
      static_for<KS0 - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        float v = 0.0f;
        if constexpr (i < NH) {
          Corners<1> C;
          pair_finish<(2 * i + 1 < NL)>(P[i], C);
          float f[1], jd[1];
          grid_combine<1, false>(C, f, jd);
          v = (2 * i + 1 < NL || h == 0) ? f[0] * grid.precondition : 0.0f;
        }
        act[i * 64] = v;
      });
    } else {
      // this half-wave's levels: l = 2 i + h (all their corner loads in flight before the first combine)
      constexpr int NH = (NL + 1) / 2;
      // pairs whose loads are in flight together: all of them for F = 1; RC_LV4_GROUP of the four for F = 4 (128 registers
      // of load destinations otherwise)
      constexpr int GRP = F == 1 ? NH : RC_LV4_GROUP;
      static_assert(NH % GRP == 0 || F == 1, "group size");
#pragma unroll
      for (int i0 = 0; i0 < NH; i0 += GRP) {
        Corners<F> C[GRP];
#pragma unroll
        for (int ii = 0; ii < GRP; ++ii) {
          const int i = i0 + ii;
          const int l = 2 * i + h;
          if (l < NL) {
            // Both level records of the pair (kernel arguments) into scalar registers, the half-wave's one selected in
            // registers: indexing grid.lvl[] with the lane-dependent l makes every field a per-lane global load from the
            // argument segment -- dependent round trips in front of the corner loads, and their s_waitcnt vmcnt(0) drains
            // the previous pair's corners, so a lane never had more than one level in flight.  Kind of a level: 2 = dense
            // with a cell table, 1 = dense, 0 = hashed (power-of-two tables only: rc_level_supported); a pair of one kind
            // branches wave-uniformly, a mixed pair runs both sides under exec masks.
            const RcGridLevel &L0 = grid.lvl[2 * i], &L1 = grid.lvl[2 * i + 1 < NL ? 2 * i + 1 : 2 * i];
            const int size = pick_half(h, L0.size, L1.size);
            const uint32_t mask = pick_half(h, L0.mask, L1.mask);
            if constexpr (F == 1) {
              const int k0 = L0.cell ? 2 : (L0.dense ? 1 : 0), k1 = L1.cell ? 2 : (L1.dense ? 1 : 0);
              const float* tab = pick_half(h, k0 == 2 ? L0.cell : L0.table, k1 == 2 ? L1.cell : L1.table);
              if (k0 == k1) {
                if (k0 == 2) grid_fetch_cell(tab, size, ux, uy, uz, C[ii]);
                else grid_fetch<1, true>(tab, size, mask, 0u, k0 == 1, ux, uy, uz, C[ii]);
              } else if ((k0 == 1) | (k1 == 1)) {
                grid_fetch<1, true>(pick_half(h, L0.table, L1.table), size, mask, 0u, pick_half(h, k0, k1) != 0, ux, uy, uz, C[ii]);   // a dense level without its cell table in the pair: plain tables on both sides
              } else {
                grid_fetch<1, true, 1, true>(tab, size, mask, 0u, pick_half(h, k0, k1) == 2, ux, uy, uz, C[ii]);
              }
            } else {
              // kind of a level: dense | hashed | hashed through its cell records (L.rec: 128 bytes per cell origin = one
              // cache line for the eight corners instead of ~4.4 sectors of the 8 MiB table)
              const bool d0 = L0.dense != 0, d1 = L1.dense != 0, r0 = L0.rec != nullptr, r1 = L1.rec != nullptr;
              const float* tab = pick_half(h, r0 ? L0.rec : L0.table, r1 ? L1.rec : L1.table);
              if (d0 == d1 && r0 == r1) grid_fetch<4, true>(tab, size, mask, 0u, d0, ux, uy, uz, C[ii], r0);
              else grid_fetch<4, true>(tab, size, mask, 0u, pick_half(h, d0, d1), ux, uy, uz, C[ii], pick_half(h, r0, r1));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F == 1) {
          // feature l of point j -> step l / 2, half l & 1 = h: this lane's own column
#pragma unroll
          for (int i = 0; i < KS0 - 1; ++i) {
            const int l = 2 * i + h;
            float v = 0.0f;
            if (i < NH && l < NL) {
              float f[1], jd[1];
              grid_combine<1, false>(C[i < NH ? i : 0], f, jd);
              v = f[0] * grid.precondition;
            }
            act[i * 64] = v;
          }
        } else {
          // F = 4: feature 4 l + c -> step 2 l + c / 2, half c & 1
#pragma unroll
          for (int ii = 0; ii < GRP; ++ii) {
            const int l = 2 * (i0 + ii) + h;
            if (l < NL) {
              float f[4], jd[1];
              grid_combine<4, false>(C[ii], f, jd);
#pragma unroll
              for (int c = 0; c < 4; ++c) act_wave[(2 * l + (c >> 1)) * 64 + j + 32 * (c & 1)] = f[c] * grid.precondition;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    act[(KS0 - 1) * 64] = h == 0 ? 1.0f : 0.0f;
    lds_sync_wave();
    f32x16 acc[2];
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_d<2, KS0, F_D0, NF, 4, W, CH>(ws, act, acc);
    park<2, true>(acc, act, 0);
    act[32 * 64] = h == 0 ? 1.0f : 0.0f;
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_d<2, 33, F_D1, NF, 4, W, CH>(ws, act, acc);
    float out[1], nokeep[1];
    dot_out1<2, 1, F_DO, NF, false, W, NOB, CH>(ws, acc, out, nokeep);     // output_density_layer on relu(acc)
    // convert_raw_density (geometry.py:318-341)
    const bool inside = (cx > -bbox) & (cx < bbox) & (cy > -bbox) & (cy < bbox) & (cz > -bbox) & (cz < bbox);
    const float d = rc_safe_exp(out[0] + density_bias);
    lds_sync_wave();        // the next tile's feature writes must not overtake this tile's activation reads
    return inside ? d : 0.0f;
  }
};

template <int F, int NL, int ND>
__global__ __launch_bounds__(LevelCfg<F>::W * 64) void k_level(RcLevelKArgs a) {
  using LK = LevelK<F, NL, ND>;
  constexpr int W = LK::W;
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float* wres = lds_dyn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* act_wave = lds_dyn + LK::kResFloats + wave * (kLvActSteps * 64);
  LK::load_weights(a.wstream, wres, wave, lane);
  WStream ws{a.wstream, wres, lane, wave};
  const int j = lane & 31, h = lane >> 5;
  const int64_t tiles = (a.n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * W + wave; tile < tiles; tile += (int64_t)gridDim.x * W) {
    const int64_t p = tile * 32 + j;
    const bool valid = p < a.n;
    const int64_t q = valid ? p : a.n - 1;
    const float d = LK::tile(a.grid, a.density_bias, a.contract_radius, ws, act_wave, lane, a.means[q], a.means[a.n + q], a.means[2 * a.n + q]);
    if (h == 0 && valid) a.density[p] = d;
  }
}

// The same per ray, with the level's sampling in front (rc_dev_sample.h sample_level_ray = k_sample_level): a wave takes a
// ray, draws its S samples, then walks its S / 32 tiles -- the sample means go from registers straight into the lookup,
// the sampler's scans and searches run in the shadow of the other waves' gathers.  One launch per proposal level.
struct RcLevelRayKArgs {
  RcLevelKArgs lv;             // lv.means unused
  RcSampleArgs sa;
  USpec us;
  float y_max;
};

constexpr int kLvSampFloats = 5 * (kSlots + 3);
static_assert(kLvSampFloats <= kLvActSteps * 64, "sampler scratch must fit the activation slice");

template <int F, int NL, int S, int ND>
__global__ __launch_bounds__(LevelCfg<F>::W * 64) void k_level_ray(RcLevelRayKArgs a) {
  using LK = LevelK<F, NL, ND>;
  constexpr int W = LK::W;
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float* wres = lds_dyn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* act_wave = lds_dyn + LK::kResFloats + wave * (kLvActSteps * 64);
  float* samp = act_wave;      // the sampler's step functions live in the wave's activation slice: it is done before the first tile writes there
  LK::load_weights(a.lv.wstream, wres, wave, lane);
  WStream ws{a.lv.wstream, wres, lane, wave};
  const int j = lane & 31, h = lane >> 5;
  for (int64_t ray = (int64_t)blockIdx.x * W + wave; ray < a.sa.n_rays; ray += (int64_t)gridDim.x * W) {
    float mx, my, mz;
    sample_level_ray(a.sa, a.us, a.y_max, ray, true, samp, lane, mx, my, mz);
#pragma unroll
    for (int t = 0; t < S / 32; ++t) {
      const float px = __shfl(mx, 32 * t + j, 64), py = __shfl(my, 32 * t + j, 64), pz = __shfl(mz, 32 * t + j, 64);
      const float d = LK::tile(a.lv.grid, a.lv.density_bias, a.lv.contract_radius, ws, act_wave, lane, px, py, pz);
      if (h == 0) a.lv.density[ray * S + 32 * t + j] = d;
    }
  }
}

// The level kernels keep a level's whole weight stream resident: up to ~145 KB of dynamic LDS, an opt-in per kernel and
// device.  `refused` remembers a device that turned it down: rc_level_supported then answers false there and the plan
// falls back to the separate gather / MLP kernels instead of failing at the launch.
std::atomic<uint64_t> g_level_refused{0};

template <class K>
bool level_prepare(K kernel, int lds, std::atomic<uint64_t>& prepared) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (rc_first_use_on_device(prepared)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      (void)hipGetLastError();
      g_level_refused.fetch_or(bit);
    }
  }
  return (g_level_refused.load() & bit) == 0;
}

template <int F, int NL, int ND>
bool prepare_level() {
  using LK = LevelK<F, NL, ND>;
  static std::atomic<uint64_t> prepared{0};
  return level_prepare(&k_level<F, NL, ND>, (LK::kResFloats + LK::W * kLvActSteps * 64) * (int)sizeof(float), prepared);
}
template <int F, int NL, int S, int ND>
bool prepare_level_ray() {
  using LK = LevelK<F, NL, ND>;
  static std::atomic<uint64_t> prepared{0};
  return level_prepare(&k_level_ray<F, NL, S, ND>, (LK::kResFloats + LK::W * (kLvActSteps * 64)) * (int)sizeof(float), prepared);
}

template <int F, int NL, int ND>
void launch_level(const RcLevelKArgs& a, hipStream_t stream) {
  using LK = LevelK<F, NL, ND>;
  constexpr int W = LK::W;
  const int lds = (LK::kResFloats + W * kLvActSteps * 64) * (int)sizeof(float);
  (void)prepare_level<F, NL, ND>();
  const int cus = rc_device_cus();
  const int64_t tiles = (a.n + 31) / 32;
  const int64_t want = (tiles + W - 1) / W;
  dim3 grid((unsigned)(want < cus ? want : cus)), block(W * 64);
  hipLaunchKernelGGL((k_level<F, NL, ND>), grid, block, lds, stream, a);
}

template <int F, int NL, int S, int ND>
void launch_level_ray(const RcLevelRayKArgs& a, hipStream_t stream, int cu_reserve = 0) {
  using LK = LevelK<F, NL, ND>;
  constexpr int W = LK::W;
  const int lds = (LK::kResFloats + W * (kLvActSteps * 64)) * (int)sizeof(float);
  (void)prepare_level_ray<F, NL, S, ND>();
  const int all = rc_device_cus();
  const int cus = cu_reserve > 0 && cu_reserve < all ? all - cu_reserve : all;
  const int64_t want = (a.sa.n_rays + W - 1) / W;
  dim3 grid((unsigned)(want < cus ? want : cus)), block(W * 64);
  hipLaunchKernelGGL((k_level_ray<F, NL, S, ND>), grid, block, lds, stream, a);
}

// 3 when the grid has the reference's level layout -- levels [0, 3) dense, each with its cell table, the others hashed --
// for which the F = 1 kernels have the compile-time form; -1 otherwise (any layout: kinds read from the level records)
constexpr int kRefDense = 3;
int level_layout(const RcGridDev& g) {
  if (g.num_features != 1) return -1;
  if (const char* e = getenv("RC_LEVEL_ANY_LAYOUT")) { if (e[0] == '1') return -1; }      // tests: force the run-time form
  for (int l = 0; l < g.num_levels; ++l) {
    const bool want_dense = l < kRefDense;
    if ((g.lvl[l].dense != 0) != want_dense || (want_dense && !g.lvl[l].cell)) return -1;
    if (!want_dense && l < kRefDense + kRcRecLevels && !g.lvl[l].rec) return -1;      // compiled for cell records there
  }
  return kRefDense;
}

}  // namespace

// true when (F, number of levels) is one of the compiled shapes
bool rc_level_supported(const RcGridDev& g) {
  // the level kernels compile the `hash & mask` form only (every table size the reference's configs use; other sizes take
  // the separate gather kernel with its modulo)
  for (int l = 0; l < g.num_levels; ++l)
    if (!g.lvl[l].dense && g.lvl[l].mask == 0) return false;
  const bool ref = level_layout(g) == kRefDense;
  if (g.num_features == 1 && g.num_levels == 6)
    return ref ? prepare_level<1, 6, kRefDense>() && prepare_level_ray<1, 6, 64, kRefDense>() : prepare_level<1, 6, -1>() && prepare_level_ray<1, 6, 64, -1>();
  if (g.num_features == 1 && g.num_levels == 7)
    return ref ? prepare_level<1, 7, kRefDense>() && prepare_level_ray<1, 7, 64, kRefDense>() : prepare_level<1, 7, -1>() && prepare_level_ray<1, 7, 64, -1>();
  if (g.num_features == 4 && g.num_levels == 8) return prepare_level<4, 8, -1>() && prepare_level_ray<4, 8, 32, -1>();
  return false;
}

// sampling + level as one launch: supported shapes of rc_level_supported with 64 (F = 1) or 32 (F = 4) samples per ray
bool rc_level_ray_supported(const RcGridDev& g, int S) {
  return rc_level_supported(g) && S == (g.num_features == 1 ? 64 : 32);
}

void rc_launch_level_ray(const RcLevelArgs& A, const RcSampleArgs& sa, hipStream_t stream) {
  if (sa.n_rays <= 0) return;
  RcLevelRayKArgs a{};
  a.lv.grid = *A.grid; a.lv.means = nullptr; a.lv.n = A.n; a.lv.wstream = A.wstream;
  a.lv.density_bias = A.density_bias; a.lv.contract_radius = A.contract_radius; a.lv.density = A.density;
  a.sa = sa;
  a.us = make_uspec(sa.S, sa.jitter != nullptr);
  a.y_max = sa.raydist_p < 0.0f ? nextafterf((sa.raydist_p - 1.0f) / sa.raydist_p, -INFINITY) : 0.0f;      // as rc_launch_sample
  const bool ref = level_layout(a.lv.grid) == kRefDense;
  if (a.lv.grid.num_features == 1 && a.lv.grid.num_levels == 6) { if (ref) launch_level_ray<1, 6, 64, kRefDense>(a, stream); else launch_level_ray<1, 6, 64, -1>(a, stream); }
  else if (a.lv.grid.num_features == 1 && a.lv.grid.num_levels == 7) { if (ref) launch_level_ray<1, 7, 64, kRefDense>(a, stream); else launch_level_ray<1, 7, 64, -1>(a, stream); }
  else if (a.lv.grid.num_features == 4 && a.lv.grid.num_levels == 8) launch_level_ray<4, 8, 32, -1>(a, stream, A.cu_reserve);
}

void rc_launch_level(const RcLevelArgs& A, hipStream_t stream) {
  if (A.n <= 0) return;
  RcLevelKArgs a{};
  a.grid = *A.grid; a.means = A.means; a.n = A.n; a.wstream = A.wstream;
  a.density_bias = A.density_bias; a.contract_radius = A.contract_radius; a.density = A.density;
  const bool ref = level_layout(a.grid) == kRefDense;
  if (a.grid.num_features == 1 && a.grid.num_levels == 6) { if (ref) launch_level<1, 6, kRefDense>(a, stream); else launch_level<1, 6, -1>(a, stream); }
  else if (a.grid.num_features == 1 && a.grid.num_levels == 7) { if (ref) launch_level<1, 7, kRefDense>(a, stream); else launch_level<1, 7, -1>(a, stream); }
  else if (a.grid.num_features == 4 && a.grid.num_levels == 8) launch_level<4, 8, -1>(a, stream);
}
