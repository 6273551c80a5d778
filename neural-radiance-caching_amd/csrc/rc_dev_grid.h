// Device-side multiresolution grid lookup for one (point, level) (shared by rc_hashgrid.hip and
// rc_fused.hip).  See rc_hashgrid.hip for the reference mapping.
#pragma once
#include "rc_internal.h"
#include <type_traits>
#include <utility>

namespace rcdev {


constexpr uint32_t kPi2 = 19349663u;   // grid_utils.py:102
constexpr uint32_t kPi3 = 83492791u;   // grid_utils.py:103

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N) -- expanded in the AST, so the fragment
// indices are constants whatever the optimizer's unroll budget says (a rolled k-loop turns the operand register sets
// into runtime-indexed ones, tools/isa_scan.py)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& body, std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& body) { static_for_impl(body, std::make_integer_sequence<int, N>{}); }

// Value of the second (half != 0) or first argument, BY VALUE: a conditional on two struct members is an lvalue, for
// which the compiler selects the address and loads per lane; two values in scalar registers become one v_cndmask.
template <class T>
__device__ __forceinline__ T pick_half(int half, T first, T second) { return half ? second : first; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int F> struct Vec;
template <> struct Vec<1> { float v[1]; };
template <> struct Vec<4> { float v[4]; };

#ifdef RC_GATHER_FAKE
// timing experiment (tools/fused_critical_path.py): every lookup goes to the same few entries -- the lookups' issue time
// without their memory time.  The mask passes through an empty asm so that the index arithmetic in front of it stays.
__device__ __forceinline__ uint32_t fake_mask() { uint32_t m = RC_GATHER_FAKE; asm volatile("" : "+s"(m)); return m; }
#endif

template <int F>
__device__ __forceinline__ Vec<F> load_entry(const float* __restrict__ table, uint32_t idx) {
  Vec<F> r;
#ifdef RC_GATHER_FAKE
  idx &= fake_mask();
#endif
  if constexpr (F == 4) {
    const float4 q = reinterpret_cast<const float4*>(table)[idx];
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = table[idx];
  }
  return r;
}

// coord.contract(x / radius)
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


// Trilinear lookup of level L at the (already contracted) position (x, y, z), in two halves so that a
// caller can put the fetches of several levels in flight before combining any of them:
//   grid_fetch   : cell + weights, the eight entry indices (one branch on the level's kind; a zero-padded
//                  corner reads entry 0 and is masked afterwards), then the 8 corner loads
//   grid_combine : acc[F] interpolated features (NOT yet scaled by the precondition factor);
//                  jacc[3*F] (JAC): d feature / d loc_a for the three location axes in the level's own
//                  axis order.  Corners are combined in the reference's order (b2 fastest).
template <int F> struct Corners { Vec<F> val[8]; float cw[3]; uint32_t zero_mask; };

// (x01, y01, z01) = unit_box(bbox, contracted position): level independent, computed once per point.
// `size`, `mask`, `entries` and `table` may differ per lane (a wave that holds two levels on its two half-waves), and so
// may `dense` (see grid_fetch).  The per-axis terms of the index are computed once per level (two
// candidates per axis), a corner then costs an xor/add, the address and the load.
// x01 = (x - bbox_min) / (bbox_max - bbox_min) (grid_utils.py:820, 863)
__device__ __forceinline__ float unit_box(float bbox, float x) { return rc_div(x - (-bbox), bbox - (-bbox)); }

// Cell of a dense level in its cell table (built by the host: for every cell origin of the zero-padded volume, (N + 3)^3
// of them, the 8 corner entries in combine order, zeros for the padding already in place) and the three interpolation
// weights: trilerp 'grid' branch, flip(coords - 0.5) then +1 for the zero padding (grid_utils.py:711, 390).
__device__ __forceinline__ uint32_t cell_index(int size, float x01, float y01, float z01, float (&cw)[3]) {
  const float N = (float)size;
  const float cx = x01 * N, cy = y01 * N, cz = z01 * N;     // x01 * grid_size (grid_utils.py:820, 863)
  const float loc[3] = {(cz - 0.5f) + 1.0f, (cy - 0.5f) + 1.0f, (cx - 0.5f) + 1.0f};
  int base[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float fl = floorf(loc[a]);
    cw[a] = loc[a] - fl;
    base[a] = (int)fl;
  }
  // cell origin clamped to [-1, N + 1]: beyond that both corners of an axis are padding anyway
  const int M = size + 3;
  const int q0 = min(max(base[0], -1), size + 1) + 1, q1 = min(max(base[1], -1), size + 1) + 1,
            q2 = min(max(base[2], -1), size + 1) + 1;
  // 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction): every operand is below 2^24
  return __umul24(__umul24((uint32_t)q2, (uint32_t)M) + (uint32_t)q1, (uint32_t)M) + (uint32_t)q0;
}

// A dense F = 1 level through its cell table when the WHOLE WAVE is on such a level: a lane reads its 8 corners as two
// 16-byte loads of one 32-byte block instead of four scattered line pairs, and the clamp / zero-mask arithmetic per
// corner disappears.  (Vector-typed loads: HIP's float4 is a struct whose load is split into four scalar loads, which the
// optimizer then merges with a hashed side's eight corner loads behind a branch -- eight 4-byte loads of one block,
// four times the sector look-ups.)
__device__ __forceinline__ void grid_fetch_cell(const float* __restrict__ cell_table, int size, float x01, float y01, float z01,
                                                Corners<1>& C) {
  C.zero_mask = 0;
  const uint32_t cell = cell_index(size, x01, y01, z01, C.cw);
  const f32x4* cp = reinterpret_cast<const f32x4*>(cell_table) + (size_t)cell * 2;
  const f32x4 lo = cp[0], hi = cp[1];
  C.val[0].v[0] = lo.x; C.val[1].v[0] = lo.y; C.val[2].v[0] = lo.z; C.val[3].v[0] = lo.w;
  C.val[4].v[0] = hi.x; C.val[5].v[0] = hi.y; C.val[6].v[0] = hi.z; C.val[7].v[0] = hi.w;
}

// POW2: the caller guarantees power-of-two hash tables (mask != 0): no modulo path.
// STRIDE: distance between consecutive entries in units of F floats (2 for the interleaved [density | appearance]
// tables of the fused kernel, where `table` already points at this lane's half of an entry pair).
// CELL: a dense level's `table` is its cell table (cell_index; with STRIDE = 2 a 256-byte block of [density | appearance]
// pairs per cell).
// `dense` may differ between the two half-waves of a wave (the level kernels split a point's levels by parity): the two
// sides then run one after the other under exec masks and only compute the eight entry INDICES; the eight loads are
// common code behind them -- loads inside the sides target the same registers and would wait for each other.
// UNIFORM: the caller guarantees a wave-uniform `dense` (the stand-alone lookup kernels: the level is blockIdx.y): each side
// then computes a corner's index and issues its load at once, pair by pair (the form of rounds 1-2), instead of eight
// indices first and the eight loads as one burst behind the merge.  Same registers, same instruction counts -- and 22.2
// against 26.2 us for k_hashgrid_fwd<4, JAC> on the bench's 32 768 points (19.3 / 23.0 without the Jacobian): eight 16-byte
// wave-loads back to back from every wave queue up in front of the texture-address unit; spaced by their index arithmetic
// they interleave with the other waves' (profiles/r04_bisect_hashgrid_standalone.txt: `abold` = this header's
// predecessor under HEAD's library).
// `rec` (may differ per lane, like `dense`): `table` is the level's cell-record table (hrec_index below; a hashed level whose
// 8 corner entries sit side by side per cell origin) -- the eight indices are then record * 8 + corner.
__device__ __forceinline__ uint32_t hrec_index(int size, float x01, float y01, float z01, float (&cw)[3]);
template <int F, bool POW2 = false, int STRIDE = 1, bool CELL = false, bool UNIFORM = false>
__device__ __forceinline__ void grid_fetch(const float* __restrict__ table, int size, uint32_t mask, uint32_t entries,
                                           bool dense, float x01, float y01, float z01, Corners<F>& C, bool rec = false) {
  C.zero_mask = 0;
  uint32_t idx[8];
  // fetch order: the two corners that differ in b0 -- x and x + 1: adjacent entries of a dense level, an index that
  // differs in the low bits only on a hashed one (same cache line 7 times out of 8) -- back to back
  auto visit = [&](auto&& index_of) {
    if constexpr (UNIFORM) {
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int c = ((o & 1) << 2) | (o >> 1);
        uint32_t i = index_of(c) * STRIDE;
        // An empty volatile asm on the index of every x-pair pins the pair behind its own index arithmetic: the optimizer
        // otherwise sinks the (identical) loads of the two sides into the block behind them and the scheduler clusters
        // them -- the burst again (inline asm is neither sunk nor crossed by memory operations; it emits nothing).
        if ((o & 1) == 0) asm volatile("" : "+v"(i));
        C.val[c] = load_entry<F>(table, i);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) idx[c] = index_of(c);
    }
  };
  if (rec) {
    const uint32_t r8 = hrec_index(size, x01, y01, z01, C.cw) * 8u;
    visit([&](int c) { return r8 + (uint32_t)c; });
  } else if (dense) {
    if constexpr (CELL) {
      const uint32_t cell = cell_index(size, x01, y01, z01, C.cw);
      visit([&](int c) { return cell * 8u + (uint32_t)c; });
    } else {
      const float N = (float)size;
      const float cx = x01 * N, cy = y01 * N, cz = z01 * N;     // x01 * grid_size (grid_utils.py:820, 863)
      // trilerp 'grid' branch: flip(coords - 0.5) then +1 for the zero padding (grid_utils.py:711, 390)
      const float loc[3] = {(cz - 0.5f) + 1.0f, (cy - 0.5f) + 1.0f, (cx - 0.5f) + 1.0f};
      int base[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float fl = floorf(loc[a]);
        C.cw[a] = loc[a] - fl;
        base[a] = (int)fl;
      }
      // clamp to the padded volume [0, N+1]; the pad (0 and N+1) holds zeros (grid_utils.py:384-390, 435-438);
      // data[loc2, loc1, loc0] = grid[x, y, z]: idx = ((k2-1) N + (k1-1)) N + (k0-1)
      uint32_t term[3][2];
      bool out[3][2];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int k = min(max(base[a] + b, 0), size + 1);
          out[a][b] = (k < 1) | (k > size);
          const uint32_t km = (uint32_t)(k - 1);
          term[a][b] = a == 0 ? km : (a == 1 ? km * (uint32_t)size : km * (uint32_t)size * (uint32_t)size);
        }
      visit([&](int c) {
        const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
        const bool zero = out[0][b0] | out[1][b1] | out[2][b2];
        C.zero_mask |= (zero ? 1u : 0u) << c;
        return zero ? 0u : term[2][b2] + term[1][b1] + term[0][b0];
      });
    }
  } else {
    const float N = (float)size;
    const float cx = x01 * N, cy = y01 * N, cz = z01 * N;       // x01 * grid_size (grid_utils.py:820, 863)
    const float loc[3] = {cx - 0.5f, cy - 0.5f, cz - 0.5f};     // grid_utils.py:61
    int base[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float fl = floorf(loc[a]);
      C.cw[a] = loc[a] - fl;
      base[a] = (int)fl;
    }
    // int32 -> uint32 wraparound hash (grid_utils.py:99-111): x ^ y * pi2 ^ z * pi3
    const uint32_t hx[2] = {(uint32_t)base[0], (uint32_t)base[0] + 1u};
    const uint32_t y0 = (uint32_t)base[1] * kPi2, z0 = (uint32_t)base[2] * kPi3;
    const uint32_t hy[2] = {y0, y0 + kPi2}, hz[2] = {z0, z0 + kPi3};
    visit([&](int c) {
      const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
      const uint32_t hsh = hx[b0] ^ hy[b1] ^ hz[b2];
      return (POW2 || mask) ? (hsh & mask) : (hsh % entries);
    });
  }
  if constexpr (!UNIFORM) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int c = ((o & 1) << 2) | (o >> 1);
      C.val[c] = load_entry<F>(table, idx[c] * STRIDE);
    }
  }
}

// ---- A pair of F = 1 levels of one point on the two half-waves of a wave ------------------------------------------
// The level kernels and the two-wave fused kernel put a point on lanes j and j + 32 and give the lower half-wave the
// even grid levels to interpolate, the upper one the odd levels (a lane's feature lands in its own MFMA B column).
// Splitting the LOADS the same way -- half-wave h reads the 8 corners of level 2 i + h -- makes the 64 lanes of every
// load instruction 64 different sectors, and a pair of levels of different kinds (dense | hashed) two code paths under
// exec masks.  Here the loads of a pair (A on the lower half, B on the upper half) are split by CORNER instead: every
// lane reads the four corners with b0 = h (b0: the x bit, corner c = 4 b0 + 2 b1 + b2) of BOTH levels, so lanes j and
// j + 32 of an instruction read x and x + 1 -- the two halves of a cell's 32-byte block in a cell table, adjacent
// entries of a hashed table (one 64-byte sector 15 times out of 16): about half the sector look-ups, which is what bounds
// a random gather (profiles/r03_two_wave_probe.txt), and no divergence, the kinds being compile-time.  When the values
// have arrived, four v_permlane32_swap hand each half-wave the other four corners of ITS level; grid_combine then adds
// them in the reference's corner order as before (bitwise the same feature).
enum : int { kLevelNone = -1, kLevelHashed = 0, kLevelCell = 2, kLevelHRec = 3 };    // kinds of a level (hashed: power-of-two table)

// kind of level l of an F = 1 grid in the reference's layout (nd leading dense levels, each with its cell table; the
// next kRcRecLevels hashed levels through their cell records)
constexpr int ref_level_kind(int l, int nd) { return l < nd ? kLevelCell : (l < nd + kRcRecLevels ? kLevelHRec : kLevelHashed); }

// Record of a hashed level in its cell-record table and the three interpolation weights: the hashed branch's location
// x01 * N - 0.5 in (x, y, z) order (grid_utils.py:61), cell origin floor(.) in [-1, N - 1] for every point inside the
// bounding box (x01 in [0, 1]); outside it the origin is clamped -- the record is then not the point's cell, and every
// consumer of an F = 1 grid zeroes the density of such a point (convert_raw_density, geometry.py:333-337) whatever its
// features are.  Weights and corner order are the hashed lookup's own: bitwise the same feature inside the box.
__device__ __forceinline__ uint32_t hrec_index(int size, float x01, float y01, float z01, float (&cw)[3]) {
  const float N = (float)size;
  const float loc[3] = {x01 * N - 0.5f, y01 * N - 0.5f, z01 * N - 0.5f};
  int q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float fl = floorf(loc[a]);
    cw[a] = loc[a] - fl;
    q[a] = min(max((int)fl, -1), size - 1) + 1;
  }
  const uint32_t M = (uint32_t)size + 1u;
  return __umul24(__umul24((uint32_t)q[0], M) + (uint32_t)q[1], M) + (uint32_t)q[2];
}

struct PairCorners { float va[4], vb[4]; float cw[3]; };

// this lane's four corners (b0 = h) of one level: loads issued into v[], interpolation weights of the level into cw[]
template <int KIND>
__device__ __forceinline__ void half_corners(const float* __restrict__ table, int size, uint32_t mask, int h, float x01,
                                             float y01, float z01, float (&cw)[3], float (&v)[4]) {
  if constexpr (KIND == kLevelCell || KIND == kLevelHRec) {
    uint32_t cell = KIND == kLevelCell ? cell_index(size, x01, y01, z01, cw) : hrec_index(size, x01, y01, z01, cw);
#ifdef RC_GATHER_FAKE
    cell &= fake_mask();
#endif
    const f32x4 q = *reinterpret_cast<const f32x4*>(table + (size_t)cell * 8 + 4 * h);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    const float N = (float)size;
    const float cx = x01 * N, cy = y01 * N, cz = z01 * N;       // x01 * grid_size (grid_utils.py:820, 863)
    const float loc[3] = {cx - 0.5f, cy - 0.5f, cz - 0.5f};     // grid_utils.py:61
    int base[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float fl = floorf(loc[a]);
      cw[a] = loc[a] - fl;
      base[a] = (int)fl;
    }
    // int32 -> uint32 wraparound hash (grid_utils.py:99-111): x ^ y * pi2 ^ z * pi3
    const uint32_t hx = (uint32_t)base[0] + (uint32_t)h;
    const uint32_t y0 = (uint32_t)base[1] * kPi2, z0 = (uint32_t)base[2] * kPi3;
    const uint32_t hy[2] = {y0, y0 + kPi2}, hz[2] = {z0, z0 + kPi3};
#pragma unroll
#ifdef RC_GATHER_FAKE
    for (int c = 0; c < 4; ++c) v[c] = table[(hx ^ hy[(c >> 1) & 1] ^ hz[c & 1]) & mask & fake_mask()];
#else
    for (int c = 0; c < 4; ++c) v[c] = table[(hx ^ hy[(c >> 1) & 1] ^ hz[c & 1]) & mask];
#endif
  }
}

// issue the loads of the pair (KB == kLevelNone: a single level A, its corners still split between the half-waves)
template <int KA, int KB>
__device__ __forceinline__ void pair_fetch(const float* __restrict__ tab_a, int size_a, uint32_t mask_a,
                                           const float* __restrict__ tab_b, int size_b, uint32_t mask_b, int h, float x01,
                                           float y01, float z01, PairCorners& P) {
  float cwa[3], cwb[3];
  half_corners<KA>(tab_a, size_a, mask_a, h, x01, y01, z01, cwa, P.va);
  if constexpr (KB != kLevelNone) {
    half_corners<KB>(tab_b, size_b, mask_b, h, x01, y01, z01, cwb, P.vb);
#pragma unroll
    for (int a = 0; a < 3; ++a) P.cw[a] = h ? cwb[a] : cwa[a];
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) P.cw[a] = cwa[a];
  }
}

// v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second; returns {first, second} afterwards
__device__ __forceinline__ void swap_halves(float& first, float& second) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(first), __float_as_uint(second), false, false);
  first = __uint_as_float(r[0]); second = __uint_as_float(r[1]);
}

// after the loads: the eight corners of this half-wave's level (A on the lower, B on the upper half) in combine order.
// Lower half: own A corners 0-3 | A corners 4-7 from lane + 32; upper half: B corners 0-3 from lane - 32 | own B corners 4-7.
template <bool HAS_B>
__device__ __forceinline__ void pair_finish(const PairCorners& P, Corners<1>& C) {
  C.zero_mask = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) C.cw[a] = P.cw[a];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float first = P.va[k], second = HAS_B ? P.vb[k] : P.va[k];
    swap_halves(first, second);
    C.val[k].v[0] = first; C.val[4 + k].v[0] = second;
  }
}

// PACKED (F = 4): the four features as two register pairs on v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 -- the same separate
// roundings per element, bitwise the same result.  The fused and level kernels use it (-6 % static vector instructions
// next to their MFMA work); the stand-alone lookup kernels do not: the register pairs it needs take k_hashgrid_fwd<4, JAC>
// from 62 to 76 vector registers -- six instead of eight waves per SIMD on a kernel that lives on its loads in flight
// (profiles/r04_bisect_hashgrid_standalone.txt: +2 us per launch at commit 241cc23 against 1a919e3).
template <int F, bool JAC, bool PACKED = true>
__device__ __forceinline__ void grid_combine(const Corners<F>& C, float (&acc)[F], float (&jacc)[JAC ? 3 * F : 1]) {
  float fw[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) fw[a] = 1.0f - C.cw[a];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0f;
  float v[8][F];
  if constexpr (F == 4 && PACKED) {
    // the four features as two register pairs: packed multiplies and adds (v_pk_mul_f32 / v_pk_add_f32, two lanes of
    // fp32 per instruction, the same separate roundings per element)
    f32x2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
      const float w0 = b0 ? C.cw[0] : fw[0], w1 = b1 ? C.cw[1] : fw[1], w2 = b2 ? C.cw[2] : fw[2];
      const float w = (w0 * w1) * w2;
      const bool zero = (C.zero_mask >> c) & 1u;
#pragma unroll
      for (int f = 0; f < 4; ++f) v[c][f] = zero ? 0.0f : C.val[c].v[f];
      const f32x2 vlo = {v[c][0], v[c][1]}, vhi = {v[c][2], v[c][3]}, ww = {w, w};
      lo = lo + vlo * ww;
      hi = hi + vhi * ww;
    }
    acc[0] = lo.x; acc[1] = lo.y; acc[2] = hi.x; acc[3] = hi.y;
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
      const float w0 = b0 ? C.cw[0] : fw[0], w1 = b1 ? C.cw[1] : fw[1], w2 = b2 ? C.cw[2] : fw[2];
      const float w = (w0 * w1) * w2;
      const bool zero = (C.zero_mask >> c) & 1u;
#pragma unroll
      for (int f = 0; f < F; ++f) v[c][f] = zero ? 0.0f : C.val[c].v[f];
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = acc[f] + v[c][f] * w;
    }
  }
  if constexpr (JAC) {
    // d feature / d loc_a = sum over the 4 corner pairs along axis a of (product of the two other axes' weights) x
    // (difference of the pair): 12 differences + 12 fused multiply-adds per feature.  (The value above keeps the
    // reference's corner order and separate roundings; the derivative only feeds the analytic normals, whose own
    // reference is an autodiff gradient with no prescribed summation order.)
    const float w0s[2] = {fw[0], C.cw[0]}, w1s[2] = {fw[1], C.cw[1]}, w2s[2] = {fw[2], C.cw[2]};
    float p12[2][2], p02[2][2], p01[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) { p12[i][k] = w1s[i] * w2s[k]; p02[i][k] = w0s[i] * w2s[k]; p01[i][k] = w0s[i] * w1s[k]; }
    if constexpr (F == 4 && PACKED) {
      // packed over feature pairs as above (v_pk_add_f32 with a negated operand, v_pk_fma_f32)
#pragma unroll
      for (int fp = 0; fp < 2; ++fp) {
        f32x2 vv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) vv[c] = f32x2{v[c][2 * fp], v[c][2 * fp + 1]};
        f32x2 g0 = {0.0f, 0.0f}, g1 = {0.0f, 0.0f}, g2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            g0 = __builtin_elementwise_fma(vv[4 + 2 * i + k] - vv[2 * i + k], f32x2{p12[i][k], p12[i][k]}, g0);
            g1 = __builtin_elementwise_fma(vv[4 * i + 2 + k] - vv[4 * i + k], f32x2{p02[i][k], p02[i][k]}, g1);
            g2 = __builtin_elementwise_fma(vv[4 * i + 2 * k + 1] - vv[4 * i + 2 * k], f32x2{p01[i][k], p01[i][k]}, g2);
          }
        jacc[0 * F + 2 * fp] = g0.x; jacc[0 * F + 2 * fp + 1] = g0.y;
        jacc[1 * F + 2 * fp] = g1.x; jacc[1 * F + 2 * fp + 1] = g1.y;
        jacc[2 * F + 2 * fp] = g2.x; jacc[2 * F + 2 * fp + 1] = g2.y;
      }
    } else
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // corner index c = 4 b0 + 2 b1 + b2
          g0 = __builtin_fmaf(v[4 + 2 * i + k][f] - v[2 * i + k][f], p12[i][k], g0);          // (b1, b2) = (i, k)
          g1 = __builtin_fmaf(v[4 * i + 2 + k][f] - v[4 * i + k][f], p02[i][k], g1);          // (b0, b2) = (i, k)
          g2 = __builtin_fmaf(v[4 * i + 2 * k + 1][f] - v[4 * i + 2 * k][f], p01[i][k], g2);  // (b0, b1) = (i, k)
        }
      jacc[0 * F + f] = g0; jacc[1 * F + f] = g1; jacc[2 * F + f] = g2;
    }
  }
}

template <int F, bool JAC>
__device__ __forceinline__ void grid_level(const RcGridLevel& L, float bbox, float x, float y, float z, float (&acc)[F],
                                           float (&jacc)[JAC ? 3 * F : 1]) {
  Corners<F> C;
  if constexpr (F == 1 && !JAC) {
    // one contiguous 32-byte read per cell instead of eight scattered corners (wave-uniform branch: blockIdx.y = level)
    if (L.cell) {
      grid_fetch_cell(L.cell, L.size, unit_box(bbox, x), unit_box(bbox, y), unit_box(bbox, z), C);
      grid_combine<F, JAC>(C, acc, jacc);
      return;
    }
  }
  // power-of-two table or not decided ONCE per level (wave-uniform): as a test per corner the modulo path put eight
  // branches and ~160 instructions between the index arithmetic and the first load (r04_bisect_hashgrid_standalone.txt)
  const float x01 = unit_box(bbox, x), y01 = unit_box(bbox, y), z01 = unit_box(bbox, z);
  if (L.rec != nullptr) {
    // a hashed level with cell records (wave-uniform): the lanes whose point lies inside the bounding box read their 8
    // corners as ONE record (32 bytes at F = 1, one 128-byte line at F = 4) instead of ~4 sectors of the hash table; a
    // point outside it has no record of its own (hrec_index clamps) and takes the hash table -- this kernel's features are
    // an output (rc_hashgrid_lookup), so they must be right out there too.  Same entries, weights and corner order.
    const bool in01 = (x01 >= 0.0f) & (x01 <= 1.0f) & (y01 >= 0.0f) & (y01 <= 1.0f) & (z01 >= 0.0f) & (z01 <= 1.0f);
    grid_fetch<F, true, 1, false, true>(in01 ? L.rec : L.table, L.size, L.mask, L.entries, false, x01, y01, z01, C, in01);
  } else if (L.dense != 0 || L.mask != 0u) grid_fetch<F, true, 1, false, true>(L.table, L.size, L.mask, L.entries, L.dense != 0, x01, y01, z01, C);
  else grid_fetch<F, false, 1, false, true>(L.table, L.size, L.mask, L.entries, false, x01, y01, z01, C);
  grid_combine<F, JAC, false>(C, acc, jacc);
}

}  // namespace rcdev
