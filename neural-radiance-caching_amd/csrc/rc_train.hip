// Backward pass of one proposal level's density field: hash-grid tables + density MLP.
//
// Replaces, for the parameters of `Cache/Sampler/MLP_l`, what jax.value_and_grad produces in the reference's train
// step (internal/train_utils.py:3128-3131) for the sub-graph
//   HashEncoding.__call__ (internal/grid_utils.py:808-905)  ->  DensityMLP.run_network (internal/geometry.py:155-168)
//   ->  convert_raw_density (internal/geometry.py:318-341)
// given the upstream gradients d L / d density [n] and (optionally) d L / d feature [n, 64].
//
// Four kernels, all on the forward pass's data layouts:
//   k_density_bwd   forward + per-point backward on the matrix cores (same transposed fp32 MFMA formulation and
//                   weight stream as k_density_mlp; the stream carries W1^T and W0^T behind the forward layers).
//                   One wave = 32 points.  Emits, point-major, what the weight gradients contract over:
//                   a1 = relu(layer 0), a2 = relu(layer 1), d2 / d1 = gradients at the two pre-activations, the
//                   staged features, g_raw; and d L / d grid feature feature-major for the scatter.
//   k_wgrad         dW1 = a1^T d2, dW0 = feat^T d1 as MFMAs whose K axis is the POINT axis (both operands are rows of
//                   point-major arrays: lane l reads point 2 s + (l >> 5), column l & 31 -- coalesced, no transpose),
//                   db = column sums, dWout = a2^T g_raw on the VALU.  Persistent waves, accumulators in registers,
//                   the four waves of a workgroup added in wave order through LDS, one partial per workgroup.
//   k_grad_reduce   sums the per-wave partials in a fixed order (deterministic weight gradients) into the caller's buffer.
//   k_grid_scatter  transposed trilinear lookup: the lanes of a (point, level) recompute the 8 corner indices / weights
//                   exactly like the forward gather and add w * dfeat with hardware float atomics
//                   (global_atomic_add_f32, executed at the memory side; table gradients are order-dependent in the
//                   last bits).  Lanes are laid out so that adds sharing a 64-byte row share a wave-instruction;
//   k_grid_scatter_small  16^3 levels, where a whole batch hits a few hundred rows, are summed in LDS first.
#include <hip/hip_runtime.h>

#include "rc_dev_grid.h"
#include "rc_dev_mlp.h"

using namespace rcdev;

namespace {

constexpr int kPartW0 = 0, kPartB0 = 2048, kPartW1 = 2112, kPartB1 = 6208, kPartWO = 6272, kPartBO = 6336;
constexpr int kPartStride = 6400;

template <int KS0>   // k-steps of layer 0 including the bias step (4, 5 or 17)
__global__ __launch_bounds__(kWaves * 64) void k_density_bwd(RcDensityBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[kRingFloats];
  __shared__ float lds[kWaves][33 * 64];
  constexpr int F_D0 = 0, F_D1 = rc_lfr32(KS0, 2), F_DO = F_D1 + rc_lfr32(33, 2), F_B1 = F_DO + rc_dfr32(1, 2), F_B0 = F_B1 + rc_lfr32(32, 2), NF = F_B0 + rc_lfr32(32, 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t p0 = tile * 32;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = p0 + j;
  const bool valid = p < a.n;        // waves past the end stay alive for the workgroup barriers
  float* act = &lds[wave][lane];
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF>(ws);

  // accumulator register (t, r) of this lane holds feature 32 t + 8 (r >> 2) + 4 h + (r & 3): four consecutive
  // columns per (t, r >> 2) -> one 16-byte store into a point-major row
  auto store_rows = [&](float* dst, const float (&v)[32]) {
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 o = make_float4(v[t * 16 + 4 * q], v[t * 16 + 4 * q + 1], v[t * 16 + 4 * q + 2], v[t * 16 + 4 * q + 3]);
        *reinterpret_cast<float4*>(dst + p * 64 + 32 * t + 8 * q + 4 * h) = o;
      }
  };

  // stage the grid features (natural k pairs) + bias step; keep a point-major copy for dW0
  // (all loads first, from clamped addresses; see k_density_mlp)
  {
    float fv[KS0 - 1];
    const int64_t pp = valid ? p : 0;
#pragma unroll
    for (int s = 0; s < KS0 - 1; ++s) {
      const int k = 2 * s + h;
      fv[s] = a.feat[(int64_t)(k < a.K ? k : a.K - 1) * a.ld + pp];
    }
#pragma unroll
    for (int s = 0; s < KS0 - 1; ++s) {
      const int k = 2 * s + h;
      const float v = (valid && k < a.K) ? fv[s] : 0.0f;
      act[s * 64] = v;
      if (valid) a.fe[p * 32 + k] = v;
    }
  }
  act[(KS0 - 1) * 64] = h == 0 ? 1.0f : 0.0f;

  f32x16 acc[2];
  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, KS0, F_D0, NF>(ws, act, acc);
  uint32_t m0 = 0, m1 = 0;           // ReLU masks, bit t*16+r
  float row[32];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      m0 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
      row[t * 16 + r] = relu0(acc[t][r]);
    }
  store_rows(a.a1, row);
  park<2, true>(acc, act, 0);
  act[32 * 64] = h == 0 ? 1.0f : 0.0f;

  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, 33, F_D1, NF>(ws, act, acc);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      m1 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
      row[t * 16 + r] = relu0(acc[t][r]);
    }
  store_rows(a.a2, row);

  float out[1], wout[32];
  dot_out1<2, 1, F_DO, NF, true>(ws, acc, out, wout);     // output_density_layer on relu(acc); both half-waves hold it

  // convert_raw_density (geometry.py:318-341) and its derivative: density = safe_exp(raw + bias) inside the box
  float g = 0.0f;
  {
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    if (valid) {
      cx = a.points[3 * p]; cy = a.points[3 * p + 1]; cz = a.points[3 * p + 2];
      contract3(cx, cy, cz, a.contract_radius);
    }
    const bool inside = (cx > -a.bbox) & (cx < a.bbox) & (cy > -a.bbox) & (cy < a.bbox) & (cz > -a.bbox) & (cz < a.bbox);
    const float x = out[0] + a.density_bias;
    const float d = inside ? rc_safe_exp(x) : 0.0f;
    // math.safe_exp is a custom_jvp (internal/math.py:153-171, 186-192): y_dot = y x_dot with the CLIPPED y, i.e. the
    // clip does not gate the gradient; jnp.where(valid, density, 0) does
    if (valid) g = a.d_density[p] * d;
    if (h == 0 && valid) {
      a.density[p] = d;
      a.graw[p] = g;
    }
  }

  // d2 = d L / d (layer-1 pre-activation) = relu'(.) (g w_out + d L / d feature)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 up = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (a.d_feature && valid) up = *reinterpret_cast<const float4*>(a.d_feature + p * 64 + 32 * t + 8 * q + 4 * h);
      const float u[4] = {up.x, up.y, up.z, up.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = t * 16 + 4 * q + i;
        row[s] = ((m1 >> s) & 1u) ? g * wout[s] + u[i] : 0.0f;
      }
    }
  store_rows(a.d2, row);
#pragma unroll
  for (int s = 0; s < 32; ++s) act[s * 64] = row[s];
  f32x16 gb[2];
  gb[0] = zero16(); gb[1] = zero16();
  mlp_layer_d<2, 32, F_B1, NF>(ws, act, gb);            // W1 . d2   (transposed layer, no bias)
#pragma unroll
  for (int s = 0; s < 32; ++s) row[s] = ((m0 >> s) & 1u) ? gb[s >> 4][s & 15] : 0.0f;
  store_rows(a.d1, row);
#pragma unroll
  for (int s = 0; s < 32; ++s) act[s * 64] = row[s];
  f32x16 gf[1];
  gf[0] = zero16();
  mlp_layer_d<1, 32, F_B0, NF>(ws, act, gf);            // W0 . d1 -> d L / d grid feature (accumulator layout)
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (i < a.K) a.dfeat[(int64_t)i * a.ld + p] = gf[0][r];
    }
  }
}

// Weight gradients: MFMAs whose K axis runs over points.  D = A B with A[m][k] = a[point k][m] (lane l supplies
// m = l & 31, k = l >> 5) and B[k][n] = d[point k][n]: D[m][n] = sum_points a[.][m] d[.][n] = dW[in m][out n].
__global__ __launch_bounds__(256) void k_wgrad(RcWgradArgs a) {
  const int wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int c = lane & 31, kh = lane >> 5;
  const int64_t nsteps = (a.n + 1) >> 1;
  const int64_t s0 = (int64_t)wave * a.steps_per_wave;
  const int64_t s1 = s0 + a.steps_per_wave < nsteps ? s0 + a.steps_per_wave : nsteps;
  f32x16 w1[2][2], w0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { w0[i] = zero16(); w1[i][0] = zero16(); w1[i][1] = zero16(); }
  float b1[2] = {0.0f, 0.0f}, b0[2] = {0.0f, 0.0f};
  // groups of 4 k-steps: all 28 operand loads of a group are in flight before its 24 MFMAs
  for (int64_t sg = s0; sg < s1; sg += 4) {
    float x1[4][2], e2[4][2], e1[4][2], xf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = 2 * (sg + u) + kh;
      const bool ok = (sg + u < s1) & (p < a.n);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        x1[u][t] = ok ? a.a1[p * 64 + 32 * t + c] : 0.0f;
        e2[u][t] = ok ? a.d2[p * 64 + 32 * t + c] : 0.0f;
        e1[u][t] = ok ? a.d1[p * 64 + 32 * t + c] : 0.0f;
      }
      xf[u] = (ok && c < a.K) ? a.fe[p * 32 + c] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int to = 0; to < 2; ++to)
          w1[ti][to] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u][ti], e2[u][to], w1[ti][to], 0, 0, 0);
#pragma unroll
      for (int to = 0; to < 2; ++to) {
        w0[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[u], e1[u][to], w0[to], 0, 0, 0);
        b1[to] += e2[u][to];
        b0[to] += e1[u][to];
      }
    }
  }
  // output layer: lane = hidden feature, sequential over this wave's points (fixed order)
  float wo = 0.0f, bo = 0.0f;
  {
    const int64_t q1 = 2 * s1 < a.n ? 2 * s1 : a.n;
    for (int64_t p = 2 * s0; p < q1; ++p) {
      const float g = a.graw[p];
      wo = __builtin_fmaf(g, a.a2[p * 64 + lane], wo);
      bo += g;
    }
  }
  // the four waves of the workgroup are added in wave order through LDS: one partial per workgroup
  __shared__ float red[kPartStride];
  const int wv = threadIdx.x >> 6;
  for (int turn = 0; turn < 4; ++turn) {
    if (wv == turn) {
      auto put = [&](int i, float v) { red[i] = turn == 0 ? v : red[i] + v; };
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
        for (int to = 0; to < 2; ++to) {
          put(kPartW0 + m * 64 + 32 * to + c, w0[to][r]);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) put(kPartW1 + (32 * ti + m) * 64 + 32 * to + c, w1[ti][to][r]);
        }
      }
#pragma unroll
      for (int to = 0; to < 2; ++to) {
        const float sb1 = b1[to] + __shfl_xor(b1[to], 32, 64), sb0 = b0[to] + __shfl_xor(b0[to], 32, 64);
        if (kh == 0) { put(kPartB1 + 32 * to + c, sb1); put(kPartB0 + 32 * to + c, sb0); }
      }
      put(kPartWO + lane, wo);
      if (lane == 0) put(kPartBO, bo);
    }
    __syncthreads();
  }
  float* P = a.partial + (int64_t)blockIdx.x * kPartStride;
  for (int i = threadIdx.x; i <= kPartBO; i += 256) P[i] = red[i];
}

// grads[.] += sum over the workgroup partials in a fixed order.  Output layout: [W0 K x 64 | b0 64 | W1 64 x 64 | b1 64 | Wout 64 | bout 1].
// A workgroup owns 16 consecutive values; thread (v = tid & 15, s = tid >> 4) adds the partials of waves
// s, s + 16, ... of the workgroup partials (64-byte row pieces, many loads in flight), the 16 slices are then added in slice order.
__global__ __launch_bounds__(256) void k_grad_reduce(const float* __restrict__ partial, int nparts, int K, float* __restrict__ grads) {
  __shared__ float part[16][17];
  const int v = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + v;
  float s = 0.0f;
  if (i <= kPartBO)
    for (int w = sl; w < nparts; w += 16) s += partial[(int64_t)w * kPartStride + i];
  part[sl][v] = s;
  __syncthreads();
  if (sl != 0 || i > kPartBO) return;
  int o;
  if (i < kPartB0) {
    if (i >= K * 64) return;       // rows of the padded 32-row input tile beyond the level's K features
    o = i;
  } else {
    o = i - kPartB0 + K * 64;
  }
  float t = 0.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) t += part[q][v];
  grads[o] += t;
}

// Transposed trilinear lookup (the index / weight arithmetic is grid_fetch's + grid_combine's, rc_dev_grid.h).
// Float atomics execute at the memory side as 64-byte requests, and scattered 4-byte adds are bound by the REQUEST
// rate (~20 G/s measured), not by bytes.  The thread mapping therefore puts adds that share a 64-byte row on
// neighbouring lanes of ONE wave-instruction:
//   F = 4: four lanes per (point, level), one feature each -> every corner instruction carries 16 whole 16-byte entries;
//   F = 1: two lanes per (point, level), one for each corner along the fastest table axis (adjacent entries for dense
//          levels, hash indices that differ in the low bit) -> 32 entry pairs per instruction.
// The index arithmetic is recomputed by the lanes of a group (a few dozen integer ops against a memory-side request).
template <int F>
__global__ __launch_bounds__(256) void k_grid_scatter(RcGridScatterArgs a) {
  constexpr int G = F == 4 ? 4 : 2;                 // lanes per (point, level)
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t p = gid / G;
  const int sub = (int)(gid % G);
  const int l = blockIdx.y + a.level0;
  if (p >= a.n || ((a.lds_levels >> l) & 1u)) return;     // small dense levels: k_grid_scatter_small
  float x = a.points[3 * p], y = a.points[3 * p + 1], z = a.points[3 * p + 2];
  if (a.contract_radius > 0.0f) contract3(x, y, z, a.contract_radius);
  const RcGridLevel L = a.grid.lvl[l];
  float* __restrict__ gt = a.gtable[l];
  const int f = F == 4 ? sub : 0;
  const int LF = a.grid.num_levels * F;
  const float df = (a.point_major ? a.dfeat[p * LF + l * F + f] : a.dfeat[(int64_t)(l * F + f) * a.ld + p]) * a.grid.precondition;
  const float N = (float)L.size;
  const float cx = unit_box(a.grid.bbox, x) * N, cy = unit_box(a.grid.bbox, y) * N, cz = unit_box(a.grid.bbox, z) * N;
  float cw[3];
  int base[3];
  if (L.dense) {
    const float loc[3] = {(cz - 0.5f) + 1.0f, (cy - 0.5f) + 1.0f, (cx - 0.5f) + 1.0f};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { const float fl = floorf(loc[ax]); cw[ax] = loc[ax] - fl; base[ax] = (int)fl; }
  } else {
    const float loc[3] = {cx - 0.5f, cy - 0.5f, cz - 0.5f};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { const float fl = floorf(loc[ax]); cw[ax] = loc[ax] - fl; base[ax] = (int)fl; }
  }
  const float fw[3] = {1.0f - cw[0], 1.0f - cw[1], 1.0f - cw[2]};
#pragma unroll
  for (int cnr = 0; cnr < 8; ++cnr) {
    const int b0 = (cnr >> 2) & 1, b1 = (cnr >> 1) & 1, b2 = cnr & 1;
    if (F == 1 && b0 != sub) continue;             // this lane's corners along the fastest axis
    const float w = ((b0 ? cw[0] : fw[0]) * (b1 ? cw[1] : fw[1])) * (b2 ? cw[2] : fw[2]);
    uint32_t idx;
    bool zero = false;
    if (L.dense) {
      const int k0 = min(max(base[0] + b0, 0), L.size + 1), k1 = min(max(base[1] + b1, 0), L.size + 1),
                k2 = min(max(base[2] + b2, 0), L.size + 1);
      zero = (k0 < 1) | (k0 > L.size) | (k1 < 1) | (k1 > L.size) | (k2 < 1) | (k2 > L.size);   // zero padding: no parameter
      idx = ((uint32_t)(k2 - 1) * (uint32_t)L.size + (uint32_t)(k1 - 1)) * (uint32_t)L.size + (uint32_t)(k0 - 1);
    } else {
      const uint32_t hsh = ((uint32_t)base[0] + (uint32_t)b0) ^ (((uint32_t)base[1] + (uint32_t)b1) * kPi2) ^
                           (((uint32_t)base[2] + (uint32_t)b2) * kPi3);
      idx = L.mask ? (hsh & L.mask) : (hsh % L.entries);
    }
    if (zero) continue;
    unsafeAtomicAdd(gt + (size_t)idx * F + f, w * df);
  }
}

// Small dense levels (16^3: 16 KiB of F = 1 entries, 64 KiB of F = 4): every sample of a batch lands in the same few
// hundred 64-byte rows, and adds to one row serialise at the memory side (152 us of the 297 us of a 65 536-sample
// level-0 call went to the 16^3 level).  A workgroup sums its share of the points into an LDS copy of the table
// (ds_add_f32) and adds the copy to HBM once, as contiguous 256-byte atomic wave-instructions (the full-rate shape).
template <int F>
__global__ __launch_bounds__(256) void k_grid_scatter_small(RcGridScatterArgs a, int l) {
  extern __shared__ float tab[];
  const RcGridLevel L = a.grid.lvl[l];
  const int total = (int)L.entries * F;
  for (int i = threadIdx.x; i < total; i += 256) tab[i] = 0.0f;
  __syncthreads();
  const int64_t per = (a.n + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < a.n ? p0 + per : a.n;
  const float N = (float)L.size;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    float x = a.points[3 * p], y = a.points[3 * p + 1], z = a.points[3 * p + 2];
    if (a.contract_radius > 0.0f) contract3(x, y, z, a.contract_radius);
    float df[F];
#pragma unroll
    for (int f = 0; f < F; ++f)
      df[f] = (a.point_major ? a.dfeat[p * (a.grid.num_levels * F) + l * F + f] : a.dfeat[(int64_t)(l * F + f) * a.ld + p]) * a.grid.precondition;
    const float loc[3] = {(unit_box(a.grid.bbox, z) * N - 0.5f) + 1.0f, (unit_box(a.grid.bbox, y) * N - 0.5f) + 1.0f,
                          (unit_box(a.grid.bbox, x) * N - 0.5f) + 1.0f};
    float cw[3], fw[3];
    int base[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { const float fl = floorf(loc[ax]); cw[ax] = loc[ax] - fl; fw[ax] = 1.0f - cw[ax]; base[ax] = (int)fl; }
#pragma unroll
    for (int cnr = 0; cnr < 8; ++cnr) {
      const int b0 = (cnr >> 2) & 1, b1 = (cnr >> 1) & 1, b2 = cnr & 1;
      const float w = ((b0 ? cw[0] : fw[0]) * (b1 ? cw[1] : fw[1])) * (b2 ? cw[2] : fw[2]);
      const int k0 = min(max(base[0] + b0, 0), L.size + 1), k1 = min(max(base[1] + b1, 0), L.size + 1),
                k2 = min(max(base[2] + b2, 0), L.size + 1);
      if ((k0 < 1) | (k0 > L.size) | (k1 < 1) | (k1 > L.size) | (k2 < 1) | (k2 > L.size)) continue;
      const int idx = ((k2 - 1) * L.size + (k1 - 1)) * L.size + (k0 - 1);
#pragma unroll
      for (int f = 0; f < F; ++f) unsafeAtomicAdd(&tab[idx * F + f], w * df[f]);
    }
  }
  __syncthreads();
  float* __restrict__ gt = a.gtable[l];
  // Flush by whole 64-byte rows: a row (16 lanes) is added if any of its entries is non-zero, all 16 lanes then take part --
  // the memory side works on 64-byte rows, and an instruction of complete rows runs at the contiguous rate where single
  // entries of a row picked by `v != 0` run at the scattered one (the 32^3 level: 49 -> 20 us).
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < total; i += 256) {      // total is a multiple of 256 for every table that fits (16^3, 32^3)
    const float v = tab[i];
    const unsigned long long nz = __ballot(v != 0.0f);
    if ((nz >> (lane & 48)) & 0xFFFFull) unsafeAtomicAdd(gt + i, v);
  }
}


// Levels with many adds per 64-byte row, ONE launch, no scattered atomics: a level's gradient table is cut into slices of
// 32 768 floats (128 KiB of LDS; F = 1: 16^3 and 32^3 are one slice, 64^3 eight, a 2^19-entry hash table sixteen; F = 4:
// 16^3 one, 32^3 four, 64^3 thirty-two) and the points into `parts` ranges; workgroup (level, slice, part) walks its range,
// computes the eight corner indices of every point and sums the corners that fall into ITS slice in LDS (ds_add_f32),
// then adds the slice to HBM as whole 64-byte rows.  Why: the memory side executes float atomics per request -- 18.5 G
// requests/s whatever they carry (k_grid_scatter<4>: 16-byte entries, k_grid_scatter<1>: 4-byte lanes, the x-pair lanes
// are not merged) -- and a batch of 65 536 samples puts sixteen adds into every row of a hashed F = 1 table (32 768
// samples: four to thirty-two into every row of the dense F = 4 levels): summed on chip first, a row costs `parts`
// full-row requests instead.  The index arithmetic is redone once per slice: ~80 instructions per point and slice against
// a memory-side request.  (Hashed F = 4 levels: two adds per row -- nothing to merge, they stay with k_grid_scatter<4>.)
struct RcSlicedPlan {
  int32_t wg_base[RC_MAX_GRID_LEVELS + 1];     // workgroups [wg_base[l], wg_base[l + 1]) belong to level l
  int32_t nslice[RC_MAX_GRID_LEVELS];          // 0: the level is not part of this launch
  int32_t nparts[RC_MAX_GRID_LEVELS];
};
constexpr int kSliceFloats = 32768;
constexpr int kSliceThreads = 1024;

template <int F>
__global__ __launch_bounds__(kSliceThreads) void k_grid_scatter_sliced(RcGridScatterArgs a, RcSlicedPlan plan) {
  extern __shared__ float tab[];
  constexpr uint32_t kEnt = kSliceFloats / F;      // entries per slice
  int l = 0;
  while (l + 1 < a.grid.num_levels && (int)blockIdx.x >= plan.wg_base[l + 1]) ++l;
  const int local = (int)blockIdx.x - plan.wg_base[l];
  const int slice = local % plan.nslice[l], part = local / plan.nslice[l];
  const RcGridLevel L = a.grid.lvl[l];
  const uint32_t lo = (uint32_t)slice * kEnt;
  const uint32_t cnt = L.entries - lo < kEnt ? L.entries - lo : kEnt;      // entries of this slice
  const int count = (int)cnt * F;
  for (int i = threadIdx.x; i < count; i += kSliceThreads) tab[i] = 0.0f;
  __syncthreads();
  const int64_t per = (a.n + plan.nparts[l] - 1) / plan.nparts[l];
  const int64_t p0 = (int64_t)part * per, p1 = p0 + per < a.n ? p0 + per : a.n;
  const float N = (float)L.size;
  const int LF = a.grid.num_levels * F;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += kSliceThreads) {
    float x = a.points[3 * p], y = a.points[3 * p + 1], z = a.points[3 * p + 2];
    if (a.contract_radius > 0.0f) contract3(x, y, z, a.contract_radius);
    float df[F];
#pragma unroll
    for (int f = 0; f < F; ++f)
      df[f] = (a.point_major ? a.dfeat[p * LF + l * F + f] : a.dfeat[(int64_t)(l * F + f) * a.ld + p]) * a.grid.precondition;
    const float cx = unit_box(a.grid.bbox, x) * N, cy = unit_box(a.grid.bbox, y) * N, cz = unit_box(a.grid.bbox, z) * N;
    float cw[3], fw[3];
    int base[3];
    if (L.dense) {
      const float loc[3] = {(cz - 0.5f) + 1.0f, (cy - 0.5f) + 1.0f, (cx - 0.5f) + 1.0f};
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { const float fl = floorf(loc[ax]); cw[ax] = loc[ax] - fl; fw[ax] = 1.0f - cw[ax]; base[ax] = (int)fl; }
    } else {
      const float loc[3] = {cx - 0.5f, cy - 0.5f, cz - 0.5f};
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { const float fl = floorf(loc[ax]); cw[ax] = loc[ax] - fl; fw[ax] = 1.0f - cw[ax]; base[ax] = (int)fl; }
    }
#pragma unroll
    for (int cnr = 0; cnr < 8; ++cnr) {
      const int b0 = (cnr >> 2) & 1, b1 = (cnr >> 1) & 1, b2 = cnr & 1;
      const float w = ((b0 ? cw[0] : fw[0]) * (b1 ? cw[1] : fw[1])) * (b2 ? cw[2] : fw[2]);
      uint32_t idx;
      if (L.dense) {
        const int k0 = min(max(base[0] + b0, 0), L.size + 1), k1 = min(max(base[1] + b1, 0), L.size + 1),
                  k2 = min(max(base[2] + b2, 0), L.size + 1);
        if ((k0 < 1) | (k0 > L.size) | (k1 < 1) | (k1 > L.size) | (k2 < 1) | (k2 > L.size)) continue;   // zero padding
        idx = ((uint32_t)(k2 - 1) * (uint32_t)L.size + (uint32_t)(k1 - 1)) * (uint32_t)L.size + (uint32_t)(k0 - 1);
      } else {
        idx = (((uint32_t)base[0] + (uint32_t)b0) ^ (((uint32_t)base[1] + (uint32_t)b1) * kPi2) ^
               (((uint32_t)base[2] + (uint32_t)b2) * kPi3)) & L.mask;
      }
      if (idx - lo < cnt) {
#pragma unroll
        for (int f = 0; f < F; ++f) unsafeAtomicAdd(&tab[(idx - lo) * F + f], w * df[f]);
      }
    }
  }
  __syncthreads();
  float* __restrict__ gt = a.gtable[l] + (size_t)lo * F;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < count; i += kSliceThreads) {      // count is a multiple of 64 (checked by the host)
    const float v = tab[i];
    const unsigned long long nz = __ballot(v != 0.0f);
    if ((nz >> (lane & 48)) & 0xFFFFull) unsafeAtomicAdd(gt + i, v);          // whole 64-byte rows (k_grid_scatter_small)
  }
}

}  // namespace

void rc_launch_density_bwd(const RcDensityBwdArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kWaves - 1) / kWaves)), block(kWaves * 64);
  switch ((a.K + 1) / 2 + 1) {
    case 4: hipLaunchKernelGGL((k_density_bwd<4>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_density_bwd<5>), grid, block, 0, stream, a); break;
    case 17: hipLaunchKernelGGL((k_density_bwd<17>), grid, block, 0, stream, a); break;
    default: break;   // rejected by the host before getting here
  }
}

int rc_wgrad_partial_floats(int nwaves) { return nwaves / 4 * kPartStride; }

// waves: enough to fill the chip once (1024 = 256 CUs x 4 SIMDs), at least 16 k-steps (32 points) each
int rc_wgrad_waves(int64_t n) {
  const int64_t nsteps = (n + 1) / 2;
  int64_t w = (nsteps + 15) / 16;
  if (w > 1024) w = 1024;
  if (w < 1) w = 1;
  return (int)((w + 3) / 4 * 4);
}

void rc_launch_wgrad(RcWgradArgs a, int K, float* grads, hipStream_t stream) {
  if (a.n <= 0) return;
  const int nwaves = rc_wgrad_waves(a.n);
  const int64_t nsteps = (a.n + 1) / 2;
  a.steps_per_wave = (nsteps + nwaves - 1) / nwaves;
  a.K = K;
  hipLaunchKernelGGL(k_wgrad, dim3(nwaves / 4), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_grad_reduce, dim3((kPartBO + 16) / 16), dim3(256), 0, stream, a.partial, nwaves / 4, K, grads);
}

// LDS opt-in of the scatter kernels beyond 64 KiB, once per device (not from inside a stream capture: rc_density_backward
// calls it before it captures).  false: the opt-in was refused, the callers keep to the kernels that need none.
bool rc_train_prepare() {
  static std::atomic<uint64_t> prepared{0}, good{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (rc_first_use_on_device(prepared)) {
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter_sliced<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kSliceFloats * 4) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter_sliced<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kSliceFloats * 4) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter_small<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter_small<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (ok) good.fetch_or(bit);
  }
  return (good.load() & bit) != 0;
}

void rc_launch_grid_scatter(const RcGridScatterArgs& a, hipStream_t stream, hipStream_t small_stream, bool use_small_stream) {
  const hipStream_t sst = use_small_stream ? small_stream : stream;
  if (a.n <= 0) return;
  const int G = a.grid.num_features == 4 ? 4 : 2;
  dim3 grid((unsigned)((a.n * G + 255) / 256), (unsigned)a.grid.num_levels), block(256);
  static const bool sliced_on = !(getenv("RC_SCATTER_SLICED") && getenv("RC_SCATTER_SLICED")[0] == '0');
  const int F = a.grid.num_features;
  uint32_t sliced_levels = 0;
  // RC_SCATTER_SLICED=4: also the dense levels of an F = 4 grid (experiment: measured slower, below)
  static const bool sliced_f4 = getenv("RC_SCATTER_SLICED") && getenv("RC_SCATTER_SLICED")[0] == '4';
  if ((F == 1 || (F == 4 && sliced_f4)) && sliced_on && rc_train_prepare()) {
    // through k_grid_scatter_sliced: every level of an F = 1 grid (power-of-two hash tables).  The dense levels of an F = 4
    // grid (4-32 adds per row at 32 768 samples) were tried too: 37 slices, every point walked once per slice with four
    // LDS adds per corner -- the level-2 call went from 0.199 to 0.280 ms; they stay with k_grid_scatter<4> / _small<4>.
    int big_slices = 0, n_small = 0;
    RcSlicedPlan plan{};
    for (int l = 0; l < a.grid.num_levels; ++l) {
      const RcGridLevel& L = a.grid.lvl[l];
      const int64_t floats = (int64_t)L.entries * F;
      const bool take = floats % 64 == 0 && (L.dense ? floats <= (1 << 20) : (F == 1 && L.mask != 0));
      if (!take) continue;
      sliced_levels |= 1u << l;
      plan.nslice[l] = (int)((floats + kSliceFloats - 1) / kSliceFloats);
      if (plan.nslice[l] > 1) big_slices += plan.nslice[l]; else ++n_small;
    }
    if (sliced_levels) {
      // about one workgroup per CU (128 KiB of LDS each): big tables 2-4 point ranges per slice, one-slice tables (heavy
      // reuse, their LDS adds are the long pole) what is left, 4-32 ranges each
      const int cus = rc_device_cus();
      int pb = big_slices ? (cus - cus / 8) / big_slices : 1;
      pb = pb < 2 ? 2 : (pb > 4 ? 4 : pb);
      int ps = n_small ? (cus - big_slices * pb) / n_small : 1;
      ps = ps < 4 ? 4 : (ps > 32 ? 32 : ps);
      int wg = 0;
      for (int l = 0; l < a.grid.num_levels; ++l) {
        plan.wg_base[l] = wg;
        plan.nparts[l] = plan.nslice[l] > 1 ? pb : ps;
        wg += plan.nslice[l] * plan.nparts[l];
      }
      for (int l = a.grid.num_levels; l <= RC_MAX_GRID_LEVELS; ++l) plan.wg_base[l] = wg;
      const bool all = sliced_levels == (1u << a.grid.num_levels) - 1u;
      const hipStream_t q = all ? stream : sst;      // beside the other levels' scatter when there are any
      if (F == 4) hipLaunchKernelGGL(k_grid_scatter_sliced<4>, dim3((unsigned)wg), dim3(kSliceThreads), kSliceFloats * 4, q, a, plan);
      else hipLaunchKernelGGL(k_grid_scatter_sliced<1>, dim3((unsigned)wg), dim3(kSliceThreads), kSliceFloats * 4, q, a, plan);
      if (all) return;
    }
  }
  RcGridScatterArgs b = a;
  b.level0 = 0;
  b.lds_levels = sliced_levels;
  // experiment switches (timing only): RC_SCATTER_SKIP = bit mask of levels whose gradient is NOT computed;
  // RC_SCATTER_LDS_MAX = largest table (floats) summed in LDS (default 32768 = 128 KiB: 16^3 at F = 1 | 4, 32^3 at F = 1)
  static const unsigned skip_levels = getenv("RC_SCATTER_SKIP") ? (unsigned)strtoul(getenv("RC_SCATTER_SKIP"), nullptr, 0) : 0u;
  static const int64_t lds_max = getenv("RC_SCATTER_LDS_MAX") ? atoll(getenv("RC_SCATTER_LDS_MAX")) : 32768;
  for (int l = 0; l < a.grid.num_levels; ++l) {
    const RcGridLevel& L = a.grid.lvl[l];
    if ((sliced_levels >> l) & 1u) continue;
    if ((skip_levels >> l) & 1u) { b.lds_levels |= 1u << l; continue; }
    if (!L.dense || (int64_t)L.entries * a.grid.num_features > lds_max) continue;
    const int lds = (int)L.entries * a.grid.num_features * (int)sizeof(float);
    if (lds > 65536 && !rc_train_prepare()) continue;       // no LDS opt-in: the level goes through the global scatter
    b.lds_levels |= 1u << l;
    int wgs = (int)((a.n + 255) / 256);            // one table flush per workgroup: at most one per CU
    if (wgs > 256) wgs = 256;
    if (lds > 65536) {                              // beyond the default LDS limit
      static const int wg_big = getenv("RC_SCATTER_BIG_WGS") ? atoi(getenv("RC_SCATTER_BIG_WGS")) : 256;
      if (wgs > wg_big) wgs = wg_big;               // (experiment knob; 16 ... 256 workgroups measured: 256 is the fastest)
    }
    if (a.grid.num_features == 4) hipLaunchKernelGGL((k_grid_scatter_small<4>), dim3(wgs), dim3(256), lds, sst, b, l);
    else hipLaunchKernelGGL((k_grid_scatter_small<1>), dim3(wgs), dim3(256), lds, sst, b, l);
  }
  const RcGridScatterArgs& a2 = b;
  if (a.grid.num_features == 4) hipLaunchKernelGGL((k_grid_scatter<4>), grid, block, 0, stream, a2);
  else hipLaunchKernelGGL((k_grid_scatter<1>), grid, block, 0, stream, a2);
}
