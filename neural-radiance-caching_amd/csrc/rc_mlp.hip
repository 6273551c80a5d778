// Dense cache MLPs on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces (reference file:line):
//   DensityMLP.run_network / convert_raw_density / pred normals   internal/geometry.py:155-168, 318-341, 467-471
//   NeRFMLP heads, get_integrated_brdf, _predict_appearance_passive internal/nerf.py:461-482, 628-634, 940-1090
//   SurfaceLightFieldMLP.run_surface_lightfield_network + heads    internal/surface_light_field.py:480-499, 1011-1059
//   ref_utils.generate_ide_fn, reflect                              internal/ref_utils.py:25-42, 131-192
//
// Formulation.  Every layer is computed transposed, Y^T = W^T X^T, so that the 32 points of a wave
// sit on the MFMA's N/lane axis and the features live along registers:
//   A operand = weight fragment  (lane l: W[k = step row of half l>>5][n = 32 t + (l & 31)])
//   B operand = activations      (lane l: X[point l & 31][k = step row of half l>>5])
//   D         = 32 features x 32 points, lane l reg r holds feature (r&3) + 8 (r>>2) + 4 (l>>5).
// A layer's output accumulators are therefore *already* in B-operand form for the next layer: no
// transposes, no cross-lane traffic.  The k order inside a layer follows the accumulator layout;
// the host packs every weight matrix into exactly that fragment order once at load time
// (rc_api.hip), so a layer is one linear stream of 256-byte fragments.  Bias is one extra k-step
// (B = 1 on the low half-wave), ReLU is a VALU max on the accumulators.
// Activations are parked in a per-wave LDS slice between layers (each lane only ever re-reads what
// it wrote itself, so no barriers are needed) which keeps the register file for accumulators and
// the prefetched weight fragments.
#include "rc_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;   // waves per workgroup; each wave owns 32 points

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
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
};

// Issue the LDS-DMA of chunk c (this wave's quarter: 4 x 1 KiB).
template <int NF, int W = kWaves>
__device__ __forceinline__ void ws_issue(const WStream& w, int c) {
#pragma unroll
  for (int k = 0; k < kChunk / 4 / W; ++k) {
    const int i = w.wave + W * k;                  // 1-KiB piece inside the chunk
    const int frag0 = c * kChunk + 4 * i;
    if (frag0 < NF) {
      const float* src = w.g + (size_t)frag0 * 64 + w.lane * 4;
      float* dst = w.ring + ((c & 1) * kChunk + 4 * i) * 64;
      __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_ptr)dst, 16, 0, 0);
    }
  }
}

template <int NF, int W = kWaves>
__device__ __forceinline__ void ws_begin(const WStream& w) {
  ws_issue<NF, W>(w, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (kChunk < NF) ws_issue<NF, W>(w, 1);
}

// Entering chunk c: it has landed (issued one chunk ago), everybody is done with chunk c-1.
template <int NF, int W = kWaves>
__device__ __forceinline__ void ws_advance(const WStream& w, int c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((c + 1) * kChunk < NF) ws_issue<NF, W>(w, c + 1);
}

// Read fragment f of the stream as a per-lane value (used for lane-layout constant vectors).
template <int NF, int W = kWaves>
__device__ __forceinline__ float ws_read(const WStream& w, int f) {
  if (f > 0 && f % kChunk == 0) ws_advance<NF, W>(w, f / kChunk);
  return w.ring[(f % (2 * kChunk)) * 64 + w.lane];
}

// One pass over KS k-steps for NT output tiles; the layer's fragments are [FBASE, FBASE + KS*NT)
// of the stream.  act: this lane's activation column (act[s * 64] is step s).
// Software pipelined in groups of SG k-steps: the LDS reads (A fragments + B activations) of group
// g+1 are issued before the MFMAs of group g, with scheduling fences so they stay there; the MFMA
// pipe then runs back to back while the next operands are in flight.
template <int NT, int KS, int FBASE, int NF, int SG = (NT >= 8 ? 1 : (NT >= 4 ? 2 : (NT >= 2 ? 4 : 8))), int W = kWaves>
__device__ __forceinline__ void mlp_layer(const WStream& w, const float* act, f32x16 (&acc)[NT]) {
  constexpr int NG = (KS + SG - 1) / SG;
  float a[2][SG][NT], b[2][SG];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int d = 0; d < SG; ++d) {
      const int s = g * SG + d;
      if (s < KS) {
        b[buf][d] = act[s * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int f = FBASE + s * NT + t;            // compile-time after unrolling
          if (f > 0 && f % kChunk == 0) ws_advance<NF, W>(w, f / kChunk);
          a[buf][d][t] = w.ring[(f % (2 * kChunk)) * 64 + w.lane];
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
  load(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    comp(g, g & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Park NT accumulator tiles as the next layer's activation steps [base, base + 16 NT).
template <int NT, bool RELU>
__device__ __forceinline__ void park(const f32x16 (&acc)[NT], float* act, int base) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[t][r];
      act[(base + t * 16 + r) * 64] = RELU ? fmaxf(v, 0.0f) : v;
    }
}

__device__ __forceinline__ float softplus(float x) {
  // jax.nn.softplus = logaddexp(x, 0) = max(x, 0) + log1p(exp(-|x|))
  return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void contract3(float& x, float& y, float& z, float radius) {
  x = x / radius; y = y / radius; z = z / radius;
  float mag = x * x + y * y + z * z;
  mag = fmaxf(1.0f, mag);
  const float scale = (2.0f * sqrtf(mag) - 1.0f) / mag;
  x = scale * x; y = scale * y; z = scale * z;
}

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

// ---------------------------------------------------------------------------------------------
// Density MLP: K -> 64 -> 64 -> 1 (+ 64 -> 3 predicted normals on the last level)
// ---------------------------------------------------------------------------------------------
constexpr int kDensActSteps = 33;

// GRAD (last level only): also the analytic normals -normalize(d raw / d x) of geometry.py:421-460:
// backward through the two hidden layers on the matrix cores (transposed weights, ReLU masks kept as
// one bit per accumulator register), then the trilinear Jacobian emitted by k_hashgrid_fwd<F, true>
// and the Jacobian of the contraction.
template <int KS0, bool GRAD>   // KS0: k-steps of layer 0 including the bias step
__global__ __launch_bounds__(kWaves * 64) void k_density_mlp(RcDensityMlpArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[kRingFloats];
  __shared__ float lds[kWaves][kDensActSteps * 64];
  constexpr int F_D0 = 0, F_D1 = 2 * KS0, F_DO = 2 * KS0 + 66, F_WO = F_DO + 33, F_B1 = F_WO + 32,
                F_B0 = F_B1 + 64, NF = GRAD ? F_B0 + 32 : F_WO;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t p0 = tile * 32;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = p0 + j;
  const bool valid = p < a.n;        // waves past the end stay alive for the workgroup barriers
  float* act = &lds[wave][lane];
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF>(ws);

  // stage the grid features (natural k pairs) + bias step
#pragma unroll
  for (int s = 0; s < KS0 - 1; ++s) {
    const int k = 2 * s + h;
    act[s * 64] = (valid && k < a.K) ? a.feat[(int64_t)k * a.ld + p] : 0.0f;
  }
  act[(KS0 - 1) * 64] = h == 0 ? 1.0f : 0.0f;

  f32x16 acc[2];
  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer<2, KS0, F_D0, NF>(ws, act, acc);
  uint32_t m0 = 0, m1 = 0;           // ReLU masks, bit t*16+r
  if constexpr (GRAD) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
  }
  park<2, true>(acc, act, 0);
  act[32 * 64] = h == 0 ? 1.0f : 0.0f;

  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer<2, 33, F_D1, NF>(ws, act, acc);
  if constexpr (GRAD) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
  }
  park<2, true>(acc, act, 0);
  if (a.last && a.hbuf && p0 < a.n) {
    // hidden feature handed to the shader in accumulator (= B operand) layout
    float* hb = a.hbuf + tile * (32 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hb[(t * 16 + r) * 64] = fmaxf(acc[t][r], 0.0f);
  }

  f32x16 out[1];
  out[0] = zero16();
  mlp_layer<1, 33, F_DO, NF>(ws, act, out);

  float cx = 0.0f, cy = 0.0f, cz = 0.0f;   // contracted position
  float zx = 0.0f, zy = 0.0f, zz = 0.0f;   // x / radius
  if (valid) {
    zx = a.means[p] / a.contract_radius; zy = a.means[a.n + p] / a.contract_radius; zz = a.means[2 * a.n + p] / a.contract_radius;
    cx = a.means[p]; cy = a.means[a.n + p]; cz = a.means[2 * a.n + p];
    contract3(cx, cy, cz, a.contract_radius);
  }
  if (h == 0 && valid) {
    // convert_raw_density (geometry.py:318-341)
    const float raw = out[0][0];
    const bool inside = (cx > -a.bbox) & (cx < a.bbox) & (cy > -a.bbox) & (cy < a.bbox) & (cz > -a.bbox) & (cz < a.bbox);
    const float d = expf(fminf(fmaxf(raw + a.density_bias, -RC_FMAX), 70.0f));
    a.density[p] = inside ? d : 0.0f;
    if (a.last && a.normals_pred) {
      float gx = out[0][1], gy = out[0][2], gz = out[0][3];
      neg_normalize(gx, gy, gz);
      a.normals_pred[p] = gx; a.normals_pred[a.n + p] = gy; a.normals_pred[2 * a.n + p] = gz;
    }
  }

  if constexpr (GRAD) {
    // d raw / d h1 = w_out (accumulator-layout constant vector from the stream), masked by ReLU'(h1)
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float wv = ws_read<NF>(ws, F_WO + s);
      act[s * 64] = ((m1 >> s) & 1u) ? wv : 0.0f;
    }
    f32x16 g[2];
    g[0] = zero16(); g[1] = zero16();
    mlp_layer<2, 32, F_B1, NF>(ws, act, g);            // W1 . (.)   (transposed layer, no bias)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(t * 16 + r) * 64] = ((m0 >> (t * 16 + r)) & 1u) ? g[t][r] : 0.0f;
    f32x16 gf[1];
    gf[0] = zero16();
    mlp_layer<1, 32, F_B0, NF>(ws, act, gf);           // W0 . (.) -> d raw / d feature (32, accumulator layout)
    // chain through the trilinear Jacobian: this lane owns features acc_feat(0, r, h)
    float gw[3] = {0.0f, 0.0f, 0.0f};
    if (valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) gw[ax] += gf[0][r] * a.jac[(int64_t)(ax * 32 + i) * a.ld + p];
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) gw[ax] += __shfl_xor(gw[ax], 32, 64);
    if (h == 0 && valid) {
      // Jacobian of contract at z = x / radius: w = s(m) z, s = (2 sqrt(m) - 1) / m, m = max(1, |z|^2)
      const float msq = zx * zx + zy * zy + zz * zz;
      float gzx = gw[0], gzy = gw[1], gzz = gw[2];
      if (msq > 1.0f) {
        const float rt = sqrtf(msq);
        const float s = (2.0f * rt - 1.0f) / msq;
        const float ds = (1.0f - rt) / (msq * msq);          // ds/dm
        const float gz_dot = gw[0] * zx + gw[1] * zy + gw[2] * zz;
        gzx = s * gw[0] + 2.0f * ds * gz_dot * zx;
        gzy = s * gw[1] + 2.0f * ds * gz_dot * zy;
        gzz = s * gw[2] + 2.0f * ds * gz_dot * zz;
      }
      float nx = gzx / a.contract_radius, ny = gzy / a.contract_radius, nz = gzz / a.contract_radius;
      neg_normalize(nx, ny, nz);
      a.normals_grad[p] = nx; a.normals_grad[a.n + p] = ny; a.normals_grad[2 * a.n + p] = nz;
    }
  }
}

// (l, m) of IDE term i for deg_view = 5: l in {1,2,4,8,16}, m = 0..l (ref_utils.py:105-115)
__host__ __device__ constexpr int ide_l(int i) { return i < 2 ? 1 : (i < 5 ? 2 : (i < 10 ? 4 : (i < 19 ? 8 : 16))); }
__host__ __device__ constexpr int ide_m(int i) { return i < 2 ? i : (i < 5 ? i - 2 : (i < 10 ? i - 5 : (i < 19 ? i - 10 : i - 19))); }

// ---------------------------------------------------------------------------------------------
// Cache shader
// ---------------------------------------------------------------------------------------------
// activation slice (steps): [0,64) bottleneck | [64,100) IDE | 100 bias(1|0) | 101 (dot|1)
#ifdef RC_STAMPS
#define RC_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif
constexpr int kShActSteps = 102;
constexpr int kStepBias = 100;
constexpr int kStepDot = 101;

__global__ __launch_bounds__(kWaves * 64) void k_cache_shader(RcShaderArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  // fragment offsets of the layers inside the kernel's weight stream (host: rc_api.hip, same order)
  constexpr int F_H = 0, F_S0 = F_H + 49 * 5, F_I0 = F_S0 + 101 * 8, F_I1 = F_I0 + 65 * 2, F_IO = F_I1 + 33 * 2,
                F_S1 = F_IO + 33, F_S2 = F_S1 + 65 * 4, F_SB = F_S2 + 65 * 4, F_SO = F_SB + 64 * 4, NF = F_SO + 65;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t p0 = tile * 32;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = p0 + j;
  const bool valid = p < a.n;        // waves past the end stay alive for the workgroup barriers
  const int64_t pc = valid ? p : a.n - 1;
  const int64_t q = a.src ? (int64_t)a.src[pc] : pc;       // source point of the last level
  const int64_t ray = pc / a.samples_per_ray;
  float* ring = lds_dyn;
  float* act = lds_dyn + kRingFloats + wave * (kShActSteps * 64) + lane;
  WStream ws{a.wstream, ring, lane, wave};
#ifdef RC_STAMPS
  unsigned long long stamps[12];
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  RC_STAMP(0);
  ws_begin<NF>(ws);

  // ---- stage feature = [density feature (64, accumulator order) | appearance grid (32)] + bias
  {
    const float* hb = a.hbuf + (q >> 5) * (32 * 64) + (q & 31) + 32 * h;
#pragma unroll
    for (int s = 0; s < 32; ++s) act[s * 64] = hb[s * 64];
#pragma unroll
    for (int s = 0; s < 16; ++s) act[(32 + s) * 64] = a.app[(int64_t)(2 * s + h) * a.n + pc];
    act[48 * 64] = h == 0 ? 1.0f : 0.0f;
  }
  RC_STAMP(1);
  // ---- heads: bottleneck (4 tiles, linear) + small heads tile
  float rough, tint[3], ad[3], idf[3];
  {
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = zero16();
    mlp_layer<5, 49, F_H, NF>(ws, act, acc);
    // heads tile, by accumulator register (same on both half-waves): 0 roughness, 1-3 tint,
    // 4-6 ambient irradiance, 7-9 irradiance
    rough = softplus(acc[4][0] + a.roughness_bias);                       // nerf.py:633-634
    tint[0] = sigmoidf(acc[4][1]); tint[1] = sigmoidf(acc[4][2]); tint[2] = sigmoidf(acc[4][3]);   // :976
    const float ar[3] = {acc[4][4], acc[4][5], acc[4][6]};
    const float ir[3] = {acc[4][7], acc[4][8], acc[4][9]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ad[c] = fminf(fmaxf(softplus(ar[c] + a.ambient_bias), 0.0f), a.rgb_max);      // nerf.py:965-969
      idf[c] = fminf(fmaxf(softplus(ir[c] + a.irradiance_bias), 0.0f), a.rgb_max);  // nerf.py:1008-1012
    }
    f32x16 bt[4] = {acc[0], acc[1], acc[2], acc[3]};
    park<4, false>(bt, act, 0);
  }
  RC_STAMP(2);
  // ---- normals, n.(-v), reflection direction, IDE
  {
    const float nx = a.normals_pred[q], ny = a.normals_pred[a.n_src + q], nz = a.normals_pred[2 * a.n_src + q];
    const float vx = a.viewdirs[3 * ray], vy = a.viewdirs[3 * ray + 1], vz = a.viewdirs[3 * ray + 2];
    const float dotp = nx * (-vx) + ny * (-vy) + nz * (-vz);        // nerf.py:474
    // reflect(-v, n) = 2 (n . -v) n - (-v)  (ref_utils.py:25-42)
    const float rx = 2.0f * dotp * nx - (-vx), ry = 2.0f * dotp * ny - (-vy), rz = 2.0f * dotp * nz - (-vz);
    act[kStepBias * 64] = h == 0 ? 1.0f : 0.0f;
    act[kStepDot * 64] = h == 0 ? dotp : 1.0f;
    // IDE (ref_utils.py:155-190): low half-wave keeps real parts, high half-wave imaginary parts.
    const RcIdeTable* tb = reinterpret_cast<const RcIdeTable*>(a.ide_coef);
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
      act[(64 + i) * 64] = (cpw[m] * poly) * att;
    }
  }
  // ---- SLF layer_0 (tiles 0-3) + input part of layer_bottleneck (tiles 4-7): one pass over
  //      [bottleneck | IDE | bias]; results stay in registers while the IBRDF chain runs.
  RC_STAMP(3);
  f32x16 s0[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) s0[t] = zero16();
  mlp_layer<8, 101, F_S0, NF>(ws, act, s0);
  RC_STAMP(4);
  // ---- integrated BRDF: (bottleneck, n.v) 129 -> 64 -> 64 -> 1 (nerf.py:461-482)
  float ibrdf;
  {
    f32x16 ib[2];
    ib[0] = zero16(); ib[1] = zero16();
    mlp_layer<2, 64, F_I0, NF>(ws, act, ib);
    mlp_layer<2, 1, F_I0 + 128, NF>(ws, act + kStepDot * 64, ib);   // (n.v | bias) step
    // IDE is dead now: steps [64, 97) are scratch for the IBRDF tail
    park<2, true>(ib, act, 64);
    act[(64 + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    ib[0] = zero16(); ib[1] = zero16();
    mlp_layer<2, 33, F_I1, NF>(ws, act + 64 * 64, ib);
    park<2, true>(ib, act, 64);
    f32x16 o[1];
    o[0] = zero16();
    mlp_layer<1, 33, F_IO, NF>(ws, act + 64 * 64, o);
    ibrdf = sigmoidf(o[0][0] + 1.0986123f);     // + log(3), nerf.py:481
  }
  RC_STAMP(5);
  // ---- SLF trunk: layer_1, layer_2, layer_bottleneck (x part accumulates onto the input part)
  float amb[3];
  {
    f32x16 acc[4] = {s0[0], s0[1], s0[2], s0[3]};
    f32x16 skip[4] = {s0[4], s0[5], s0[6], s0[7]};
    park<4, true>(acc, act, 0);
    act[64 * 64] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    mlp_layer<4, 65, F_S1, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    mlp_layer<4, 65, F_S2, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
    mlp_layer<4, 64, F_SB, NF>(ws, act, skip);
    park<4, true>(skip, act, 0);
    f32x16 o[1];
    o[0] = zero16();
    mlp_layer<1, 65, F_SO, NF>(ws, act, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) amb[c] = fmaxf(softplus(o[0][c] + a.slf_ambient_bias), 0.0f);   // slf.py:1053-1059
  }
  RC_STAMP(6);
#ifdef RC_STAMPS
  if (lane == 0 && a.debug) {
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = reinterpret_cast<unsigned long long*>(a.debug) + tile * 10;
    for (int i = 0; i < 7; ++i) d[i] = stamps[i];
    d[7] = rt0; d[8] = rt1;
  }
#endif
  // ---- combine (nerf.py:1034-1053); ambient_specular is an exact 0 (ref_acc == 1)
  if (h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float is = fminf(fmaxf(tint[c] * ibrdf * (amb[c] * 1.0f), 0.0f), a.rgb_max);
      const float ambient = ad[c] + 0.0f;
      const float indirect = idf[c] + is;
      a.shade[(int64_t)(RC_SH_RGB + c) * a.n + p] = ambient + indirect;
      a.shade[(int64_t)(RC_SH_AD + c) * a.n + p] = ad[c];
      a.shade[(int64_t)(RC_SH_ID + c) * a.n + p] = idf[c];
      a.shade[(int64_t)(RC_SH_IS + c) * a.n + p] = is;
      a.shade[(int64_t)(RC_SH_TINT + c) * a.n + p] = tint[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Model-level EnvMap (background of secondary rays): pos_enc(dir, 0, 4) [27] -> 256 -> 256 -> 256
// -> concat input (283) -> 128 -> rgba; rgb = clip(softplus(raw + rgb_bias), 0, inf).
// Replaces Model._handle_env_map -> SurfaceLightFieldMLP.__call__ as configured by
// NeRFModel.env_map_params (internal/models.py:360-421, internal/surface_light_field.py:480-499,
// 1011-1058, internal/coord.py:298-312).  One wave = 32 rays; 2 waves per workgroup (the 256-wide
// activations need 129 LDS steps per wave).
// ---------------------------------------------------------------------------------------------
constexpr int kEnvWaves = 2;
constexpr int kEnvActSteps = 130;

__global__ __launch_bounds__(kEnvWaves * 64) void k_envmap(RcEnvMapArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  constexpr int KS_IN = 15;   // 14 natural pairs of the 27 inputs (+1 zero pad) + bias
  constexpr int F_E0 = 0, F_E1 = F_E0 + KS_IN * 8, F_E2 = F_E1 + 129 * 8, F_EB = F_E2 + 129 * 8,
                F_EI = F_EB + 128 * 4, F_EO = F_EI + KS_IN * 4, NF = F_EO + 65;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kEnvWaves + wave;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = tile * 32 + j;
  const bool valid = p < a.n;
  const int64_t pc = valid ? p : a.n - 1;
  float* ring = lds_dyn;
  float* act = lds_dyn + kRingFloats + wave * (kEnvActSteps * 64) + lane;
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF, kEnvWaves>(ws);

  // pos_enc(x, 0, 4, append_identity): [x(3), sin(2^j x)(12), sin(2^j x + pi/2)(12)]
  const float d[3] = {a.viewdirs[3 * pc], a.viewdirs[3 * pc + 1], a.viewdirs[3 * pc + 2]};
  auto enc = [&](int k) -> float {
    if (k < 3) return d[k];
    if (k >= 27) return 0.0f;
    const int q = (k - 3) % 12, second = (k - 3) / 12;
    const float sx = d[q % 3] * (float)(1 << (q / 3));
    return sinf(second ? sx + 1.5707963267948966f : sx);
  };
  auto stage_inputs = [&]() {
#pragma unroll
    for (int s = 0; s < KS_IN - 1; ++s) act[s * 64] = enc(2 * s + h);
    act[(KS_IN - 1) * 64] = h == 0 ? 1.0f : 0.0f;
  };
  stage_inputs();
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, KS_IN, F_E0, NF, 1, kEnvWaves>(ws, act, acc);
  park<8, true>(acc, act, 0);
  act[128 * 64] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, 129, F_E1, NF, 1, kEnvWaves>(ws, act, acc);
  park<8, true>(acc, act, 0);
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, 129, F_E2, NF, 1, kEnvWaves>(ws, act, acc);
  park<8, true>(acc, act, 0);
  // layer_bottleneck on concat([x2 (256), inputs (27)]): x part, then the re-staged input part (+bias)
  f32x16 bt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bt[t] = zero16();
  mlp_layer<4, 128, F_EB, NF, 2, kEnvWaves>(ws, act, bt);
  stage_inputs();
  mlp_layer<4, KS_IN, F_EI, NF, 2, kEnvWaves>(ws, act, bt);
  park<4, true>(bt, act, 0);
  act[64 * 64] = h == 0 ? 1.0f : 0.0f;
  f32x16 o[1];
  o[0] = zero16();
  mlp_layer<1, 65, F_EO, NF, 8, kEnvWaves>(ws, act, o);
  if (h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.env_rgb[3 * p + c] = fmaxf(softplus(o[0][c] + a.rgb_bias), 0.0f);
  }
}

}  // namespace

void rc_launch_density_mlp(const RcDensityMlpArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kWaves - 1) / kWaves)), block(kWaves * 64);
  const int ks0 = (a.K + 1) / 2 + 1;
  switch (ks0) {
    case 4: hipLaunchKernelGGL((k_density_mlp<4, false>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_density_mlp<5, false>), grid, block, 0, stream, a); break;
    case 17:
      if (a.jac && a.normals_grad) hipLaunchKernelGGL((k_density_mlp<17, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((k_density_mlp<17, false>), grid, block, 0, stream, a);
      break;
    default: break;   // rejected by the host before getting here
  }
}

int rc_shader_lds_bytes() { return (kRingFloats + kWaves * kShActSteps * 64) * (int)sizeof(float); }
int rc_weight_chunk_floats() { return kChunk * 64; }

// The shader's per-wave activation slices exceed the default 64 KiB dynamic-LDS limit.
void rc_shader_prepare() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_shader),
                            hipFuncAttributeMaxDynamicSharedMemorySize, rc_shader_lds_bytes());
  done = true;
}

void rc_launch_shader(const RcShaderArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kWaves - 1) / kWaves)), block(kWaves * 64);
  hipLaunchKernelGGL(k_cache_shader, grid, block, rc_shader_lds_bytes(), stream, a);
}

void rc_launch_envmap(const RcEnvMapArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  static bool prepared = false;
  const int lds = (kRingFloats + kEnvWaves * kEnvActSteps * 64) * (int)sizeof(float);
  if (!prepared) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_envmap), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    prepared = true;
  }
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kEnvWaves - 1) / kEnvWaves)), block(kEnvWaves * 64);
  hipLaunchKernelGGL(k_envmap, grid, block, lds, stream, a);
}
