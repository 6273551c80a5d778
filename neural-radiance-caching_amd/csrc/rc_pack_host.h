// Host-only part of the weight packing (no HIP in this header): Flax layers -> MFMA A-fragment streams, the folded
// shader bottleneck, the directional-encoding coefficient table, and the index rule of the dense levels' cell tables.
// rc_api.hip uses it at rc_load_weights time; csrc/hostcheck.cpp compiles the same code for the CPU with
// -fsanitize=address,undefined and tests/test_hostcheck.py compares its results with numpy restatements
// (`make hostcheck`: SURVEY 5, sanitizers on the CPU build only).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

#ifndef RC_IDE_TERMS
#define RC_IDE_TERMS 36
#define RC_IDE_ZPOW 17
struct RcIdeTable {
  float coef[RC_IDE_TERMS][RC_IDE_ZPOW];
  int32_t m[RC_IDE_TERMS];
  float sigma[RC_IDE_TERMS];
};
#endif

#if defined(__HIPCC__)
#define RC_HD __host__ __device__
#else
#define RC_HD
#endif

// Cell tables of the dense levels (rc_fused.hip k_build_cells): record i = cell * 8 + corner of the zero-padded
// volume, (N + 3)^3 cells whose origin runs over [-1, N + 1] per axis; corner bits (b0 b1 b2) in combine order.
// Returns true and the entry index ((k2 - 1) N + (k1 - 1)) N + (k0 - 1) of the source level when the corner lies
// inside the level, false for the zero padding (grid_utils.py:384-390).
RC_HD inline bool rc_cell_corner(int N, int64_t i, int64_t* entry) {
  const int M = N + 3;
  const int c = (int)(i & 7);
  int64_t cell = i >> 3;
  const int q0 = (int)(cell % M); cell /= M;
  const int q1 = (int)(cell % M);
  const int q2 = (int)(cell / M);
  const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
  const int lo = 0, hi = N + 1;
  int k0 = q0 - 1 + b0, k1 = q1 - 1 + b1, k2 = q2 - 1 + b2;
  k0 = k0 < lo ? lo : (k0 > hi ? hi : k0); k1 = k1 < lo ? lo : (k1 > hi ? hi : k1); k2 = k2 < lo ? lo : (k2 > hi ? hi : k2);
  const bool inside = (k0 >= 1) & (k0 <= N) & (k1 >= 1) & (k1 <= N) & (k2 >= 1) & (k2 <= N);
  *entry = ((int64_t)(k2 - 1) * N + (k1 - 1)) * N + (k0 - 1);
  return inside;
}


// ---------------------------------------------------------------------------------------------
// Arithmetic of the MLP layers (compile-time, the whole library): RC_SPLIT_MFMA = 1 (default) runs every weight layer on
// the bf16 matrix pipe with each fp32 operand split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits:
// hi + mid + lo == x bit for bit) and the six piece products whose weight is >= 2^-16 of the full product
// (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; what is dropped is <= 2^-23 of |a b|, the size of one fp32 rounding),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k where v_mfma_f32_32x32x2_f32 takes 8 x 64
// (tools/micro/split_mfma.hip: 1.9 x the layer rate at the same error against fp64).  RC_SPLIT_MFMA = 0 is the exact
// fp32 MFMA chain of rounds 1-4 (`make diag DIAG_EXTRA=-DRC_SPLIT_MFMA=0`).
// Stream geometry: a layer of KS k-steps (= pairs of inputs) and NT 32-row tiles is KS * NT fragments of 256 bytes in
// the fp32 form; in the split form ceil(KS / 8) blocks x NT tiles x 3 pieces x 1 KiB (= 4 fragments each: lane l holds
// the 8 bf16 of steps 8 q .. 8 q + 7, half l / 32, row l % 32).
// ---------------------------------------------------------------------------------------------
#ifndef RC_SPLIT_MFMA
#define RC_SPLIT_MFMA 1
#endif
constexpr bool kRcSplit = RC_SPLIT_MFMA != 0;
// fragments of a weight layer / of a dot_out block (NO outputs over NT hidden tiles + NO bias fragments; padded to whole
// 1-KiB pieces in the split form so that every layer behind it starts on one)
constexpr int rc_lfr(int ks, int nt) { return kRcSplit ? ((ks + 7) / 8) * nt * 12 : ks * nt; }
constexpr int rc_dfr(int no, int nt) { return kRcSplit ? ((no * (nt * 16 + 1) + 3) / 4) * 4 : no * (nt * 16 + 1); }
// The density MLPs of the proposal levels stay on the fp32 MFMA in every build (mlp_layer_d, rc_dev_mlp.h: with them in
// the split form the two-wave kernel was unstable -- see INSTABILITY there): their
// layers and dot blocks keep the fp32 geometry, and a stream that continues with split layers is padded to a whole piece.
constexpr int rc_lfr32(int ks, int nt) { return ks * nt; }
constexpr int rc_dfr32(int no, int nt) { return no * (nt * 16 + 1); }
constexpr int rc_align_piece(int f) { return kRcSplit ? (f + 3) / 4 * 4 : f; }

namespace rcpack {

struct HostLayer {
  std::vector<float> kernel, bias;
  int in = 0, out = 0;
  bool have_kernel = false, have_bias = false;
};

// ---------------------------------------------------------------------------------------------
// MFMA fragment packing
// ---------------------------------------------------------------------------------------------
struct Step { int row[2]; };               // >= 0 input row, -1 zero, -2 bias
struct Col {
  const HostLayer* L = nullptr;            // nullptr -> zero column
  int col = 0;
  int row_off = 0;
  bool bias_ok = true;
};
using Tile = std::array<Col, 32>;

inline int acc_feat(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

inline void steps_natural(std::vector<Step>& s, int K, int base) {
  for (int i = 0; i < (K + 1) / 2; ++i) s.push_back({{base + 2 * i, (2 * i + 1 < K) ? base + 2 * i + 1 : -1}});
}
inline void steps_acc(std::vector<Step>& s, int ntiles, int base) {
  for (int t = 0; t < ntiles; ++t)
    for (int r = 0; r < 16; ++r) s.push_back({{base + acc_feat(t, r, 0), base + acc_feat(t, r, 1)}});
}
inline void step_bias(std::vector<Step>& s) { s.push_back({{-2, -1}}); }

inline Tile tile_full(const HostLayer* L, int t, int row_off = 0, bool bias_ok = true) {
  Tile tl;
  for (int i = 0; i < 32; ++i) {
    const int c = 32 * t + i;
    if (c < L->out) tl[i] = Col{L, c, row_off, bias_ok};
  }
  return tl;
}
// Output `regs[r]` lands in accumulator register r of BOTH half-waves.
inline Tile tile_by_reg(const std::vector<Col>& regs) {
  Tile tl;
  for (int i = 0; i < 32; ++i) {
    const int r = (i & 3) + 4 * (i >> 3);
    if (r < (int)regs.size()) tl[i] = regs[r];
  }
  return tl;
}

// Fragments of dot_out (rc_dev_mlp.h): per output, per tile, per accumulator register the weight of the feature that
// register holds on each half-wave; then one bias fragment per output.
inline std::vector<float> pack_dot(const std::vector<Col>& outs, int ntiles, bool pad_to_piece = kRcSplit) {
  std::vector<float> v;
  for (const Col& c : outs)
    for (int t = 0; t < ntiles; ++t)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = acc_feat(t, r, lane >> 5) + c.row_off;
          v.push_back((c.L && row < c.L->in) ? c.L->kernel[(size_t)row * c.L->out + c.col] : 0.0f);
        }
  for (const Col& c : outs)
    for (int lane = 0; lane < 64; ++lane) v.push_back(c.L ? c.L->bias[c.col] : 0.0f);
  if (pad_to_piece) v.resize((size_t)rc_dfr((int)outs.size(), ntiles) * 64, 0.0f);
  return v;
}

// weight of k-step s, half-wave h, for the output row lane i of a tile holds
inline float pack_value(const Step& st, int h, const Col& c) {
  if (!c.L) return 0.0f;
  const int row = st.row[h];
  if (row == -2) return c.bias_ok ? c.L->bias[c.col] : 0.0f;
  if (row >= 0) {
    const int rr = row + c.row_off;
    if (rr < c.L->in) return c.L->kernel[(size_t)rr * c.L->out + c.col];
  }
  return 0.0f;
}

// x == hi + mid + lo exactly, each piece the TRUNCATED top 16 bits of what is left (a bf16): 8 significand bits apiece
inline void split3(float x, uint16_t (&piece)[3]) {
  float r = x;
  for (int p = 0; p < 3; ++p) {
    uint32_t u;
    memcpy(&u, &r, 4);
    u &= 0xffff0000u;
    float top;
    memcpy(&top, &u, 4);
    piece[p] = (uint16_t)(u >> 16);
    r = r - top;               // exact: top shares r's leading bits
  }
}

inline std::vector<float> pack_f32(const std::vector<Step>& steps, const std::vector<Tile>& tiles) {
  const size_t NT = tiles.size();
  std::vector<float> out(steps.size() * NT * 64, 0.0f);
  for (size_t s = 0; s < steps.size(); ++s)
    for (size_t t = 0; t < NT; ++t)
      for (int lane = 0; lane < 64; ++lane) out[(s * NT + t) * 64 + lane] = pack_value(steps[s], lane >> 5, tiles[t][lane & 31]);
  return out;
}

// split form (see rc_lfr): [block q][tile t][piece p][lane][4 dwords]; dword d of a lane = bf16 of step 8 q + 2 d in the low
// half, of step 8 q + 2 d + 1 in the high half; steps past the layer's last are zero
inline std::vector<float> pack_split(const std::vector<Step>& steps, const std::vector<Tile>& tiles) {
  const size_t NT = tiles.size(), NB = (steps.size() + 7) / 8;
  std::vector<uint32_t> out(NB * NT * 3 * 64 * 4, 0u);
  for (size_t s = 0; s < steps.size(); ++s)
    for (size_t t = 0; t < NT; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        uint16_t pc[3];
        split3(pack_value(steps[s], lane >> 5, tiles[t][lane & 31]), pc);
        const size_t q = s / 8, j = s % 8;
        for (int p = 0; p < 3; ++p) {
          uint32_t& d = out[((((q * NT + t) * 3 + p) * 64) + lane) * 4 + j / 2];
          d = (j & 1) ? ((d & 0x0000ffffu) | ((uint32_t)pc[p] << 16)) : ((d & 0xffff0000u) | pc[p]);
        }
      }
  std::vector<float> f(out.size());
  memcpy(f.data(), out.data(), out.size() * 4);
  return f;
}

inline std::vector<float> pack(const std::vector<Step>& steps, const std::vector<Tile>& tiles) {
  return kRcSplit ? pack_split(steps, tiles) : pack_f32(steps, tiles);
}


// A linear layer `b` (no activation) folded into the layer L that consumes it: rows [row0, row0 + b.out) of L take b's
// output, the following `extra_rows` rows are kept as they are.  W' = W_b W[rows], b' = b_b W[rows] (fp64 products,
// rounded once); L's own bias is added where the result is packed.
inline HostLayer fold_linear(const HostLayer& b, const HostLayer& L, int row0, int extra_rows) {
  const int FE = b.in, BW = b.out;
  HostLayer f;
  f.in = FE + extra_rows; f.out = L.out;
  f.kernel.assign((size_t)f.in * f.out, 0.0f); f.bias.assign(f.out, 0.0f);
  f.have_kernel = f.have_bias = true;
  for (int o = 0; o < L.out; ++o) {
    for (int i = 0; i < FE; ++i) {
      double a = 0.0;
      for (int m = 0; m < BW; ++m) a += (double)b.kernel[(size_t)i * BW + m] * (double)L.kernel[(size_t)(row0 + m) * L.out + o];
      f.kernel[(size_t)i * f.out + o] = (float)a;
    }
    double bb = 0.0;
    for (int m = 0; m < BW; ++m) bb += (double)b.bias[m] * (double)L.kernel[(size_t)(row0 + m) * L.out + o];
    f.bias[o] = (float)bb;
    for (int e = 0; e < extra_rows; ++e) f.kernel[(size_t)(FE + e) * f.out + o] = L.kernel[(size_t)(row0 + BW + e) * L.out + o];
  }
  return f;
}

// ref_utils.py:75-153 coefficient table for deg_view = 5, rounded to float32 like `mat` is.
inline double fact(int n) { double f = 1; for (int i = 2; i <= n; ++i) f *= i; return f; }
inline double gen_binom(double a, int k) { double p = 1; for (int j = 0; j < k; ++j) p *= (a - j); return p / fact(k); }
inline double sph_coeff(int l, int m, int k) {
  const double al = ((m & 1) ? -1.0 : 1.0) * pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) *
                    gen_binom(0.5 * (l + k + m - 1.0), l);
  return sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * al;
}
inline void build_ide_table(RcIdeTable& tb) {
  memset(&tb, 0, sizeof(tb));
  int i = 0;
  for (int d = 0; d < 5; ++d) {
    const int l = 1 << d;
    for (int m = 0; m <= l; ++m, ++i) {
      tb.m[i] = m;
      tb.sigma[i] = (float)(0.5 * l * (l + 1));
      for (int k = 0; k <= l - m; ++k) tb.coef[i][k] = (float)sph_coeff(l, m, k);
    }
  }
}


}  // namespace rcpack
