// Host-only check program for the weight packing of rc_pack_host.h: built WITHOUT HIP under
// -fsanitize=address,undefined (`make hostcheck`), driven by tests/test_hostcheck.py, which writes the inputs as raw
// float32 files, runs this program and compares what it writes with numpy restatements of the same layouts.
//
//   hostcheck <dir>      reads  <dir>/in_*.bin   writes <dir>/out_*.bin
//
// Cases (sizes fixed here and in the test):
//   pack      a [K_IN = 37, OUT = 70] layer (+ bias) as 3 tiles over [natural steps | bias step]
//   pack_acc  a [64, 40] layer over accumulator-order steps (2 tiles of hidden features), no bias, row offset 0
//   by_reg    5 single-column outputs in a "by register" tile over natural steps + bias
//   dot       pack_dot of 3 outputs over 2 tiles
//   fold      bottleneck [12 -> 16] folded into a consumer [16 + 5 -> 9] with 5 extra rows, row0 = 0
//   cells     cell table of a dense N = 5 level with F = 2 features (the index rule of k_build_cells)
//   ide       the directional-encoding coefficient table
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "rc_pack_host.h"

using namespace rcpack;

static std::vector<float> read_f32(const std::string& path, size_t count) {
  std::vector<float> v(count);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), sizeof(float), count, f) != count) { fprintf(stderr, "hostcheck: cannot read %zu floats from %s\n", count, path.c_str()); exit(2); }
  fclose(f);
  return v;
}
static void write_f32(const std::string& path, const std::vector<float>& v) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(v.data(), sizeof(float), v.size(), f) != v.size()) { fprintf(stderr, "hostcheck: cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}
static HostLayer layer(const std::string& dir, const char* name, int in, int out) {
  HostLayer L;
  L.in = in; L.out = out;
  L.kernel = read_f32(dir + "/in_" + name + "_kernel.bin", (size_t)in * out);
  L.bias = read_f32(dir + "/in_" + name + "_bias.bin", (size_t)out);
  L.have_kernel = L.have_bias = true;
  return L;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: hostcheck <dir>\n"); return 2; }
  const std::string dir = argv[1];
  {
    HostLayer L = layer(dir, "a", 37, 70);
    std::vector<Step> s;
    steps_natural(s, 37, 0); step_bias(s);
    write_f32(dir + "/out_pack.bin", pack(s, {tile_full(&L, 0), tile_full(&L, 1), tile_full(&L, 2)}));
  }
  {
    HostLayer L = layer(dir, "b", 64, 40);
    std::vector<Step> s;
    steps_acc(s, 2, 0);
    write_f32(dir + "/out_pack_acc.bin", pack(s, {tile_full(&L, 0, 0, false), tile_full(&L, 1, 0, false)}));
  }
  {
    HostLayer L = layer(dir, "c", 9, 5);
    std::vector<Step> s;
    steps_natural(s, 9, 0); step_bias(s);
    std::vector<Col> regs;
    for (int c = 0; c < 5; ++c) regs.push_back(Col{&L, c});
    write_f32(dir + "/out_by_reg.bin", pack(s, {tile_by_reg(regs)}));
  }
  {
    HostLayer L = layer(dir, "d", 64, 3);
    write_f32(dir + "/out_dot.bin", pack_dot({Col{&L, 0}, Col{&L, 1}, Col{&L, 2}}, 2));
  }
  {
    HostLayer b = layer(dir, "bott", 12, 16), L = layer(dir, "cons", 21, 9);
    HostLayer f = fold_linear(b, L, 0, 5);
    if (f.in != 17 || f.out != 9) return 3;
    write_f32(dir + "/out_fold_kernel.bin", f.kernel);
    write_f32(dir + "/out_fold_bias.bin", f.bias);
  }
  {
    const int N = 5, F = 2, M = N + 3;
    std::vector<float> src = read_f32(dir + "/in_grid.bin", (size_t)N * N * N * F);
    std::vector<float> dst((size_t)M * M * M * 8 * F);
    for (int64_t i = 0; i < (int64_t)M * M * M * 8; ++i) {
      int64_t e;
      const bool inside = rc_cell_corner(N, i, &e);
      for (int f = 0; f < F; ++f) dst[(size_t)i * F + f] = inside ? src[(size_t)e * F + f] : 0.0f;   // `e` is only read when inside
    }
    write_f32(dir + "/out_cells.bin", dst);
  }
  {
    RcIdeTable tb;
    build_ide_table(tb);
    std::vector<float> v;
    for (int i = 0; i < RC_IDE_TERMS; ++i) for (int k = 0; k < RC_IDE_ZPOW; ++k) v.push_back(tb.coef[i][k]);
    for (int i = 0; i < RC_IDE_TERMS; ++i) v.push_back((float)tb.m[i]);
    for (int i = 0; i < RC_IDE_TERMS; ++i) v.push_back(tb.sigma[i]);
    write_f32(dir + "/out_ide.bin", v);
  }
  printf("hostcheck ok\n");
  return 0;
}
