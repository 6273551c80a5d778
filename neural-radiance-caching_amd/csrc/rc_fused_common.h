// Declarations shared by the two forms of the fused cache kernel: rc_fused.hip (one wavefront per ray) and
// rc_fused2.hip (two wavefronts per ray, two workgroups per CU).  Same weight stream, same LDS step map, same arguments.
#pragma once
#include "rc_dev_grid.h"
#include "rc_dev_mlp.h"
#include "rc_dev_sample.h"

namespace rcfused {
using namespace rcdev;

constexpr int kTileStride = 33 * 64;          // floats between the two point-tiles of levels 0/1
constexpr int kScratch = 7 * 68;              // per-wave step-function scratch (floats)
constexpr int kAppTmp = 33;                   // act steps [33, 49): appearance features parked during the density MLP
constexpr int kFusedDense = kRcFusedDenseLevels;               // grid levels [0, 3) of every grid are dense (16, 32, 64 cells a side), the rest hashed: fused_geometry_ok
constexpr int kJac = 49;                      // act steps [49, 97): d feature / d position of the level-2 density grid

// fragments of the fused weight stream
template <int KS0> struct DensFrags {
  static constexpr int NO = KS0 == 17 ? 4 : 1;       // output rows: density (+ 3 predicted normals on the last level)
  static constexpr int D0 = 0, D1 = rc_lfr32(KS0, 2), DO = D1 + rc_lfr32(33, 2), B1 = DO + rc_dfr32(NO, 2), B0 = B1 + rc_lfr32(32, 2), END = B0 + rc_lfr32(32, 1);
};
constexpr int F_L0 = 0;                              // K = 6: KS0 = 4
constexpr int F_L1 = F_L0 + DensFrags<4>::B1;        // K = 7: KS0 = 5
constexpr int F_L2 = F_L1 + DensFrags<5>::B1;        // K = 32: KS0 = 17, with the 96 backward fragments
constexpr int F_SH = rc_align_piece(F_L2 + DensFrags<17>::END);      // the shader's split layers start on a whole piece
constexpr int NF_FUSED = F_SH + ShaderFrags::COUNT;

#ifdef RC_STAMPS
#define RC_FSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RC_FSTAMP(i) do { } while (0)
#endif

#ifdef RC_STAMPS
#define RC_FSTAMP_NOWAIT(i) do { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RC_FSTAMP_NOWAIT(i) do { } while (0)
#endif
struct RcFusedArgs {
  const float* origins; const float* directions; const float* viewdirs; const float* near; const float* far;
  const float* lights;
  int64_t n;
  const float* jitter[3];
  RcGridDev grid[4];
  const float* pair_table[RC_MAX_GRID_LEVELS];
  const float* cell_table[2][RC_MAX_GRID_LEVELS];
  const float* wstream; const float* ide_coef;
  USpec us[3];
  float anneal, padding, density_bias, contract_radius, bg;
  float pct[3];
  ShaderConsts sh;
  rc_outputs out;
  unsigned long long* stamps;
  int32_t stagger_cycles;   // rc_fused2.hip: every second workgroup that arrives on a CU starts this many cycles late (0 = off)
  int32_t prio_mode;        // rc_fused2.hip: wave priorities of the workgroup that arrived second on its CU (see the kernel)
  int32_t* cu_slots;        // rc_fused2.hip: arrival counters per physical CU (4096 entries, zero-initialised once)
  // FRONT variant (time-resolved cache): the proposal sampler only; what the launch-per-stage front end leaves in the
  // workspace for the stages behind it, in its layouts (np = n * 32 shaded samples)
  int32_t use_raydist; float raydist_p, raydist_premult, y_max;      // power-ladder distances (coord.py:223-260)
  float* f_tdist;          // [n][33]
  float* f_density;        // [np]
  float* f_means;          // SoA [3][np]
  float* f_normals_pred;   // SoA [3][np]
  float* f_normals_grad;   // SoA [3][np] (GRAD) or nullptr
  float* f_hbuf;           // [n][32 steps][64 lanes]: hidden feature in accumulator layout, one tile per ray
  float* f_app;            // feature-major [32][np]
};


}  // namespace rcfused

// rc_fused2.hip: the plain cache pass with two wavefronts per ray (GRAD: analytic normals requested)
void rc_launch_fused_team(const rcfused::RcFusedArgs& a, bool grad, hipStream_t stream);
