// On-device ray generation for pinhole and panoramic cameras.
//
// Replaces camera_utils.pixels_to_rays (internal/camera_utils.py:896-1072) + cast_ray_batch (:1225-1329) for the
// configuration of the BASELINE scenes: ProjectionType.PERSPECTIVE (and PANORAMIC = cast_spherical_rays, :1415-1443, the
// secondary-ray visualisation), no distortion, no NDC, no z_range, no pixel jitter, one camera per call.  Same arithmetic in the same order: pixel centre (x + 0.5, y + 0.5, 1) and its
// +1 neighbours in x and y through pixtocam, flip to OpenGL axes (y, z negated), rotate by camtoworld[:3, :3],
// viewdirs = directions / |directions|, radii = 0.5 (|dx - d| + |dy - d|) * 2 / sqrt(12).
#include "rc_internal.h"

namespace {

__device__ __forceinline__ void mat3_vec(const float* m, float x, float y, float z, float& ox, float& oy, float& oz) {
  // np.matmul(A, b[..., None]): row . column, left to right
  ox = (m[0] * x + m[1] * y) + m[2] * z;
  oy = (m[3] * x + m[4] * y) + m[5] * z;
  oz = (m[6] * x + m[7] * y) + m[8] * z;
}

__global__ void k_cast_rays(RcCastArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int px, py;
  if (a.pix_x) {
    px = a.pix_x[i]; py = a.pix_y[i];
  } else {
    px = a.x0 + (int)(i % a.width); py = a.y0 + (int)(i / a.width);
  }
  float d[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x = (float)(px + (k == 1 ? 1 : 0)) + 0.5f, y = (float)(py + (k == 2 ? 1 : 0)) + 0.5f;
    float cx, cy, cz;
    mat3_vec(a.pixtocam, x, y, 1.0f, cx, cy, cz);
    if (a.camtype == 1) {
      // ProjectionType.PANORAMIC (camera_utils.py:1013-1024): (theta, phi) = the first two components
      const float theta = cx, phi = cy;
      cx = -sinf(phi) * sinf(theta);
      cy = -cosf(phi);
      cz = -sinf(phi) * cosf(theta);
    }
    // OpenCV -> OpenGL: diag(1, -1, -1)
    cy = -cy; cz = -cz;
    if (k == 0 && a.imageplane) { a.imageplane[2 * i] = cx; a.imageplane[2 * i + 1] = cy; }
    mat3_vec(a.rot, cx, cy, cz, d[k][0], d[k][1], d[k][2]);
  }
  const float nrm = sqrtf((d[0][0] * d[0][0] + d[0][1] * d[0][1]) + d[0][2] * d[0][2]);
  auto dist = [&](int k) {
    const float ex = d[k][0] - d[0][0], ey = d[k][1] - d[0][1], ez = d[k][2] - d[0][2];
    return sqrtf((ex * ex + ey * ey) + ez * ez);
  };
  const float radius = (0.5f * (dist(1) + dist(2))) * 2.0f / 3.4641016151377544f;       // sqrt(12)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (a.origins) a.origins[3 * i + c] = a.trans[c];
    if (a.directions) a.directions[3 * i + c] = d[0][c];
    if (a.viewdirs) a.viewdirs[3 * i + c] = d[0][c] / nrm;
    if (a.lights) a.lights[3 * i + c] = a.light[c];
    if (a.look) a.look[3 * i + c] = -a.rot[3 * c + 2];       // -camtoworld[:3, 2]
    if (a.up) a.up[3 * i + c] = a.rot[3 * c + 1];            //  camtoworld[:3, 1]
  }
  if (a.radii) a.radii[i] = radius;
  if (a.near) a.near[i] = a.near_v;
  if (a.far) a.far[i] = a.far_v;
}

}  // namespace

void rc_launch_cast_rays(const RcCastArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_cast_rays, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
}
