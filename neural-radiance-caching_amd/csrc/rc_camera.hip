// On-device ray generation for pinhole and panoramic cameras.
//
// Replaces camera_utils.pixels_to_rays (internal/camera_utils.py:896-1072) + cast_ray_batch (:1225-1329), one camera per
// call: ProjectionType.PERSPECTIVE (the BASELINE scenes), PANORAMIC (= cast_spherical_rays, :1415-1443, the secondary-ray
// visualisation), FISHEYE / FISHEYE_EQUISOLID (:991-1011), radial + tangential distortion undone by the reference's 10
// Newton steps (:795-890), NDC rays (convert_to_ndc, :50-111, radii from the NDC origin offsets :1058-1066), sub-pixel
// jitter offsets handed over as tensors (:943-957), z_range cropping (:1143-1164, 1291-1299).  Same arithmetic in the
// same order: pixel centre (x + 0.5, y + 0.5, 1) and its +1 neighbours in x and y through pixtocam, flip to OpenGL axes (y, z negated), rotate by camtoworld[:3, :3],
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
    // pix_to_dir(pix_x_int (+ 1) + dx, pix_y_int (+ 1) + dy): the integer sum, then the offset, then the half pixel
    float x = (float)(px + (k == 1 ? 1 : 0)), y = (float)(py + (k == 2 ? 1 : 0));
    if (a.pix_dx) { x = x + a.pix_dx[i]; y = y + a.pix_dy[i]; }
    x = x + 0.5f; y = y + 0.5f;
    float cx, cy, cz;
    mat3_vec(a.pixtocam, x, y, 1.0f, cx, cy, cz);
    if (a.has_distortion) {
      // _radial_and_tangential_undistort (camera_utils.py:844-890) on (cx, cy); the third component becomes 1
      const float k1 = a.dist[0], k2 = a.dist[1], k3 = a.dist[2], k4 = a.dist[3], p1 = a.dist[4], p2 = a.dist[5];
      const float xd = cx, yd = cy;
      float ux = xd, uy = yd;
      for (int it = 0; it < 10; ++it) {
        // _compute_residual_and_jacobian (:795-841), operation for operation
        const float r = ux * ux + uy * uy;
        const float dd = 1.0f + r * (k1 + r * (k2 + r * (k3 + r * k4)));
        const float fx = ((dd * ux + (2.0f * p1 * ux) * uy) + p2 * (r + (2.0f * ux) * ux)) - xd;
        const float fy = ((dd * uy + (2.0f * p2 * ux) * uy) + p1 * (r + (2.0f * uy) * uy)) - yd;
        const float d_r = k1 + r * (2.0f * k2 + r * (3.0f * k3 + (r * 4.0f) * k4));
        const float d_x = (2.0f * ux) * d_r, d_y = (2.0f * uy) * d_r;
        const float fx_x = ((dd + d_x * ux) + (2.0f * p1) * uy) + (6.0f * p2) * ux;
        const float fx_y = (d_y * ux + (2.0f * p1) * ux) + (2.0f * p2) * uy;
        const float fy_x = (d_x * uy + (2.0f * p2) * uy) + (2.0f * p1) * ux;
        const float fy_y = ((dd + d_y * uy) + (2.0f * p2) * ux) + (6.0f * p1) * uy;
        const float den = fy_x * fx_y - fx_x * fy_y;
        const float xn = fx * fy_y - fy * fx_y, yn = fy * fx_x - fx * fy_x;
        const bool ok = fabsf(den) > 1e-9f;
        ux = ux + (ok ? xn / den : 0.0f);
        uy = uy + (ok ? yn / den : 0.0f);
      }
      cx = ux; cy = uy; cz = 1.0f;
    }
    if (a.camtype == 2 || a.camtype == 3) {
      // fisheye (camera_utils.py:991-1011): r = image-plane radius over the focal length
      const float r = sqrtf(cx * cx + cy * cy);
      const float theta = a.camtype == 2 ? fminf(3.14159265358979323846f, r) : 2.0f * asinf(r / 2.0f);
      const float s_over_r = sinf(theta) / r;
      cx = cx * s_over_r; cy = cy * s_over_r; cz = cosf(theta);
    }
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
  // viewdirs are taken before the NDC conversion (camera_utils.py:1032)
  const float nrm = sqrtf((d[0][0] * d[0][0] + d[0][1] * d[0][1]) + d[0][2] * d[0][2]);
  const float vd[3] = {d[0][0] / nrm, d[0][1] / nrm, d[0][2] / nrm};
  float o[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k][0] = a.trans[0]; o[k][1] = a.trans[1]; o[k][2] = a.trans[2]; }
  if (a.has_ndc) {
    // convert_to_ndc(origins, directions, pixtocam_ndc, near = 1) for the ray and its two neighbours (:50-111, 1052-1062)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float t = -(1.0f + o[k][2]) / d[k][2];
      const float ox = o[k][0] + t * d[k][0], oy = o[k][1] + t * d[k][1], oz = o[k][2] + t * d[k][2];
      const float nx = a.ndc_xmult * ox / oz, ny = a.ndc_ymult * oy / oz;
      const float ix = a.ndc_xmult * d[k][0] / d[k][2], iy = a.ndc_ymult * d[k][1] / d[k][2];
      o[k][0] = nx; o[k][1] = ny; o[k][2] = -1.0f;
      d[k][0] = ix - nx; d[k][1] = iy - ny; d[k][2] = 1.0f - (-1.0f);
    }
  }
  auto dist = [&](int k) {
    // distance of the neighbour's direction (NDC: of its origin) from the ray's
    const float* p = a.has_ndc ? o[k] : d[k];
    const float* q = a.has_ndc ? o[0] : d[0];
    const float ex = p[0] - q[0], ey = p[1] - q[1], ez = p[2] - q[2];
    return sqrtf((ex * ex + ey * ey) + ez * ez);
  };
  const float radius = (0.5f * (dist(1) + dist(2))) * 2.0f / 3.4641016151377544f;       // sqrt(12)
  if (a.has_z_range) {
    // rays_planes_intersection (:1143-1164) + the crop of cast_ray_batch (:1291-1299); np.minimum / np.maximum
    // propagate a NaN (directions.z == 0 with the origin on a plane), fminf / fmaxf would drop it
    const float t1 = (a.z_lo - o[0][2]) / d[0][2], t2 = (a.z_hi - o[0][2]) / d[0][2];
    const bool nan = (t1 != t1) || (t2 != t2);
    const float t_min = nan ? __builtin_nanf("") : fminf(t1, t2), t_max = nan ? __builtin_nanf("") : fmaxf(t1, t2);
    if (!(t_max < t_min)) {
      const float span = t_max - t_min;
#pragma unroll
      for (int c = 0; c < 3; ++c) { o[0][c] = o[0][c] + d[0][c] * t_min; d[0][c] = d[0][c] * span; }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (a.origins) a.origins[3 * i + c] = o[0][c];
    if (a.directions) a.directions[3 * i + c] = d[0][c];
    if (a.viewdirs) a.viewdirs[3 * i + c] = vd[c];
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
