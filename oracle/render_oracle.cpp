// TEST INFRASTRUCTURE ONLY (imported by tests/, smoke(), bench.py's cpu_baseline leg; never by the product).
//
// CPU restatement of PoseEstimator::rejectByRender ("next" row N2):
//   src/perception/src/PoseEstimator.cpp:345-463   scene assembly, per-hypothesis render, per-pixel score, keep the best
//   src/perception/src/Renderer.cpp:41-81          addObject / doRender (camera axes, metres, clamp to [0.1, 2.0])
//   src/depth_sim/src/range_likelihood.cpp:391-449 projection matrix and camera transform
//   src/depth_sim/src/simulation_io.cpp:411-460    depth read-back (flip, z-buffer -> millimetres, rounding), :486-507 colours
//   src/depth_sim/src/model.cpp:114-211            flat vertex colours, GL_POLYGON, no culling, no lighting
//
// PARITY UNPINNED: the images come from an OpenGL rasteriser (absent here: no GL, GLEW, GLUT, VTK); pixel coverage at
// triangle edges and the 24-bit depth quantisation are the driver's.  What is restated: the camera model those files
// define -- window x = fx X/Z + cx, window y (top-down after the read-back flip) = fy Y/Z + (H - cy), i.e. the principal
// point mirrored vertically, samples at pixel centres (+0.5) --, nearest fragment with GL_LESS in draw order (hand first,
// object last), depth = the read-back's float expression rounded to whole millimetres, the score loop with its float sums
// in row order and its always-true sub-conditions, and the selection of the smallest wrong ratios.  Checked on
// hand-computed triangles (tests/test_render_oracle.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

struct Cam {
  float fx, fy, cx, cy;
  int H, W;
};

// z-buffer value of an eye depth Z (metres) for zn = 0.1, zf = 2.0 (setupProjectionMatrix), as a float
inline float window_depth(double Z) { return (float)((2.0 / 1.9) * (1.0 - 0.1 / Z)); }
// simulation_io.cpp:427: the read-back's float expression, rounded to unsigned short millimetres
inline unsigned short readback_mm(float d) {
  const float zn = 0.1f, zf = 2.0f;
  return (unsigned short)std::round(1000 * (-zf * zn / ((zf - zn) * (d - zf / (zf - zn)))));
}

// A triangle in window coordinates (x, y, 1/Z per vertex).
struct WinTri {
  double x[3], y[3], iz[3];
};
// OpenGL clips a primitive against the frustum before it rasterises it; the planes that matter here are near (Z >= 0.1) and far.
// Every fragment is tested against [0.1, 2.0] below, so clipping at ANY plane 0 < Zc <= 0.1 draws the same pixels: the triangle is
// cut at Zc = 0.05 (Sutherland-Hodgman, one plane) only to keep the perspective division away from Z <= 0, and the quadrilateral a
// cut can leave is drawn as a fan of two triangles.  A triangle wholly at Z >= Zc is projected from its float vertices as before.
constexpr float Z_CLIP = 0.05f;
inline int clip_project(const float p[3][3], const Cam& c, WinTri out[2]) {
  const bool in[3] = {p[0][2] >= Z_CLIP, p[1][2] >= Z_CLIP, p[2][2] >= Z_CLIP};
  const double oy = (double)c.H - c.cy;
  if (in[0] && in[1] && in[2]) {
    for (int k = 0; k < 3; ++k) {
      out[0].x[k] = (double)c.fx * p[k][0] / p[k][2] + c.cx;
      out[0].y[k] = (double)c.fy * p[k][1] / p[k][2] + oy;
      out[0].iz[k] = 1.0 / p[k][2];
    }
    return 1;
  }
  if (!in[0] && !in[1] && !in[2]) return 0;
  double px[4], py[4], piz[4];
  int n = 0;
  for (int k = 0; k < 3; ++k) {
    const int m = (k + 1) % 3;
    const double ax = p[k][0], ay = p[k][1], az = p[k][2], bx = p[m][0], by = p[m][1], bz = p[m][2];
    if (in[k]) px[n] = (double)c.fx * ax / az + c.cx, py[n] = (double)c.fy * ay / az + oy, piz[n] = 1.0 / az, ++n;
    if (in[k] != in[m]) {  // the edge crosses the plane: the crossing point, always computed from the inside vertex
      const double ix = in[k] ? ax : bx, iy = in[k] ? ay : by, izz = in[k] ? az : bz;
      const double ox = in[k] ? bx : ax, oyy = in[k] ? by : ay, oz = in[k] ? bz : az;
      const double t = ((double)Z_CLIP - izz) / (oz - izz);
      const double X = ix + t * (ox - ix), Y = iy + t * (oyy - iy), Z = (double)Z_CLIP;
      px[n] = (double)c.fx * X / Z + c.cx, py[n] = (double)c.fy * Y / Z + oy, piz[n] = 1.0 / Z, ++n;
    }
  }
  int nt = 0;
  for (int k = 1; k + 1 < n; ++k, ++nt) {
    const int id[3] = {0, k, k + 1};
    for (int v = 0; v < 3; ++v) out[nt].x[v] = px[id[v]], out[nt].y[v] = py[id[v]], out[nt].iz[v] = piz[id[v]];
  }
  return nt;
}

// nearest fragment per pixel (window depth, float; 1.0 = cleared) of a triangle soup given in the camera frame
void rasterise(const float* V, const int* F, int nf, const Cam& c, std::vector<float>& depth, std::vector<unsigned char>* owner, unsigned char id) {
  for (int f = 0; f < nf; ++f) {
    float p[3][3];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) p[k][a] = V[3 * (size_t)F[3 * f + k] + a];
    WinTri tri[2];
    const int nt = clip_project(p, c, tri);
    for (int s = 0; s < nt; ++s) {
      const double* x = tri[s].x;
      const double* y = tri[s].y;
      const double* iz = tri[s].iz;
      const double area = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
      if (area == 0.0) continue;
      const int w0 = std::max(0, (int)std::floor(std::min({x[0], x[1], x[2]}) - 0.5)), w1 = std::min(c.W - 1, (int)std::ceil(std::max({x[0], x[1], x[2]}) - 0.5));
      const int h0 = std::max(0, (int)std::floor(std::min({y[0], y[1], y[2]}) - 0.5)), h1 = std::min(c.H - 1, (int)std::ceil(std::max({y[0], y[1], y[2]}) - 0.5));
      for (int h = h0; h <= h1; ++h)
        for (int w = w0; w <= w1; ++w) {
          const double px = w + 0.5, py = h + 0.5;
          const double e0 = (x[2] - x[1]) * (py - y[1]) - (y[2] - y[1]) * (px - x[1]);
          const double e1 = (x[0] - x[2]) * (py - y[2]) - (y[0] - y[2]) * (px - x[2]);
          const double e2 = (x[1] - x[0]) * (py - y[0]) - (y[1] - y[0]) * (px - x[0]);
          if (!((e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0))) continue;
          const double Z = area / (e0 * iz[0] + e1 * iz[1] + e2 * iz[2]);  // 1/Z is affine in window coordinates
          if (!(Z >= 0.1 && Z <= 2.0)) continue;                            // near / far planes
          const float d = window_depth(Z);
          const size_t i = (size_t)h * c.W + w;
          if (d < depth[i]) {  // GL_LESS
            depth[i] = d;
            if (owner) (*owner)[i] = id;
          }
        }
    }
  }
}

}  // namespace

extern "C" {

// depth image (metres as Renderer::doRender returns it: millimetre-rounded, clamped to [0.1, 2.0]; background 2.0) and
// owner (0 nothing, 1 hand, 2 object) of hand + object meshes given in the camera frame
int orc_render(const float* hand_V, const int* hand_F, int hand_nf, const float* obj_V, const int* obj_F, int obj_nf, const float* K9, int H, int W,
               float* depth_m_out, unsigned char* owner_out) {
  const Cam c{K9[0], K9[4], K9[2], K9[5], H, W};
  std::vector<float> depth((size_t)H * W, 1.0f);
  std::vector<unsigned char> owner((size_t)H * W, 0);
  if (hand_nf > 0) rasterise(hand_V, hand_F, hand_nf, c, depth, &owner, 1);
  if (obj_nf > 0) rasterise(obj_V, obj_F, obj_nf, c, depth, &owner, 2);
  for (size_t i = 0; i < depth.size(); ++i) {
    float m = (float)readback_mm(depth[i]) / 1000.0f;  // convertTo(CV_32FC1) / 1000.0
    if (m > 2.0f) m = 2.0f;
    if (m < 0.1f) m = 0.1f;
    depth_m_out[i] = m;
    if (owner_out) owner_out[i] = owner[i];
  }
  return 0;
}

// rejectByRender: wrong_ratio of every hypothesis (object mesh moved by its pose, hand fixed) and the indices kept, in the
// order the reference pops them (ascending wrong ratio; ties by index -- the reference's heap order is unspecified there).
// depth_m: the frame's depth in metres as Utils::readDepthImage leaves it (out-of-range pixels 0).
int orc_reject_by_render(const float* depth_m, int H, int W, const float* K9, const float* hand_V, const int* hand_F, int hand_nf, const float* obj_V, int obj_nv,
                         const int* obj_F, int obj_nf, const float* poses16, int n_hyp, float roi_weight, float keep_ratio, float* wrong_ratio_out, int* keep_out,
                         int* n_keep_out) {
  const Cam c{K9[0], K9[4], K9[2], K9[5], H, W};
  const size_t npx = (size_t)H * W;
  std::vector<float> hand_depth(npx, 1.0f);
  std::vector<unsigned char> hand_owner(npx, 0);
  if (hand_nf > 0) rasterise(hand_V, hand_F, hand_nf, c, hand_depth, &hand_owner, 1);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n_hyp; ++i) {
    const float* T = poses16 + 16 * (size_t)i;
    std::vector<float> Vt(3 * (size_t)obj_nv);
    for (int v = 0; v < obj_nv; ++v) {  // Utils::transformPolygonMesh -> pcl::transformPointCloud: ((m0 x + m1 y) + m2 z) + m3
      const float* p = obj_V + 3 * (size_t)v;
      for (int r = 0; r < 3; ++r) Vt[3 * (size_t)v + r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3];
    }
    std::vector<float> depth = hand_depth;
    std::vector<unsigned char> owner = hand_owner;
    rasterise(Vt.data(), obj_F, obj_nf, c, depth, &owner, 2);
    float roi_diff = 0, bg_diff = 0;
    int roi_cnt = 0, bg_cnt = 0;
    for (size_t px = 0; px < npx; ++px) {
      float sim = (float)readback_mm(depth[px]) / 1000.0f;
      if (sim > 2.0f) sim = 2.0f;
      if (sim < 0.1f) sim = 0.1f;
      const float real = depth_m[px];
      float diff = 0;
      if ((real <= 0.1 || real >= 2.0) && (sim > 0.1 || sim < 2.0)) diff = 2.0;       // (second operand always true)
      else if ((sim <= 0.1 || sim >= 2.0) && (real > 0.1 || real < 2.0)) diff = 2.0;  // (double literals, as there)
      else diff = std::abs(sim - real);
      if (owner[px] == 2) roi_diff += diff, roi_cnt++;
      else bg_diff += diff, bg_cnt++;
    }
    wrong_ratio_out[i] = roi_weight * roi_diff / roi_cnt + bg_diff / bg_cnt;
  }
  int num_to_keep = std::max((int)(keep_ratio * n_hyp), 10);
  num_to_keep = std::min(num_to_keep, n_hyp);
  std::vector<int> order(n_hyp);
  std::iota(order.begin(), order.end(), 0);
  // NaN ratios (no object pixel: 0/0) never compare greater in CompareWrongRatio; here they go last
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const float x = wrong_ratio_out[a], y = wrong_ratio_out[b];
    if (std::isnan(x) || std::isnan(y)) return !std::isnan(x) && std::isnan(y);
    return x < y;
  });
  for (int k = 0; k < num_to_keep; ++k) keep_out[k] = order[k];
  *n_keep_out = num_to_keep;
  return 0;
}

}  // extern "C"
