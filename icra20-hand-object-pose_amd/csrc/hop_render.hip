// hop_render.hip -- "next" row N2: PoseEstimator::rejectByRender without OpenGL.
//
//   src/perception/src/PoseEstimator.cpp:345-463   scene assembly, per-hypothesis render, per-pixel score, keep the best
//   src/perception/src/Renderer.cpp:41-81          addObject / doRender (camera axes, metres, clamp to [0.1, 2.0])
//   src/depth_sim/src/range_likelihood.cpp:391-449 projection matrix and camera transform
//   src/depth_sim/src/simulation_io.cpp:411-460    depth read-back (vertical flip, z-buffer -> millimetres, rounding)
//   src/depth_sim/src/model.cpp:114-211            flat vertex colours, no culling, no lighting
//
// The reference renders the hand meshes (fixed) plus the object mesh under every hypothesis with a fixed-function OpenGL
// pipeline, one hypothesis after the other, reads depth + colour back and scores every pixel on the CPU.  Here:
//   * the camera model those files define: window x = fx X/Z + cx, window y (top-down, after the read-back flip)
//     = fy Y/Z + (H - cy) -- the principal point ends up mirrored vertically --, one sample per pixel centre;
//   * a z-buffer rasteriser: the hand once per frame, the object once per hypothesis (one thread per (hypothesis, face),
//     ordered-float atomicMin per covered pixel; triangles crossing the near plane are clipped as OpenGL clips them, triangles
//     whose window box exceeds 256 pixels are queued and drawn by a workgroup each), composed with GL_LESS in the reference's
//     draw order (hand, then object);
//   * depth through the read-back's float expression, rounded to whole millimetres, clamped to [0.1, 2.0] m;
//   * the score loop of :398-440 with its always-true sub-conditions.  sum_mode 0 adds the 307 200 per-pixel terms into
//     one float in row order like the reference (after ~1e5 background pixels worth 2.0 each, millimetre-sized terms
//     round away: the result depends on that order) -- one lane per hypothesis fed through LDS by the rest of its workgroup;
//     sum_mode 1 reduces them in double (block reduction);
//   * the survivors: max(int(keep_ratio n), 10) smallest wrong ratios, ascending (the reference's priority queue).
// Parity unpinned: no OpenGL exists here to pin pixel coverage or depth quantisation against; the CPU restatement
// (oracle/render_oracle.cpp) is checked on hand-computed triangles.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
#include <vector>

#include "hop_ctx_ext.h"
#include "hop_math.h"

using namespace hop;

#define RDCHK(ctx, call)                                                             \
  do {                                                                               \
    hipError_t _e = (call);                                                          \
    if (_e != hipSuccess) {                                                          \
      hop_ctx_set_error((ctx), std::string(#call) + ": " + hipGetErrorString(_e));   \
      return HOP_E_HIP;                                                              \
    }                                                                                \
  } while (0)

namespace {

struct RCam {
  float fx, fy, cx, cy;
  int H, W;
};

struct Render : HopExt {
  DevBuf raw, real, hand_V, hand_F, hand_z, obj_V, obj_F, zbuf, terms, sums, poses, out_depth, out_owner, big_q;
  RCam cam{};
  int hand_nf = 0, obj_nv = 0, obj_nf = 0;
  bool have_frame = false, have_object = false;
  ~Render() override {
    for (DevBuf* b : {&raw, &real, &hand_V, &hand_F, &hand_z, &obj_V, &obj_F, &zbuf, &terms, &sums, &poses, &out_depth, &out_owner, &big_q}) b->release();
  }
};
Render* render_ext(hop_ctx* c) {
  HopExt*& e = hop_ctx_ext(c, HOP_EXT_RENDER);
  if (!e) e = new Render;
  return static_cast<Render*>(e);
}

constexpr unsigned Z_CLEAR = 0x3f800000u;  // 1.0f: glClearDepth(1.0)

// Utils::readDepthImage (Utils.cpp:36-55)
__global__ void k_real_depth(const unsigned short* __restrict__ raw, int n, double unit, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = (float)((double)(float)raw[i] * unit);
  if (d > 2.0 || d < 0.1) d = 0.0f;
  out[i] = d;
}

__device__ __forceinline__ float window_depth(double Z) { return (float)((2.0 / 1.9) * (1.0 - 0.1 / Z)); }
__device__ __forceinline__ float readback_m(float d) {  // simulation_io.cpp:427 + Renderer.cpp:68-72
  const float zn = 0.1f, zf = 2.0f;
  const unsigned short mm = (unsigned short)round((double)(1000 * (-zf * zn / ((zf - zn) * (d - zf / (zf - zn))))));
  float m = (float)mm / 1000.0f;
  if (m > 2.0f) m = 2.0f;
  if (m < 0.1f) m = 0.1f;
  return m;
}

// A triangle in window coordinates (x, y, 1/Z per vertex).
struct WinTri {
  double x[3], y[3], iz[3];
};
// OpenGL clips a primitive against the frustum before it rasterises it.  Every fragment is tested against [0.1, 2.0] below, so
// clipping at ANY plane 0 < Zc <= 0.1 draws the same pixels: the triangle is cut at Zc = 0.05 (Sutherland-Hodgman, one plane) to
// keep the perspective division away from Z <= 0, and the quadrilateral a cut can leave is a fan of two triangles.  A triangle
// wholly at Z >= Zc is projected from its float vertices as before (oracle/render_oracle.cpp clip_project: the same operations).
constexpr float Z_CLIP = 0.05f;
__device__ __forceinline__ int clip_project(const V3 q[3], const RCam& c, WinTri out[2]) {
  const bool in[3] = {q[0].z >= Z_CLIP, q[1].z >= Z_CLIP, q[2].z >= Z_CLIP};
  const double oy = (double)c.H - c.cy;
  if (in[0] && in[1] && in[2]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      out[0].x[k] = (double)c.fx * q[k].x / q[k].z + c.cx;
      out[0].y[k] = (double)c.fy * q[k].y / q[k].z + oy;
      out[0].iz[k] = 1.0 / q[k].z;
    }
    return 1;
  }
  if (!in[0] && !in[1] && !in[2]) return 0;
  double px[4], py[4], piz[4];
  int n = 0;
  for (int k = 0; k < 3; ++k) {
    const int m = (k + 1) % 3;
    const double ax = q[k].x, ay = q[k].y, az = q[k].z, bx = q[m].x, by = q[m].y, bz = q[m].z;
    if (in[k]) px[n] = (double)c.fx * ax / az + c.cx, py[n] = (double)c.fy * ay / az + oy, piz[n] = 1.0 / az, ++n;
    if (in[k] != in[m]) {  // the crossing point, always computed from the inside vertex
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
__device__ __forceinline__ int face_triangles(const float* __restrict__ V, const int* __restrict__ F, int f, const float* __restrict__ pose, const RCam& c,
                                               WinTri out[2]) {
  V3 q[3];
  for (int k = 0; k < 3; ++k) {
    const float* p = V + 3 * (size_t)F[3 * f + k];
    q[k] = v3(p[0], p[1], p[2]);
    if (pose) q[k] = m4_point(pose, q[k]);  // Utils::transformPolygonMesh
  }
  return clip_project(q, c, out);
}
struct TriBox {
  double area;
  int w0, w1, h0, h1;
};
__device__ __forceinline__ bool tri_box(const WinTri& t, const RCam& c, TriBox& b) {
  const double* x = t.x;
  const double* y = t.y;
  b.area = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
  if (b.area == 0.0) return false;
  b.w0 = max(0, (int)floor(fmin(fmin(x[0], x[1]), x[2]) - 0.5)), b.w1 = min(c.W - 1, (int)ceil(fmax(fmax(x[0], x[1]), x[2]) - 0.5));
  b.h0 = max(0, (int)floor(fmin(fmin(y[0], y[1]), y[2]) - 0.5)), b.h1 = min(c.H - 1, (int)ceil(fmax(fmax(y[0], y[1]), y[2]) - 0.5));
  return b.w0 <= b.w1 && b.h0 <= b.h1;
}
__device__ __forceinline__ void tri_pixel(const WinTri& t, double area, int w, int h, int W, unsigned* __restrict__ zb) {
  const double* x = t.x;
  const double* y = t.y;
  const double px = w + 0.5, py = h + 0.5;
  const double e0 = (x[2] - x[1]) * (py - y[1]) - (y[2] - y[1]) * (px - x[1]);
  const double e1 = (x[0] - x[2]) * (py - y[2]) - (y[0] - y[2]) * (px - x[2]);
  const double e2 = (x[1] - x[0]) * (py - y[0]) - (y[1] - y[0]) * (px - x[0]);
  if (!((e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0))) return;
  const double Z = area / (e0 * t.iz[0] + e1 * t.iz[1] + e2 * t.iz[2]);
  if (!(Z >= 0.1 && Z <= 2.0)) return;
  atomicMin(&zb[(size_t)h * W + w], __float_as_uint(window_depth(Z)));  // positive floats order like their bits
}

// one thread per (hypothesis, face); poses == nullptr: the vertices are used as they are (hand meshes, camera frame).  A triangle
// whose window box holds more than RASTER_BIG pixels (a mesh close to the camera, a coarse mesh: up to the whole image) is not walked
// by its thread -- that one lane would set the duration of the launch -- but queued for k_raster_big, where a workgroup shares its box.
// (queue full: walked here after all.)  The z-buffer's atomicMin makes the result independent of who draws what.
constexpr int RASTER_BIG = 256;
struct BigQueue {
  unsigned* count;
  unsigned long long* item;  // (hypothesis * nf + face) * 2 + sub-triangle
  unsigned cap;
};
__global__ void k_raster(const float* __restrict__ V, const int* __restrict__ F, int nf, const float* __restrict__ poses, int n_hyp, RCam c,
                         unsigned* __restrict__ zbuf, BigQueue bq) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nf * n_hyp) return;
  const int hyp = (int)(t / nf), f = (int)(t - (long long)hyp * nf);
  unsigned* zb = zbuf + (size_t)hyp * c.H * c.W;
  WinTri tri[2];
  const int nt = face_triangles(V, F, f, poses ? poses + 16 * (size_t)hyp : nullptr, c, tri);
  for (int s = 0; s < nt; ++s) {
    TriBox b;
    if (!tri_box(tri[s], c, b)) continue;
    if ((long long)(b.w1 - b.w0 + 1) * (b.h1 - b.h0 + 1) > RASTER_BIG && bq.cap) {
      const unsigned at = atomicAdd(bq.count, 1u);
      if (at < bq.cap) {
        bq.item[at] = (unsigned long long)t * 2ull + (unsigned)s;
        continue;
      }
    }
    for (int h = b.h0; h <= b.h1; ++h)
      for (int w = b.w0; w <= b.w1; ++w) tri_pixel(tri[s], b.area, w, h, c.W, zb);
  }
}
// the queued triangles, one workgroup each (grid-stride over the queue: its length stays on the device)
__global__ __launch_bounds__(256) void k_raster_big(const float* __restrict__ V, const int* __restrict__ F, int nf, const float* __restrict__ poses, RCam c,
                                                    unsigned* __restrict__ zbuf, BigQueue bq) {
  const unsigned n = min(*bq.count, bq.cap);
  for (unsigned qi = blockIdx.x; qi < n; qi += gridDim.x) {
    const unsigned long long it = bq.item[qi];
    const long long t = (long long)(it >> 1);
    const int s = (int)(it & 1ull);
    const int hyp = (int)(t / nf), f = (int)(t - (long long)hyp * nf);
    unsigned* zb = zbuf + (size_t)hyp * c.H * c.W;
    WinTri tri[2];
    (void)face_triangles(V, F, f, poses ? poses + 16 * (size_t)hyp : nullptr, c, tri);
    TriBox b;
    if (!tri_box(tri[s], c, b)) continue;
    const int bw = b.w1 - b.w0 + 1, np = bw * (b.h1 - b.h0 + 1);
    for (int k = threadIdx.x; k < np; k += 256) tri_pixel(tri[s], b.area, b.w0 + k % bw, b.h0 + k / bw, c.W, zb);
  }
}

__device__ __forceinline__ void pixel_term(float real, unsigned zh, unsigned zo, float& diff, bool& roi) {
  // GL_LESS in draw order: the object (drawn last) owns the pixel only where it is strictly nearer than the hand
  roi = zo < zh;
  const float sim = readback_m(__uint_as_float(roi ? zo : zh));
  if ((real <= 0.1 || real >= 2.0) && (sim > 0.1 || sim < 2.0)) diff = 2.0f;
  else if ((sim <= 0.1 || sim >= 2.0) && (real > 0.1 || real < 2.0)) diff = 2.0f;
  else diff = fabsf(sim - real);
}

// sum_mode 0, pass 1: the per-pixel term of every hypothesis, [hypothesis][pixel]; bit 31 = object pixel (diff >= 0)
__global__ void k_score_terms(const float* __restrict__ real, const unsigned* __restrict__ hand_z, const unsigned* __restrict__ zbuf, int npx, int n_hyp,
                              unsigned* __restrict__ terms) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)npx * n_hyp) return;
  const int hyp = (int)(t / npx), px = (int)(t - (long long)hyp * npx);
  float diff;
  bool roi;
  pixel_term(real[px], hand_z[px], zbuf[(size_t)hyp * npx + px], diff, roi);
  terms[t] = __float_as_uint(diff) | (roi ? 0x80000000u : 0u);
}
// sum_mode 0, pass 2: the reference's loop (:402-437) adds the 307 200 terms of a hypothesis into two floats in row order -- a serial
// chain of float additions whose result depends on that order.  One workgroup per hypothesis: three wavefronts stream its terms into a
// double-buffered LDS tile, already split into the two sums' operands (adding +0 to the sum a term does not belong to leaves that sum's
// bits unchanged: the terms are >= 0) and count the object pixels (integers: any order); ONE lane of the fourth wavefront does nothing
// but the additions, 2.5 instructions per term.  (Before: one lane per hypothesis reading [pixel][hypothesis] with 64 loads in flight,
// 6.3 ms for the ~20 hypotheses of a frame -- bound by the latency of its own loads.)
constexpr int SCORE_CH = 2048;
__global__ __launch_bounds__(256) void k_score_serial(const unsigned* __restrict__ terms, int npx, int n_hyp, float roi_weight, float* __restrict__ wrong) {
  __shared__ __attribute__((aligned(16))) float sa[2][SCORE_CH], sb[2][SCORE_CH];
  __shared__ int s_cnt[3];
  const int hyp = blockIdx.x;
  const unsigned* __restrict__ tp = terms + (size_t)hyp * npx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nch = (npx + SCORE_CH - 1) / SCORE_CH;
  int my_roi = 0;
  float roi_diff = 0, bg_diff = 0;
  auto fill = [&](int c) {  // wavefronts 1..3
    const int b = c & 1, base = c * SCORE_CH;
    for (int i = (wave - 1) * 64 + lane; i < SCORE_CH; i += 192) {
      const int px = base + i;
      const unsigned v = px < npx ? tp[px] : 0u;
      const float d = __uint_as_float(v & 0x7fffffffu);
      const bool roi = (v & 0x80000000u) != 0u;
      sa[b][i] = roi ? d : 0.f, sb[b][i] = roi ? 0.f : d;
      my_roi += roi ? 1 : 0;
    }
  };
  if (wave > 0) fill(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    if (wave > 0) {
      if (c + 1 < nch) fill(c + 1);
    } else if (lane == 0) {
      const float4* __restrict__ a4 = reinterpret_cast<const float4*>(sa[c & 1]);
      const float4* __restrict__ b4 = reinterpret_cast<const float4*>(sb[c & 1]);
#pragma unroll 4
      for (int i = 0; i < SCORE_CH / 4; ++i) {
        const float4 a = a4[i], b = b4[i];
        roi_diff += a.x, bg_diff += b.x;
        roi_diff += a.y, bg_diff += b.y;
        roi_diff += a.z, bg_diff += b.z;
        roi_diff += a.w, bg_diff += b.w;
      }
    }
    __syncthreads();
  }
  if (wave > 0) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_roi += __shfl_down(my_roi, off);
    if (lane == 0) s_cnt[wave - 1] = my_roi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int roi_cnt = s_cnt[0] + s_cnt[1] + s_cnt[2], bg_cnt = npx - roi_cnt;
    wrong[hyp] = roi_weight * roi_diff / roi_cnt + bg_diff / bg_cnt;
  }
}
// sum_mode 1: one block per hypothesis, sums reduced in double
__global__ __launch_bounds__(256) void k_score_reduce(const float* __restrict__ real, const unsigned* __restrict__ hand_z, const unsigned* __restrict__ zbuf, int npx,
                                                      float roi_weight, float* __restrict__ wrong) {
  __shared__ double sd[4][2];
  __shared__ int sc[4][2];
  const int hyp = blockIdx.x;
  double roi_diff = 0, bg_diff = 0;
  int roi_cnt = 0, bg_cnt = 0;
  for (int px = threadIdx.x; px < npx; px += 256) {
    float diff;
    bool roi;
    pixel_term(real[px], hand_z[px], zbuf[(size_t)hyp * npx + px], diff, roi);
    if (roi) roi_diff += diff, roi_cnt++;
    else bg_diff += diff, bg_cnt++;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    roi_diff += __shfl_down(roi_diff, off), bg_diff += __shfl_down(bg_diff, off);
    roi_cnt += __shfl_down(roi_cnt, off), bg_cnt += __shfl_down(bg_cnt, off);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sd[wave][0] = roi_diff, sd[wave][1] = bg_diff, sc[wave][0] = roi_cnt, sc[wave][1] = bg_cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double r = (sd[0][0] + sd[1][0]) + (sd[2][0] + sd[3][0]), b = (sd[0][1] + sd[1][1]) + (sd[2][1] + sd[3][1]);
    const int rc = sc[0][0] + sc[1][0] + sc[2][0] + sc[3][0], bc = sc[0][1] + sc[1][1] + sc[2][1] + sc[3][1];
    wrong[hyp] = roi_weight * (float)r / (float)rc + (float)b / (float)bc;
  }
}

__global__ void k_compose_image(const unsigned* __restrict__ hand_z, const unsigned* __restrict__ obj_z, int npx, float* __restrict__ depth_m,
                                unsigned char* __restrict__ owner) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= npx) return;
  const unsigned zh = hand_z[px], zo = obj_z ? obj_z[px] : Z_CLEAR;
  const bool roi = zo < zh;
  const unsigned z = roi ? zo : zh;
  depth_m[px] = readback_m(__uint_as_float(z));
  owner[px] = roi ? 2 : (zh < Z_CLEAR ? 1 : 0);
}

// both passes of the rasteriser over `work` = nf * n_hyp (hypothesis, face) pairs
constexpr unsigned BIG_CAP = 1u << 20;
int run_raster(hop_ctx* c, Render* r, const float* V, const int* F, int nf, const float* poses, int n_hyp, unsigned* zbuf) {
  hipStream_t st = hop_ctx_stream(c);
  RDCHK(c, r->big_q.ensure(16 + sizeof(unsigned long long) * (size_t)BIG_CAP));
  BigQueue bq{r->big_q.as<unsigned>(), reinterpret_cast<unsigned long long*>(static_cast<char*>(r->big_q.p) + 16), BIG_CAP};
  RDCHK(c, hipMemsetAsync(r->big_q.p, 0, 16, st));
  const long long work = (long long)nf * n_hyp;
  k_raster<<<(unsigned)((work + 63) / 64), 64, 0, st>>>(V, F, nf, poses, n_hyp, r->cam, zbuf, bq);
  k_raster_big<<<(unsigned)std::min<long long>(2 * work, 4096), 256, 0, st>>>(V, F, nf, poses, r->cam, zbuf, bq);
  return HOP_OK;
}

int fill_clear(hop_ctx* c, void* p, size_t words) {
  RDCHK(c, hipMemsetD32Async((hipDeviceptr_t)p, (int)Z_CLEAR, words, hop_ctx_stream(c)));
  return HOP_OK;
}

}  // namespace

extern "C" {

int hop_render_set_frame(hop_ctx* c, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float* hand_V, int hand_nv,
                         const int32_t* hand_F, int hand_nf) {
  if (!c || !depth_raw || H <= 0 || W <= 0 || !K9 || hand_nv < 0 || hand_nf < 0 || (hand_nf > 0 && (!hand_V || !hand_F))) return HOP_E_INVALID;
  for (int i = 0; i < 3 * hand_nf; ++i)
    if (hand_F[i] < 0 || hand_F[i] >= hand_nv) return HOP_E_INVALID;
  RDCHK(c, hipSetDevice(hop_ctx_device(c)));
  Render* r = render_ext(c);
  hipStream_t st = hop_ctx_stream(c);
  const size_t npx = (size_t)H * W;
  r->cam = RCam{K9[0], K9[4], K9[2], K9[5], H, W};
  RDCHK(c, r->raw.ensure(sizeof(uint16_t) * npx));
  RDCHK(c, r->real.ensure(sizeof(float) * npx));
  RDCHK(c, r->hand_z.ensure(sizeof(unsigned) * npx));
  RDCHK(c, hop_ctx_h2d(c, r->raw.p, depth_raw, sizeof(uint16_t) * npx));
  k_real_depth<<<(unsigned)((npx + 255) / 256), 256, 0, st>>>(r->raw.as<unsigned short>(), (int)npx, depth_unit, r->real.as<float>());
  int rc = fill_clear(c, r->hand_z.p, npx);
  if (rc) return rc;
  r->hand_nf = hand_nf;
  if (hand_nf > 0) {
    RDCHK(c, r->hand_V.ensure(sizeof(float) * 3 * (size_t)hand_nv));
    RDCHK(c, r->hand_F.ensure(sizeof(int) * 3 * (size_t)hand_nf));
    RDCHK(c, hop_ctx_h2d(c, r->hand_V.p, hand_V, sizeof(float) * 3 * (size_t)hand_nv));
    RDCHK(c, hop_ctx_h2d(c, r->hand_F.p, hand_F, sizeof(int) * 3 * (size_t)hand_nf));
    rc = run_raster(c, r, r->hand_V.as<float>(), r->hand_F.as<int>(), hand_nf, nullptr, 1, r->hand_z.as<unsigned>());
    if (rc) return rc;
  }
  RDCHK(c, hipGetLastError());
  RDCHK(c, hipStreamSynchronize(st));
  r->have_frame = true;
  return HOP_OK;
}

int hop_render_set_object(hop_ctx* c, const float* V, int nv, const int32_t* F, int nf) {
  if (!c || !V || !F || nv <= 0 || nf <= 0) return HOP_E_INVALID;
  for (int i = 0; i < 3 * nf; ++i)
    if (F[i] < 0 || F[i] >= nv) return HOP_E_INVALID;
  RDCHK(c, hipSetDevice(hop_ctx_device(c)));
  Render* r = render_ext(c);
  hipStream_t st = hop_ctx_stream(c);
  RDCHK(c, r->obj_V.ensure(sizeof(float) * 3 * (size_t)nv));
  RDCHK(c, r->obj_F.ensure(sizeof(int) * 3 * (size_t)nf));
  RDCHK(c, hop_ctx_h2d(c, r->obj_V.p, V, sizeof(float) * 3 * (size_t)nv));
  RDCHK(c, hop_ctx_h2d(c, r->obj_F.p, F, sizeof(int) * 3 * (size_t)nf));
  RDCHK(c, hipStreamSynchronize(st));
  r->obj_nv = nv, r->obj_nf = nf, r->have_object = true;
  return HOP_OK;
}

int hop_render_depth(hop_ctx* c, const float* pose16, float* depth_m_out, unsigned char* owner_out) {
  if (!c || !depth_m_out) return HOP_E_INVALID;
  Render* r = render_ext(c);
  if (!r->have_frame || (pose16 && !r->have_object)) return HOP_E_STATE;
  RDCHK(c, hipSetDevice(hop_ctx_device(c)));
  hipStream_t st = hop_ctx_stream(c);
  const size_t npx = (size_t)r->cam.H * r->cam.W;
  RDCHK(c, r->out_depth.ensure(sizeof(float) * npx));
  RDCHK(c, r->out_owner.ensure(npx));
  const unsigned* oz = nullptr;
  if (pose16) {
    RDCHK(c, r->zbuf.ensure(sizeof(unsigned) * npx));
    RDCHK(c, r->poses.ensure(sizeof(float) * 16));
    RDCHK(c, hop_ctx_h2d(c, r->poses.p, pose16, sizeof(float) * 16));
    int rc = fill_clear(c, r->zbuf.p, npx);
    if (rc) return rc;
    rc = run_raster(c, r, r->obj_V.as<float>(), r->obj_F.as<int>(), r->obj_nf, r->poses.as<float>(), 1, r->zbuf.as<unsigned>());
    if (rc) return rc;
    oz = r->zbuf.as<unsigned>();
  }
  k_compose_image<<<(unsigned)((npx + 255) / 256), 256, 0, st>>>(r->hand_z.as<unsigned>(), oz, (int)npx, r->out_depth.as<float>(), r->out_owner.as<unsigned char>());
  RDCHK(c, hipGetLastError());
  RDCHK(c, hop_ctx_d2h(c, depth_m_out, r->out_depth.p, sizeof(float) * npx));
  if (owner_out) RDCHK(c, hop_ctx_d2h(c, owner_out, r->out_owner.p, npx));
  RDCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_reject_by_render(hop_ctx* c, float roi_weight, float keep_ratio, int sum_mode, float* wrong_ratio_out, int* keep_index_out, int* n_keep_out) {
  if (!c || (sum_mode != 0 && sum_mode != 1) || !(keep_ratio >= 0)) return HOP_E_INVALID;
  Render* r = render_ext(c);
  if (!r->have_frame || !r->have_object) return HOP_E_STATE;
  RDCHK(c, hipSetDevice(hop_ctx_device(c)));
  hipStream_t st = hop_ctx_stream(c);
  HopHypView hv = hop_ctx_hyp(c);
  const int n = hv.n;
  if (n_keep_out) *n_keep_out = 0;
  if (n == 0) return HOP_OK;
  const size_t npx = (size_t)r->cam.H * r->cam.W;
  std::vector<float> wrong(n);
  RDCHK(c, r->sums.ensure(sizeof(float) * (size_t)n));
  // batches bound the z-buffers (1.2 MB per hypothesis at 640 x 480) and, in sum_mode 0, the staged terms
  const int HB = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)1 << 30) / (sizeof(unsigned) * npx)));
  RDCHK(c, r->zbuf.ensure(sizeof(unsigned) * npx * (size_t)HB));
  if (sum_mode == 0) RDCHK(c, r->terms.ensure(sizeof(unsigned) * npx * (size_t)HB));
  for (int h0 = 0; h0 < n; h0 += HB) {
    const int hb = std::min(HB, n - h0);
    int rc = fill_clear(c, r->zbuf.p, npx * (size_t)hb);
    if (rc) return rc;
    rc = run_raster(c, r, r->obj_V.as<float>(), r->obj_F.as<int>(), r->obj_nf, hv.pose + 16 * (size_t)h0, hb, r->zbuf.as<unsigned>());
    if (rc) return rc;
    if (sum_mode == 0) {
      const long long t = (long long)npx * hb;
      k_score_terms<<<(unsigned)((t + 255) / 256), 256, 0, st>>>(r->real.as<float>(), r->hand_z.as<unsigned>(), r->zbuf.as<unsigned>(), (int)npx, hb,
                                                                r->terms.as<unsigned>());
      k_score_serial<<<hb, 256, 0, st>>>(r->terms.as<unsigned>(), (int)npx, hb, roi_weight, r->sums.as<float>() + h0);
    } else {
      k_score_reduce<<<hb, 256, 0, st>>>(r->real.as<float>(), r->hand_z.as<unsigned>(), r->zbuf.as<unsigned>(), (int)npx, roi_weight, r->sums.as<float>() + h0);
    }
    RDCHK(c, hipGetLastError());
  }
  RDCHK(c, hop_ctx_d2h(c, wrong.data(), r->sums.p, sizeof(float) * (size_t)n));
  // the survivors, ascending wrong ratio (ties by position; NaN -- no object pixel -- last), gathered on the host: n is small
  std::vector<float> pose((size_t)16 * n), score(n);
  std::vector<int> id(n);
  RDCHK(c, hop_ctx_d2h(c, pose.data(), hv.pose, sizeof(float) * 16 * (size_t)n));
  RDCHK(c, hop_ctx_d2h(c, score.data(), hv.score, sizeof(float) * (size_t)n));
  RDCHK(c, hop_ctx_d2h(c, id.data(), hv.id, sizeof(int) * (size_t)n));
  RDCHK(c, hipStreamSynchronize(st));
  int num_to_keep = std::max((int)(keep_ratio * n), 10);
  num_to_keep = std::min(num_to_keep, n);
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const float x = wrong[a], y = wrong[b];
    if (std::isnan(x) || std::isnan(y)) return !std::isnan(x) && std::isnan(y);
    return x < y;
  });
  std::vector<float> p2((size_t)16 * num_to_keep), s2(num_to_keep);
  std::vector<int> i2(num_to_keep);
  for (int k = 0; k < num_to_keep; ++k) {
    std::copy(pose.begin() + 16 * (size_t)order[k], pose.begin() + 16 * (size_t)order[k] + 16, p2.begin() + 16 * (size_t)k);
    s2[k] = score[order[k]], i2[k] = id[order[k]];
    if (keep_index_out) keep_index_out[k] = order[k];
  }
  RDCHK(c, hop_ctx_h2d(c, hv.pose, p2.data(), sizeof(float) * 16 * (size_t)num_to_keep));
  RDCHK(c, hop_ctx_h2d(c, hv.score, s2.data(), sizeof(float) * (size_t)num_to_keep));
  RDCHK(c, hop_ctx_h2d(c, hv.id, i2.data(), sizeof(int) * (size_t)num_to_keep));
  RDCHK(c, hipStreamSynchronize(st));
  hop_ctx_hyp_set_count(c, num_to_keep);
  if (wrong_ratio_out) std::copy(wrong.begin(), wrong.end(), wrong_ratio_out);
  if (n_keep_out) *n_keep_out = num_to_keep;
  return HOP_OK;
}

}  // extern "C"
