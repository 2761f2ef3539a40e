// Voxeliser on the GPU: analytic shapes placed with <place_fiber> -> phase volume fractions and interface normals.
//
// What is computed is the reference's definition (LSSolver::initPhi F:17489-17581, integratePhiVoxel F:16622-16752):
//   * a voxel (centre p, half diagonal r0) sees the shapes of its material with bounding-ball distance <= r0 and signed
//     distance d <= r0 (closestFibers F:3336-3361);
//   * |d_min| >= r0: the voxel is full (d_min < 0) or empty;
//   * otherwise an octree refinement: a node (centre q, half diagonal r) is FULL if some shape has d(q) <= -r, else it
//     keeps the shapes with |d(q)| < r; it becomes a leaf when the error estimate (r K)^2 (r / r0)^(2/3) of the closest
//     shape (curvature K) drops below smooth_tol (smooth_levels < 0) or at depth smooth_levels; a leaf's volume is the
//     sum over its shapes of the box volume cut off by the tangent plane at the closest surface point, clipped to the
//     box volume; every inner node clips the sum of its children the same way.
// How it is computed is this library's own:
//   * membership of a node's list depends only on the node (the distance is 1-Lipschitz and a child's centre lies
//     exactly r_child from its parent's, so a shape inside a child's band is inside every ancestor's): no lists are
//     carried down the tree -- every node re-derives its shapes from the candidate list of its 8^3 voxel brick, and
//     the tree is walked by an explicit per-thread stack;
//   * the plane / box cut is closed form: the cut fraction is a repeated moving average of a ramp (one average per axis,
//     width |n_i| d_i), evaluated piece by piece with exact quadrature -- sums of non-negative terms, stable for any
//     normal, no polyhedron is ever built (box_fraction_below_plane);
//   * bricks of 8 x 8 x 8 voxels get their candidate shapes from a host-side binning of the bounding balls; a first
//     kernel classifies every voxel (empty / full / interface) and compacts the interface voxels, a second one walks
//     their trees -- the expensive 3 % of the voxels get their own, evenly loaded launch: one thread per interface voxel
//     on grids that have enough of them, a team of 8 / 64 / 512 threads (one subtree each) on coarse ones.
// The checker (oracle/c/fg_voxel_ref.cpp) is the host restatement of the reference's recursive form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "../../include/fibergen_amd.h"
#include "fg_hip_util.h"
#include "fg_plane_cut.h"

namespace {

constexpr int kBrick = 8;          // voxels per brick edge
constexpr int kMaxDepth = 20;      // refinement levels below the voxel (8^20 nodes: never reached by a sane project)
constexpr double kEps = 2.220446049250313e-16;

struct V3 {
  double x, y, z;
};
__host__ __device__ inline V3 mk(double a, double b, double c) { V3 r = {a, b, c}; return r; }
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__host__ __device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline double norm(V3 a) { return sqrt(dot(a, a)); }

// CapsuleFiber (F:5237-5524: sphero-cylinder around the segment c1 + t a, t in [0, L]) / HalfSpaceFiber (F:5529-5640)
struct Shape {
  int kind;       // 0 capsule, 1 half space
  int material;
  V3 a, rvec, c1, c;   // capsule: unit axis, a vector of length R orthogonal to it, first end point, centre
  double R, L, B;      // radius, cylinder length, bounding-ball radius
  V3 p, n;             // half space: point on the plane, outward unit normal
};

// signed distance of q and the closest surface point (distanceTo)
__host__ __device__ inline double shape_distance(const Shape& s, V3 q, V3* xs) {
  if (s.kind == 1) {
    const double d = dot(q - s.p, s.n);
    *xs = q - d * s.n;
    return d;
  }
  double t = dot(q - s.c1, s.a);
  t = fmin(fmax(0.0, t), s.L);
  V3 x = s.c1 + t * s.a;
  const double d = norm(q - x);
  if (d < kEps * s.R) x = x + s.rvec;
  else x = x + (s.R / d) * (q - x);
  *xs = x;
  return d - s.R;
}

// outward unit normal of the distance field at q (distanceGrad)
__host__ __device__ inline V3 shape_normal(const Shape& s, V3 q) {
  if (s.kind == 1) return s.n;
  double t = dot(q - s.c1, s.a);
  t = fmin(fmax(0.0, t), s.L);
  const V3 g = q - s.c1 - t * s.a;
  const double ng = norm(g);
  if (ng < 1.4901161193847656e-08) return ((t < 0.5 * s.L) ? -1.0 : 1.0) * s.a;   // sqrt(eps)
  return (1.0 / ng) * g;
}

__host__ __device__ inline double shape_ball_distance(const Shape& s, V3 q) {   // bbDistanceMin  F:3046
  if (s.kind == 1) return -INFINITY;
  return norm(q - s.c) - s.B;
}

// ------------------------------------------------------------------------------------------ plane / box cut (fg_plane_cut.h)
__device__ inline double box_fraction_below_plane(V3 xs_rel, V3 n, double dx, double dy, double dz) {
  const double xr[3] = {xs_rel.x, xs_rel.y, xs_rel.z}, nn[3] = {n.x, n.y, n.z}, dd[3] = {dx, dy, dz};
  return fg::box_fraction_below_plane(xr, nn, dd);
}

// ------------------------------------------------------------------------------------------ kernels
struct VoxGrid {
  int nx, ny, nz;
  double hx, hy, hz;   // voxel edges
  double x0, y0, z0;   // origin
  double r0;           // voxel half diagonal
  int bx, by, bz;      // bricks per axis
};

__device__ inline V3 voxel_centre(const VoxGrid& g, int i, int j, int k) {
  return mk(g.hx * (i + 0.5) + g.x0, g.hy * (j + 0.5) + g.y0, g.hz * (k + 0.5) + g.z0);
}

// one thread per voxel: empty / full voxels are final, interface voxels are appended to `iface`
__global__ __launch_bounds__(kBrick* kBrick* kBrick) void k_vox_classify(VoxGrid g, const Shape* shapes, const int* brick_start,
                                                                         const int* brick_list, double* phi, unsigned* iface,
                                                                         unsigned* iface_count) {
  const int b = blockIdx.x;
  const int bk = b % g.bz, bj = (b / g.bz) % g.by, bi = b / (g.bz * g.by);
  const int t = threadIdx.x;
  const int i = bi * kBrick + t / (kBrick * kBrick), j = bj * kBrick + (t / kBrick) % kBrick, k = bk * kBrick + t % kBrick;
  if (i >= g.nx || j >= g.ny || k >= g.nz) return;
  const V3 p = voxel_centre(g, i, j, k);
  bool any = false;
  double dmin = INFINITY;
  for (int q = brick_start[b]; q < brick_start[b + 1]; ++q) {
    const Shape& s = shapes[brick_list[q]];
    if (shape_ball_distance(s, p) <= g.r0) {
      V3 xs;
      const double d = shape_distance(s, p, &xs);
      if (d <= g.r0) {
        any = true;
        dmin = fmin(dmin, d);
      }
    }
  }
  const size_t o = ((size_t)i * g.ny + j) * g.nz + k;
  if (!any || fabs(dmin) >= g.r0) {
    phi[o] = (any && dmin < 0) ? 1.0 : 0.0;
    return;
  }
  iface[atomicAdd(iface_count, 1u)] = (unsigned)o;
}

// what a node of an interface voxel's octree needs: the voxel's candidate list (its brick's) and centre
struct RefineCtx {
  VoxGrid g;
  const Shape* shapes;
  const int* brick_list;
  int q0, q1;
  V3 p0;
  int smooth_levels;
  double smooth_tol;
  int* error;
};

// classification of the node (level, cen): true when it is closed -- full, empty or a leaf -- with its volume in *value
__device__ bool node_closed(const RefineCtx& c, int level, V3 cen, double* value) {
  const VoxGrid& g = c.g;
  const double sc = ldexp(1.0, -level);
  const double ex = g.hx * sc, ey = g.hy * sc, ez = g.hz * sc, r = g.r0 * sc, vbox = ex * ey * ez;
  bool full = false, any = false;
  double dmin = INFINITY, kmin = 0.0;
  for (int q = c.q0; q < c.q1; ++q) {
    const Shape& s = c.shapes[c.brick_list[q]];
    if (shape_ball_distance(s, c.p0) > g.r0) continue;   // not in the voxel's list
    V3 xs;
    if (level == 0) {
      const double d = shape_distance(s, c.p0, &xs);
      if (d > g.r0) continue;
      any = true;
      if (d < dmin) dmin = d, kmin = s.kind == 1 ? 0.0 : 1.0 / s.R;
    } else {
      if (shape_distance(s, c.p0, &xs) > g.r0) continue;
      const double d = shape_distance(s, cen, &xs);
      if (d <= -r) {
        full = true;
        break;
      }
      if (fabs(d) < r) {
        any = true;
        if (d < dmin) dmin = d, kmin = s.kind == 1 ? 0.0 : 1.0 / s.R;
      }
    }
  }
  if (full) {
    *value = vbox;
    return true;
  }
  if (!any) {
    *value = 0.0;
    return true;
  }
  bool leaf;
  if (c.smooth_levels < 0) {   // error estimate of the closest shape's tangent-plane approximation
    const double Kd = r * kmin;
    const double err = Kd > 1 ? 1.0 : Kd * Kd * pow(sc, 2.0 / 3.0);
    leaf = err < c.smooth_tol;
  } else {
    leaf = level >= c.smooth_levels;
  }
  if (!leaf && level == kMaxDepth) {
    *c.error = 1;
    leaf = true;
  }
  if (!leaf) return false;
  const V3 org = mk(cen.x - 0.5 * ex, cen.y - 0.5 * ey, cen.z - 0.5 * ez);
  double v = 0.0;
  for (int q = c.q0; q < c.q1; ++q) {
    const Shape& s = c.shapes[c.brick_list[q]];
    if (shape_ball_distance(s, c.p0) > g.r0) continue;
    V3 xs;
    if (shape_distance(s, c.p0, &xs) > g.r0) continue;
    const double d = shape_distance(s, cen, &xs);
    if (level > 0 && !(fabs(d) < r)) continue;
    v += vbox * box_fraction_below_plane(xs - org, shape_normal(s, xs), ex, ey, ez);
  }
  *value = fmin(v, vbox);
  return true;
}

__device__ inline V3 child_centre(const VoxGrid& g, int level, V3 cen, int c) {   // child c of the node (level, cen)
  const double sc = ldexp(1.0, -level);
  const double qx = 0.25 * g.hx * sc, qy = 0.25 * g.hy * sc, qz = 0.25 * g.hz * sc;
  return mk(cen.x + ((c & 4) ? qx : -qx), cen.y + ((c & 2) ? qy : -qy), cen.z + ((c & 1) ? qz : -qz));
}

// volume of the subtree below the node (level0, cen0): the octree walk with an explicit stack
__device__ double subtree_volume(const RefineCtx& c, int level0, V3 cen0) {
  double acc[kMaxDepth + 1];   // volume collected by the open node of every level
  int child[kMaxDepth + 1];    // next child of the open node of every level
  V3 cen[kMaxDepth + 1];
  int level = level0;
  cen[level] = cen0;
  bool entering = true;        // the node at `level` has just been reached (not yet classified)
  for (;;) {
    double value = 0.0;        // volume of this node, once closed
    bool closed = false;
    if (entering) {
      closed = node_closed(c, level, cen[level], &value);
      if (!closed) {
        acc[level] = 0.0;
        child[level] = 0;
      }
    }
    if (!closed) {
      // descend into the next child, or close the node when all eight are done
      if (child[level] < 8) {
        cen[level + 1] = child_centre(c.g, level, cen[level], child[level]++);
        ++level;
        entering = true;
        continue;
      }
      const double sc = ldexp(1.0, -level);
      value = fmin(acc[level], c.g.hx * sc * (c.g.hy * sc) * (c.g.hz * sc));
    }
    // the node is closed: hand its volume to the parent
    if (level == level0) return value;
    --level;
    acc[level] += value;
    entering = false;
  }
}

__device__ inline RefineCtx refine_ctx(const VoxGrid& g, const Shape* shapes, const int* brick_start, const int* brick_list, size_t o,
                                       int smooth_levels, double smooth_tol, int* error) {
  const int k = (int)(o % g.nz), j = (int)((o / g.nz) % g.ny), i = (int)(o / ((size_t)g.nz * g.ny));
  const int b = ((i / kBrick) * g.by + j / kBrick) * g.bz + k / kBrick;
  RefineCtx c = {g, shapes, brick_list, brick_start[b], brick_start[b + 1], voxel_centre(g, i, j, k), smooth_levels, smooth_tol, error};
  return c;
}

// one thread per interface voxel (grids with enough interface voxels to fill the device)
__global__ __launch_bounds__(64) void k_vox_refine(VoxGrid g, const Shape* shapes, const int* brick_start, const int* brick_list,
                                                   const unsigned* iface, unsigned n_iface, int smooth_levels, double smooth_tol,
                                                   double* phi, int* error) {
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_iface) return;
  const size_t o = iface[w];
  const RefineCtx c = refine_ctx(g, shapes, brick_start, brick_list, o, smooth_levels, smooth_tol, error);
  phi[o] = subtree_volume(c, 0, c.p0) / (g.hx * g.hy * g.hz);
}

// A team of 8^D threads per interface voxel, for grids with few of them (a coarse grid with deep trees kept ONE lane busy
// for seconds).  Thread t of the team owns the level-D node whose path from the root are the D octal digits of t; the
// nodes above it are classified by every thread below them (cheap), the subtree by its owner alone.  The values then
// climb the tree through LDS, the eight children of a node added in child order and clipped to the node's box exactly
// as the one-thread walk does: the result is bit-identical to k_vox_refine's.
template <int D>
__global__ __launch_bounds__((1 << (3 * D)) < 64 ? 64 : (1 << (3 * D))) void k_vox_refine_team(
    VoxGrid g, const Shape* shapes, const int* brick_start, const int* brick_list, const unsigned* iface, unsigned n_iface,
    int smooth_levels, double smooth_tol, double* phi, int* error) {
  constexpr int T = 1 << (3 * D), B = T < 64 ? 64 : T;
  __shared__ double vals[B];
  const int t = threadIdx.x % T;
  const unsigned w = blockIdx.x * (B / T) + threadIdx.x / T;
  const bool active = w < n_iface;
  const size_t o = active ? iface[w] : 0;
  int closed_at = D;   // level of the first closed node on this thread's path (D: its own subtree is open above)
  double mine = 0.0;
  if (active) {
    const RefineCtx c = refine_ctx(g, shapes, brick_start, brick_list, o, smooth_levels, smooth_tol, error);
    V3 cen = c.p0;
    for (int l = 0; l < D; ++l) {
      double v;
      if (node_closed(c, l, cen, &v)) {
        closed_at = l;
        mine = (t & ((1 << (3 * (D - l))) - 1)) == 0 ? v : 0.0;   // carried by the node's first descendant
        break;
      }
      cen = child_centre(g, l, cen, (t >> (3 * (D - 1 - l))) & 7);
    }
    if (closed_at == D) mine = subtree_volume(c, D, cen);
  }
  vals[threadIdx.x] = mine;
  __syncthreads();
#pragma unroll
  for (int m = D - 1; m >= 0; --m) {
    const int stride = 1 << (3 * (D - 1 - m));
    if (active && (t & (8 * stride - 1)) == 0) {
      double sum = 0.0;
      for (int ch = 0; ch < 8; ++ch) sum += vals[threadIdx.x + ch * stride];
      if (closed_at > m) {   // the node at level m is open: clip to its box
        const double sc = ldexp(1.0, -m);
        sum = fmin(sum, g.hx * sc * (g.hy * sc) * (g.hz * sc));
      }
      vals[threadIdx.x] = sum;
    }
    __syncthreads();
  }
  if (active && t == 0) phi[o] = vals[threadIdx.x] / (g.hx * g.hy * g.hz);
}

// interface normals: gradient of the distance to the closest shape of ANY material at the voxel centre
// (sampleSlice NORMALS  F:6905-6925); shapes staged through LDS in tiles
__global__ __launch_bounds__(256) void k_vox_normals(VoxGrid g, const Shape* shapes, int nshapes, double* normals) {
  __shared__ Shape tile[32];
  const size_t N = (size_t)g.nx * g.ny * g.nz;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = o < N;
  const int k = live ? (int)(o % g.nz) : 0, j = live ? (int)((o / g.nz) % g.ny) : 0, i = live ? (int)(o / ((size_t)g.nz * g.ny)) : 0;
  const V3 p = voxel_centre(g, i, j, k);
  double dbest = INFINITY;
  V3 nbest = mk(0, 0, 0);
  bool have = false;
  for (int base = 0; base < nshapes; base += 32) {
    __syncthreads();
    if (threadIdx.x < 32 && base + (int)threadIdx.x < nshapes) tile[threadIdx.x] = shapes[base + threadIdx.x];
    __syncthreads();
    const int cnt = min(32, nshapes - base);
    for (int q = 0; q < cnt; ++q) {
      V3 xs;
      const double d = shape_distance(tile[q], p, &xs);
      if (!have || d < dbest) {
        have = true;
        dbest = d;
        nbest = shape_normal(tile[q], p);
      }
    }
  }
  if (live) {
    normals[o] = nbest.x;
    normals[N + o] = nbest.y;
    normals[2 * N + o] = nbest.z;
  }
}

// a vector of length 1 orthogonal to the unit vector v (only used to pick a surface point on the capsule's axis)
V3 any_orthonormal(V3 v) {
  const double ax = std::fabs(v.x), ay = std::fabs(v.y), az = std::fabs(v.z);
  const V3 e = (ax <= ay && ax <= az) ? mk(1, 0, 0) : (ay <= az ? mk(0, 1, 0) : mk(0, 0, 1));
  V3 w = e - dot(e, v) * v;
  return (1.0 / norm(w)) * w;
}

template <class T>
struct DeviceArray {
  T* p = nullptr;
  ~DeviceArray() {
    if (p) (void)hipFree(p);
  }
  void alloc(size_t n) { FG_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); }
  void upload(const std::vector<T>& v) {
    alloc(v.size());
    if (!v.empty()) FG_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  }
};

}  // namespace

namespace {
std::atomic<int> g_team_depth{-1};   // fg_voxelize_team_depth: -1 = by the number of interface voxels
}

extern "C" int fg_voxelize_team_depth(int depth) { return g_team_depth.exchange(depth < 0 ? -1 : std::min(depth, 3)); }

extern "C" int fg_voxelize(const fg_fiber* fibers, int nfibers, int nx, int ny, int nz, double dx, double dy, double dz,
                           const double* x0, int nphases, int matrix_mat, int smooth_levels, double smooth_tol,
                           double* phi, double* normals, double* real_volume, int device, char* err, int errlen) {
  auto fail = [&](const std::string& m) {
    if (err && errlen > 0) std::snprintf(err, errlen, "%s", m.c_str());
    return FG_ERROR;
  };
  try {
    if (nx < 1 || ny < 1 || nz < 1 || nphases < 1 || !phi || !x0) return fail("fg_voxelize: bad arguments");
    if (nfibers > 0 && !fibers) return fail("fg_voxelize: fibers is NULL");
    std::vector<Shape> shapes(nfibers);
    for (int i = 0; i < nfibers; ++i) {
      const fg_fiber& f = fibers[i];
      Shape& s = shapes[i];
      s = Shape();
      s.kind = f.kind;
      s.material = f.material;
      if (f.material < 0 || f.material >= nphases) return fail("fg_voxelize: fiber material out of range");
      const V3 c = mk(f.c[0], f.c[1], f.c[2]), a = mk(f.a[0], f.a[1], f.a[2]);
      const double na = norm(a);
      if (f.kind == 0) {
        // CapsuleFiber(c, a, L0, R)  F:5254-5275: L0 is the length of the cylinder of equal volume, the cylindrical part
        // of the capsule is L0 - 4/3 R long
        s.R = std::fabs(f.R);
        s.L = std::max(0.0, std::fabs(f.L) - (4.0 / 3.0) * s.R);
        if (na != 0) s.a = (1.0 / na) * a;
        else if (s.L != 0) return fail("CapsuleFiber: given nonzero fiber length without orientation vector!");
        else s.a = mk(0, 0, 0);
        s.c = c;
        s.c1 = c - (s.L / 2) * s.a;
        s.B = s.L / 2 + s.R;
        s.rvec = na != 0 ? s.R * any_orthonormal(s.a) : mk(0, 0, 0);
      } else if (f.kind == 1) {   // HalfSpaceFiber(p, n)  F:5537-5549
        if (na == 0) return fail("HalfSpaceFiber: given zero normal vector!");
        s.n = (1.0 / na) * a;
        s.p = c;
      } else {
        return fail("Unknown fiber type");
      }
    }
    if (real_volume) {
      for (int m = 0; m < nphases; ++m) real_volume[m] = 0.0;
      for (const Shape& s : shapes)
        real_volume[s.material] += s.kind == 1 ? std::numeric_limits<double>::infinity() : M_PI * s.R * s.R * (s.L + 4.0 / 3.0 * s.R);
    }
    int ndev = 0;
    FG_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail("no HIP device available: fibergen_amd needs an AMD GPU (gfx950)");
    if (device < 0 || device >= ndev) return fail("invalid device index");
    FG_HIP_CHECK(hipSetDevice(device));

    VoxGrid g;
    g.nx = nx, g.ny = ny, g.nz = nz;
    g.hx = dx / nx, g.hy = dy / ny, g.hz = dz / nz;
    g.x0 = x0[0], g.y0 = x0[1], g.z0 = x0[2];
    g.r0 = 0.5 * std::sqrt(g.hx * g.hx + g.hy * g.hy + g.hz * g.hz);
    g.bx = (nx + kBrick - 1) / kBrick, g.by = (ny + kBrick - 1) / kBrick, g.bz = (nz + kBrick - 1) / kBrick;
    const size_t N = (size_t)nx * ny * nz;
    const long nbricks = (long)g.bx * g.by * g.bz;
    if (N >= (1ull << 32)) return fail("fg_voxelize: grid too large");

    DeviceArray<Shape> d_shapes;
    d_shapes.upload(shapes);
    DeviceArray<double> d_phi;
    d_phi.alloc(N);
    DeviceArray<unsigned> d_iface, d_count;
    d_iface.alloc(N);
    d_count.alloc(1);
    DeviceArray<int> d_error;
    d_error.alloc(1);
    FG_HIP_CHECK(hipMemset(d_error.p, 0, sizeof(int)));

    for (int m = 0; m < nphases; ++m) {
      double* out = phi + (size_t)m * N;
      if (m == matrix_mat) {   // the matrix is present everywhere until normalizePhi hands it the remainder
        std::fill(out, out + N, 1.0);
        continue;
      }
      // candidate shapes of every brick: bounding ball within (brick half diagonal + r0) of the brick centre
      std::vector<std::vector<int>> per_brick(nbricks);
      const double ebx = kBrick * g.hx, eby = kBrick * g.hy, ebz = kBrick * g.hz;
      const double rb = 0.5 * std::sqrt(ebx * ebx + eby * eby + ebz * ebz) + g.r0;
      for (int si = 0; si < nfibers; ++si) {
        const Shape& s = shapes[si];
        if (s.material != m) continue;
        int lo[3] = {0, 0, 0}, hi[3] = {g.bx - 1, g.by - 1, g.bz - 1};
        if (s.kind == 0) {
          const double reach = s.B + g.r0;
          const double cc[3] = {s.c.x - x0[0], s.c.y - x0[1], s.c.z - x0[2]}, eb[3] = {ebx, eby, ebz};
          for (int a = 0; a < 3; ++a) {
            lo[a] = std::max(lo[a], (int)std::floor((cc[a] - reach) / eb[a]) - 1);
            hi[a] = std::min(hi[a], (int)std::floor((cc[a] + reach) / eb[a]) + 1);
          }
        }
        for (int bi = lo[0]; bi <= hi[0]; ++bi)
          for (int bj = lo[1]; bj <= hi[1]; ++bj)
            for (int bk = lo[2]; bk <= hi[2]; ++bk) {
              const V3 bc = mk(x0[0] + (bi + 0.5) * ebx, x0[1] + (bj + 0.5) * eby, x0[2] + (bk + 0.5) * ebz);
              if (s.kind == 0 && norm(bc - s.c) - s.B > rb) continue;
              per_brick[((long)bi * g.by + bj) * g.bz + bk].push_back(si);
            }
      }
      std::vector<int> start(nbricks + 1, 0), list;
      for (long b = 0; b < nbricks; ++b) {
        start[b] = (int)list.size();
        list.insert(list.end(), per_brick[b].begin(), per_brick[b].end());
      }
      start[nbricks] = (int)list.size();
      DeviceArray<int> d_start, d_list;
      d_start.upload(start);
      d_list.upload(list);
      FG_HIP_CHECK(hipMemset(d_count.p, 0, sizeof(unsigned)));
      hipLaunchKernelGGL(k_vox_classify, dim3((unsigned)nbricks), dim3(kBrick * kBrick * kBrick), 0, 0, g, d_shapes.p, d_start.p,
                         d_list.p, d_phi.p, d_iface.p, d_count.p);
      FG_HIP_CHECK(hipGetLastError());
      unsigned n_iface = 0;
      FG_HIP_CHECK(hipMemcpy(&n_iface, d_count.p, sizeof(unsigned), hipMemcpyDeviceToHost));
      if (n_iface) {
        // threads per interface voxel: 8^depth, so that a coarse grid still fills the device (fg_voxelize_team_depth: test hook)
        int depth = n_iface <= 2048 ? 3 : n_iface <= 16384 ? 2 : n_iface <= 65536 ? 1 : 0;
        if (const int forced = g_team_depth.load(); forced >= 0) depth = forced;
        if (smooth_levels >= 0) depth = std::min(depth, smooth_levels);
#define FG_REFINE(KERNEL, BLOCKS, THREADS)                                                                                   \
  hipLaunchKernelGGL(KERNEL, dim3(BLOCKS), dim3(THREADS), 0, 0, g, d_shapes.p, d_start.p, d_list.p, d_iface.p, n_iface, \
                     smooth_levels, smooth_tol, d_phi.p, d_error.p)
        if (depth == 3) FG_REFINE(k_vox_refine_team<3>, n_iface, 512);
        else if (depth == 2) FG_REFINE(k_vox_refine_team<2>, n_iface, 64);
        else if (depth == 1) FG_REFINE(k_vox_refine_team<1>, (n_iface + 7) / 8, 64);
        else FG_REFINE(k_vox_refine, (n_iface + 63) / 64, 64);
#undef FG_REFINE
        FG_HIP_CHECK(hipGetLastError());
      }
      FG_HIP_CHECK(hipMemcpy(out, d_phi.p, N * sizeof(double), hipMemcpyDeviceToHost));
    }
    int herr = 0;
    FG_HIP_CHECK(hipMemcpy(&herr, d_error.p, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) return fail("fg_voxelize: interface refinement exceeded 20 levels (shapes far below the voxel size?)");
    if (normals) {
      if (shapes.empty()) {
        std::fill(normals, normals + 3 * N, 0.0);
      } else {
        DeviceArray<double> d_n;
        d_n.alloc(3 * N);
        hipLaunchKernelGGL(k_vox_normals, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, g, d_shapes.p, nfibers, d_n.p);
        FG_HIP_CHECK(hipGetLastError());
        FG_HIP_CHECK(hipMemcpy(normals, d_n.p, 3 * N * sizeof(double), hipMemcpyDeviceToHost));
      }
    }
    return FG_OK;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
