// Exchange plan of the slab-decomposed solver: which doubles of which buffer go to / come from which peer.
// Pure index arithmetic (no GPU): the driver turns it into device pointers, the C ABI exports it (fg_slab_plan) so the
// CPU test-suite can execute it over gloo on NumPy buffers and check the global transposition.
//
// Decomposition: rank r owns x planes [r nxl, (r+1) nxl), nxl = nx/P, of every real-space field and,
// between the two all-to-alls, the ky rows [r nyl, (r+1) nyl), nyl = ny/P, of every spectrum.
//   x-slab, plain layout    [nxl][ny][nzc]              what the z passes and the stencil sweeps see
//   x-slab, blocked layout  [q][nxl][nyl][nzc]          written by the forward y pass: block q is what peer q receives
//   y-slab                  [nx][nyl][nzc]              = [p][nxl][nyl][nzc]: block p came from peer p; the fused
//                                                        x pass runs on it as on a single-GPU field with ny := nyl
// One component is g.n = nxl*ny*nzp doubles in all three layouts; the all-to-all moves blocks of
// nxl*nyl*nzp doubles, one component at a time (so component c's transfer overlaps the transforms of c+1).
#pragma once

#include <vector>

#include "../../include/fibergen_amd.h"
#include "fg_common.h"

namespace fg {

struct SlabDims {
  int nx, ny, nz, nranks, rank;
  int nxl, nyl;
  long nzp, plane;   // doubles per z row, per x plane of one component
  long n;            // doubles per component (without halo planes)
  long ucs;          // doubles per displacement / moduli component including its 4 spare planes
  long block;        // doubles per all-to-all block
  bool loopback;     // test mode: a rank's own blocks / planes travel through the transport too (send / receive to itself)
  int ncomp_u;       // components of the displacement buffer whose halo planes travel (3; heat / porous: the one potential)
};

// All-to-all of the three components in ONE message per peer (comp = -1): a peer's block holds its three components one
// after the other, x-slab side [q][c][nxl][nyl][nzc], y-slab side [p][c][nxl][nyl][nzc] (the y pass writes / reads that
// layout, the fused x pass walks it with its (j >> split) * jump term).
inline long slab_peer_block(const SlabDims& d) { return 3 * d.block; }

inline SlabDims slab_dims(int nx, int ny, int nz, int nranks, int rank) {
  SlabDims d;
  d.nx = nx; d.ny = ny; d.nz = nz; d.nranks = nranks; d.rank = rank;
  d.nxl = nx / nranks;
  d.nyl = ny / nranks;
  const Grid g = make_grid(d.nxl, ny, nz, 1.0, 1.0, 1.0);
  d.nzp = g.nzp;
  d.plane = g.nyzp;
  d.n = g.n;
  d.ucs = g.n + 4 * g.nyzp;
  d.block = (long)d.nxl * d.nyl * g.nzp;
  d.loopback = false;
  d.ncomp_u = 3;
  return d;
}

// spare planes of a displacement / moduli component (see Grid::xw_lo / xw_hi)
inline long slab_hi_plane(const SlabDims& d) { return (long)d.nxl; }        // copy of the right neighbour's plane 0
inline long slab_lo_plane(const SlabDims& d) { return (long)d.nxl + 3; }    // copy of the left neighbour's last plane

// Ops of one exchange for `d.rank`, peers other than itself only (a rank's own all-to-all block is a local copy, see
// `self`; with d.loopback -- the single-GPU test of the RCCL transport -- the rank is its own peer instead).  Offsets and counts in doubles relative to the start of the named buffer.
struct SlabPlan {
  std::vector<fg_plan_op> ops;
  fg_plan_op self_src, self_dst;   // count == 0: nothing to copy locally
};

inline fg_plan_op plan_op(int send, int peer, int buffer, long offset, long count) {
  fg_plan_op o;
  o.send = send;
  o.peer = peer;
  o.buffer = buffer;
  o.offset = offset;
  o.count = count;
  return o;
}

// what: FG_PLAN_*; comp: component (all-to-all: 0..2, or -1 = the three components interleaved per peer; halos: ignored,
// all components of the exchange are listed)
inline SlabPlan slab_plan(const SlabDims& d, int what, int comp) {
  SlabPlan p;
  p.self_src = p.self_dst = plan_op(0, d.rank, 0, 0, 0);
  const int P = d.nranks, me = d.rank;
  const int left = (me + P - 1) % P, right = (me + 1) % P;
  switch (what) {
    case FG_PLAN_A2A_FORWARD:    // blocked x-slab (buffer S) -> y-slab (buffer R)
    case FG_PLAN_A2A_BACKWARD: { // y-slab (R) -> blocked x-slab (S)
      const int from = what == FG_PLAN_A2A_FORWARD ? FG_BUF_SPECTRUM_X : FG_BUF_SPECTRUM_Y;
      const int to = what == FG_PLAN_A2A_FORWARD ? FG_BUF_SPECTRUM_Y : FG_BUF_SPECTRUM_X;
      // comp >= 0: one component, blocks of d.block at comp * n + q * block; comp < 0: the three components of a peer in
      // one block of 3 * d.block at q * 3 * block
      const long base = comp < 0 ? 0 : (long)comp * d.n;
      const long blk = comp < 0 ? slab_peer_block(d) : d.block;
      for (int q = 0; q < P; ++q) {
        if (q == me && !d.loopback) continue;
        p.ops.push_back(plan_op(0, q, to, base + q * blk, blk));
      }
      for (int q = 0; q < P; ++q) {
        if (q == me && !d.loopback) continue;
        p.ops.push_back(plan_op(1, q, from, base + q * blk, blk));
      }
      if (!d.loopback) {
        p.self_src = plan_op(1, me, from, base + me * blk, blk);
        p.self_dst = plan_op(0, me, to, base + me * blk, blk);
      }
      break;
    }
    case FG_PLAN_HALO_U:      // three displacement components: plane 0 -> left's hi plane, last plane -> right's lo plane
    case FG_PLAN_HALO_MODULI: {
      const int buf = what == FG_PLAN_HALO_U ? FG_BUF_U : FG_BUF_MODULI;
      const int nc = what == FG_PLAN_HALO_U ? d.ncomp_u : 2;
      if (P == 1 && !d.loopback) break;   // periodic inside the slab: the driver copies its own planes
      for (int c = 0; c < nc; ++c) {
        const long b = (long)c * d.ucs;
        p.ops.push_back(plan_op(0, left, buf, b + slab_lo_plane(d) * d.plane, d.plane));
        p.ops.push_back(plan_op(0, right, buf, b + slab_hi_plane(d) * d.plane, d.plane));
      }
      for (int c = 0; c < nc; ++c) {
        const long b = (long)c * d.ucs;
        p.ops.push_back(plan_op(1, right, buf, b + (long)(d.nxl - 1) * d.plane, d.plane));
        p.ops.push_back(plan_op(1, left, buf, b, d.plane));
      }
      break;
    }
    case FG_PLAN_HALO_TAU: {  // strain-state pipeline: tau0 last plane -> right, (tau5, tau4) first planes -> left
      if (P == 1 && !d.loopback) break;
      p.ops.push_back(plan_op(0, left, FG_BUF_HALO_RECV_LO, 0, d.plane));         // tau0 of plane -1
      p.ops.push_back(plan_op(0, right, FG_BUF_HALO_RECV_HI, 0, 2 * d.plane));    // tau5, tau4 of plane nxl
      p.ops.push_back(plan_op(1, right, FG_BUF_HALO_SEND_HI, 0, d.plane));
      p.ops.push_back(plan_op(1, left, FG_BUF_HALO_SEND_LO, 0, 2 * d.plane));
      break;
    }
    default:
      break;
  }
  return p;
}

}  // namespace fg
