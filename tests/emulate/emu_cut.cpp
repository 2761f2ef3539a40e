// Host build of the voxeliser's plane / box cut (fibergen_amd/csrc/fg_plane_cut.h) for tests/test_plane_cut.py.
#include "../../fibergen_amd/csrc/fg_plane_cut.h"

extern "C" double emu_cut_fraction(double alpha, double a1, double a2, double a3) { return fg::cut_v3(alpha, a1, a2, a3); }
extern "C" double emu_box_fraction(const double* xs_rel, const double* n, const double* d) {
  return fg::box_fraction_below_plane(xs_rel, n, d);
}
